"""HIP decode-path kernels against the committed golden vectors."""
import numpy as np
import os
import pytest
import torch

from _util import assert_bf16_close, assert_bf16_ulp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)


def test_golden_gla_decode_gpu():
    from vllm_hymeta import hip_ops

    z = np.load(os.path.join(GOLD, "gla_decode.npz"))
    state = bf(z["state0"]).cuda()
    slot = torch.from_numpy(z["slot_idx"]).cuda()
    live = torch.from_numpy(z["slot_idx"]) >= 0
    for s in range(4):
        out = hip_ops.gla_decode(bf(z[f"q{s}"]).cuda(), bf(z[f"k{s}"]).cuda(), bf(z[f"v{s}"]).cuda(),
                                 torch.from_numpy(z[f"g{s}"]).cuda(), state, slot)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu()[live], bf(z[f"o{s}"])[live], f"golden gla decode step {s}")
    assert_bf16_ulp(state.cpu(), bf(z["state_final"]), "golden gla decode final state", ulps=4)


def test_golden_reshape_and_cache_gpu():
    from vllm_hymeta import hip_ops

    z = np.load(os.path.join(GOLD, "reshape_and_cache.npz"))
    kv = bf(z["kv_before"]).cuda()
    hip_ops.reshape_and_cache(bf(z["key"]).cuda(), bf(z["value"]).cuda(), kv[0], kv[1],
                              torch.from_numpy(z["slot_mapping"]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(kv.cpu().view(torch.int16), bf(z["kv_after"]).view(torch.int16))


def test_golden_paged_decode_gpu():
    from vllm_hymeta import hip_ops

    z = np.load(os.path.join(GOLD, "paged_decode.npz"))
    q, kc, vc = bf(z["q"]).cuda(), bf(z["key_cache"]).cuda(), bf(z["value_cache"]).cuda()
    lens, bt = torch.from_numpy(z["cache_seqlens"]).cuda(), torch.from_numpy(z["block_tables"]).cuda()
    k2, v2 = bf(z["k2"]).cuda(), bf(z["v2"]).cuda()
    for name, kw in (("o_meta", dict(meta_key=k2, meta_value=v2, window_left=int(z["window_left_70b"]))),
                     ("o_full", dict(meta_key=k2, meta_value=v2, window_left=-1)),
                     ("o_7b", dict(window_left=int(z["window_left_7b"])))):
        out = hip_ops.meta_swa_paged_decode(q, kc, vc, lens, bt, **kw)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), torch.from_numpy(z[name]), f"golden paged decode {name}")
