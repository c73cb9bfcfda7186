"""CPU checks of the restatement itself: internal consistency and the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
D = 128


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)


def test_decode_steps_equal_recurrence():
    """T successive single-token steps (fp32 state) == the sequential recurrence."""
    from oracle import gla_decode_step_ref, gla_recurrent_ref

    g_ = gen(0)
    T, H, Hkv = 9, 4, 2
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    g = (torch.nn.functional.logsigmoid(torch.randn(T, Hkv, D, generator=g_)) / 16)
    state = torch.zeros(1, H, D, D)
    slot = torch.zeros(1, dtype=torch.int32)
    outs = [gla_decode_step_ref(q[t:t + 1], k[t:t + 1], v[t:t + 1], g[t:t + 1], state, slot)[0] for t in range(T)]
    o_ref, s_ref = gla_recurrent_ref(q, k, v, g)
    # bf16 rounding of k v^T in the decode kernel is the only modelled difference
    assert_bf16_close(torch.stack(outs), o_ref, "decode steps vs recurrence", rel=1e-2)
    assert torch.allclose(state[0].double(), s_ref, rtol=2e-2, atol=2e-2)


def test_prefill_chunk_resume_matches_single_pass():
    from oracle import gla_prefill_varlen_ref

    g_ = gen(1)
    T, H = 40, 2
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, H, D), randn_bf16(g_, T, H, D)
    g = torch.nn.functional.logsigmoid(torch.randn(T, H, D, generator=g_)) / 16
    s1 = torch.zeros(1, H, D, D)
    o1 = gla_prefill_varlen_ref(q, k, v, g, s1, torch.tensor([0]), [0, T], [0])
    s2 = torch.full((1, H, D, D), 5.0)  # stale content: context_len 0 must ignore it
    oa = gla_prefill_varlen_ref(q[:24], k[:24], v[:24], g[:24], s2, torch.tensor([0]), [0, 24], [0])
    ob = gla_prefill_varlen_ref(q[24:], k[24:], v[24:], g[24:], s2, torch.tensor([0]), [0, 16], [24])
    assert torch.allclose(torch.cat([oa, ob]).float(), o1.float(), rtol=2e-2, atol=2e-2)
    assert torch.allclose(s1, s2, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("Lq,Lk,M,wl", [(37, 37, 32, 15), (5, 50, 32, 20), (1, 70, 0, 16), (20, 20, 32, -1)])
def test_two_pass_merge_equals_single_softmax(Lq, Lk, M, wl):
    """The reference's two-launch + LSE-merge structure == one softmax over the union."""
    from oracle import meta_swa_dense_ref, meta_swa_two_pass_ref

    g_ = gen(2)
    H, Hkv = 4, 2
    q, k, v = randn_bf16(g_, Lq, H, D), randn_bf16(g_, Lk, Hkv, D), randn_bf16(g_, Lk, Hkv, D)
    if M == 0:
        pytest.skip("two-pass form needs meta keys")
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    q2 = randn_bf16(g_, M, H, D) if Lq == Lk else None
    dense, _ = meta_swa_dense_ref(q, k, v, k2, v2, q2, wl)
    two = meta_swa_two_pass_ref(q, k, v, k2, v2, q2, wl)
    assert_bf16_close(two, dense, "two-pass vs dense")


def test_dense_mask_matches_naive_definition():
    """Mask of the restatement == the mask naive_metatoken_flash_attn builds
    (fa_kernel_metatoken.py:388-400), for sliding_window=(window_left+1, 0)."""
    from oracle.meta_swa import _window_mask

    for Lq, Lk, wl in [(7, 7, 3), (3, 11, 4), (1, 9, 0), (6, 6, 100)]:
        q_id = Lk - Lq + torch.arange(Lq)
        k_id = torch.arange(Lk)
        naive = (q_id[:, None] >= k_id[None, :]) & ((q_id[:, None] - k_id[None, :]) < wl + 1)
        assert torch.equal(_window_mask(Lq, Lk, wl), naive)


def test_paged_decode_equals_dense_last_row():
    from oracle import gather_paged_ref, meta_swa_dense_ref, meta_swa_paged_decode_ref

    g_ = gen(3)
    H, Hkv, bs, L, M, wl = 4, 2, 16, 45, 32, 20
    nblk = 6
    kc, vc = randn_bf16(g_, nblk, bs, Hkv, D), randn_bf16(g_, nblk, bs, Hkv, D)
    bt = torch.tensor([[4, 1, 5]], dtype=torch.int32)
    q = randn_bf16(g_, 1, H, D)
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    out = meta_swa_paged_decode_ref(q, kc, vc, [L], bt, k2, v2, wl)
    kk, vv = gather_paged_ref(kc, bt[0], L), gather_paged_ref(vc, bt[0], L)
    assert torch.equal(kk[17], kc[1, 1])
    full_q = randn_bf16(g_, L, H, D)
    full_q[-1] = q[0]
    dense, _ = meta_swa_dense_ref(full_q, kk, vv, k2, v2, None, wl)
    assert torch.allclose(out[0], dense[-1], rtol=1e-9, atol=1e-12)


def test_slot_mapping_and_cache_shape():
    from oracle import kv_cache_shape_ref, slot_mapping_ref

    assert kv_cache_shape_ref(10, 256, 4, 128) == (2, 10, 256, 4, 128)
    with pytest.raises(ValueError):
        kv_cache_shape_ref(10, 24, 4, 128)
    bt = [7, 2, 9]
    assert slot_mapping_ref(bt, 5, 0, 4, True, 5) == [28, 29, 30, 31, 8]
    assert slot_mapping_ref(bt, 10, 9, 4, False, 1) == [9 * 4 + 1]
    assert slot_mapping_ref(None, 3, 0, 4, True, 3) == [-1, -1, -1]
    # prompt longer than the window: the leading tokens are never cached
    assert slot_mapping_ref(bt, 10, 0, 4, True, 10, sliding_window=6)[:4] == [-1] * 4


# ------------------------------------------------------------------ golden vectors
def test_golden_gla_decode():
    from oracle import gla_decode_step_ref

    z = np.load(os.path.join(GOLD, "gla_decode.npz"))
    state = bf(z["state0"]).clone()
    slot = torch.from_numpy(z["slot_idx"])
    for s in range(4):
        o = gla_decode_step_ref(bf(z[f"q{s}"]), bf(z[f"k{s}"]), bf(z[f"v{s}"]),
                                torch.from_numpy(z[f"g{s}"]), state, slot)
        assert torch.equal(o.view(torch.int16), bf(z[f"o{s}"]).view(torch.int16))
    assert torch.equal(state.view(torch.int16), bf(z["state_final"]).view(torch.int16))


def test_golden_reshape_and_cache():
    from oracle import reshape_and_cache_ref

    z = np.load(os.path.join(GOLD, "reshape_and_cache.npz"))
    kv = bf(z["kv_before"]).clone()
    reshape_and_cache_ref(bf(z["key"]), bf(z["value"]), kv[0], kv[1], torch.from_numpy(z["slot_mapping"]))
    assert torch.equal(kv.view(torch.int16), bf(z["kv_after"]).view(torch.int16))


def test_golden_paged_decode():
    from oracle import meta_swa_paged_decode_ref

    z = np.load(os.path.join(GOLD, "paged_decode.npz"))
    args = (bf(z["q"]), bf(z["key_cache"]), bf(z["value_cache"]), z["cache_seqlens"].tolist(),
            torch.from_numpy(z["block_tables"]))
    k2, v2 = bf(z["k2"]), bf(z["v2"])
    for name, kw in (("o_meta", dict(k2=k2, v2=v2, window_left=int(z["window_left_70b"]))),
                     ("o_full", dict(k2=k2, v2=v2, window_left=-1)),
                     ("o_7b", dict(window_left=int(z["window_left_7b"])))):
        out = meta_swa_paged_decode_ref(*args, **kw)
        assert np.allclose(out.float().numpy(), z[name], rtol=1e-6, atol=1e-7), name


def test_golden_swa_prefill():
    from oracle import meta_swa_varlen_ref

    z = np.load(os.path.join(GOLD, "swa_prefill.npz"))
    cu = z["cu_seqlens"]
    q, k, v = bf(z["q"]), bf(z["k"]), bf(z["v"])
    out = meta_swa_varlen_ref(q, k, v, cu, cu, bf(z["k2"]), bf(z["v2"]), bf(z["q2"]), int(z["window_left_70b"]))
    assert np.allclose(out.float().numpy(), z["o_meta_q2"], rtol=1e-6, atol=1e-7)
    out = meta_swa_varlen_ref(q, k, v, cu, cu, None, None, None, int(z["window_left_7b"]))
    assert np.allclose(out.float().numpy(), z["o_7b"], rtol=1e-6, atol=1e-7)


def test_golden_gla_prefill():
    from oracle import gla_prefill_varlen_ref
    from oracle.layers import gate_7b_ref

    z = np.load(os.path.join(GOLD, "gla_prefill.npz"))
    q7, k7, g7 = gate_7b_ref(bf(z["q"]), bf(z["k"]), bf(z["gk"]))
    st = bf(z["state0"]).clone()
    o = gla_prefill_varlen_ref(q7, k7, bf(z["v"]), g7, st, torch.from_numpy(z["slot_idx"]),
                               z["cu_seqlens"], z["context_lens"], None)
    assert torch.equal(o.view(torch.int16), bf(z["o_7b"]).view(torch.int16))
    assert torch.equal(st.view(torch.int16), bf(z["state_7b"]).view(torch.int16))
