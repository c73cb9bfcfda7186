"""Varlen sliding-window + meta-token prefill attention vs the dense-mask restatement, through the C ABI."""
import numpy as np
import os
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cu(lens):
    return torch.tensor(np.cumsum([0] + list(lens)), dtype=torch.int32)


@pytest.mark.parametrize("lens,H,Hkv,M,wl,with_q2", [
    ([5, 300], 4, 2, 32, 63, True),        # meta + window, warm-up call with meta queries
    ([5, 300], 4, 2, 32, 63, False),
    ([5, 300, 129], 7, 1, 128, -1, False), # 70B full-attention layer shard
    ([200, 1, 64], 28, 4, 0, 64, False),   # 7B SWA layer: no meta, W+1 keys
    ([1000], 4, 4, 128, 255, False),       # several 128-row blocks, window spanning blocks
])
def test_swa_prefill_unpaged(lens, H, Hkv, M, wl, with_q2):
    from oracle import meta_swa_varlen_ref
    from vllm_hymeta import hip_ops

    T = sum(lens)
    g_ = gen(0)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    k2 = randn_bf16(g_, M, Hkv, D) if M else None
    v2 = randn_bf16(g_, M, Hkv, D) if M else None
    q2 = randn_bf16(g_, M, H, D) if with_q2 else None
    cu = _cu(lens)
    ref = meta_swa_varlen_ref(q, k, v, cu, cu, k2, v2, q2, wl)
    dev = lambda t: None if t is None else t.cuda()
    out = hip_ops.meta_swa_prefill(q.cuda(), cu.cuda(), max(lens), key=k.cuda(), value=v.cuda(),
                                   cu_seqlens_k=cu.cuda(), meta_query=dev(q2), meta_key=dev(k2),
                                   meta_value=dev(v2), window_left=wl)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"swa prefill lens={lens} M={M} wl={wl} q2={with_q2}")


def test_swa_prefill_paged_chunked():
    """Chunked prefill: the chunk's queries attend through the block table to context + chunk,
    bottom-right aligned; rows of the cache beyond each sequence are poisoned."""
    from oracle.meta_swa import meta_swa_paged_prefill_ref
    from vllm_hymeta import hip_ops

    H, Hkv, bs, M, wl = 4, 2, 16, 32, 40
    seq_lens = [100, 37, 16]     # context + chunk
    q_lens = [30, 37, 1]
    g_ = gen(1)
    nb = [(L + bs - 1) // bs for L in seq_lens]
    nblk = sum(nb) + 2
    perm = torch.randperm(nblk, generator=g_)
    bt = torch.zeros(len(seq_lens), max(nb), dtype=torch.int32)
    off = 0
    for i, nbi in enumerate(nb):
        bt[i, :nbi] = perm[off:off + nbi].to(torch.int32)
        off += nbi
    kc, vc = randn_bf16(g_, nblk, bs, Hkv, D), randn_bf16(g_, nblk, bs, Hkv, D)
    q = randn_bf16(g_, sum(q_lens), H, D)
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    cuq = _cu(q_lens)
    ref = meta_swa_paged_prefill_ref(q, kc, vc, cuq, seq_lens, bt, k2, v2, wl)
    kc_bad, vc_bad = kc.clone(), vc.clone()
    live = torch.zeros(nblk, bs, dtype=torch.bool)
    for b, L in enumerate(seq_lens):
        for p in range(L):
            live[int(bt[b, p // bs]), p % bs] = True
    kc_bad[~live] = float("nan")
    vc_bad[~live] = float("nan")
    out = hip_ops.meta_swa_prefill(q.cuda(), cuq.cuda(), max(q_lens), key_cache=kc_bad.cuda(),
                                   value_cache=vc_bad.cuda(), block_tables=bt.cuda(),
                                   seq_lens_k=torch.tensor(seq_lens, dtype=torch.int32).cuda(),
                                   meta_key=k2.cuda(), meta_value=v2.cuda(), window_left=wl)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "paged chunked prefill")


def test_swa_prefill_golden():
    from vllm_hymeta import hip_ops

    z = np.load(os.path.join(GOLD, "swa_prefill.npz"))

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16).cuda()

    cu = torch.from_numpy(z["cu_seqlens"]).cuda()
    maxlen = int(np.diff(z["cu_seqlens"]).max())
    q, k, v, q2, k2, v2 = (bf(z[n]) for n in ("q", "k", "v", "q2", "k2", "v2"))
    cases = (("o_meta_q2", dict(meta_query=q2, meta_key=k2, meta_value=v2, window_left=int(z["window_left_70b"]))),
             ("o_meta", dict(meta_key=k2, meta_value=v2, window_left=int(z["window_left_70b"]))),
             ("o_full", dict(meta_key=k2, meta_value=v2, window_left=-1)),
             ("o_7b", dict(window_left=int(z["window_left_7b"]))))
    for name, kw in cases:
        out = hip_ops.meta_swa_prefill(q, cu, maxlen, key=k, value=v, cu_seqlens_k=cu, **kw)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), torch.from_numpy(z[name]), f"golden {name}")


def test_swa_prefill_softmax_spike():
    """A key far above the rest, placed late, forces the online-softmax rescale of earlier tiles."""
    from oracle import meta_swa_varlen_ref
    from vllm_hymeta import hip_ops

    H, Hkv, T = 2, 1, 300
    g_ = gen(2)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    k[250, 0] = (q[280, 1].float() * 3).to(torch.bfloat16)
    cu = _cu([T])
    ref = meta_swa_varlen_ref(q, k, v, cu, cu, None, None, None, -1)
    out = hip_ops.meta_swa_prefill(q.cuda(), cu.cuda(), T, key=k.cuda(), value=v.cuda(), cu_seqlens_k=cu.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "prefill spike")


@pytest.mark.parametrize("ramp", [1.5, 6.0, 40.0])
def test_swa_prefill_growing_scores(ramp):
    """Key norms grow along the sequence, so row maxima keep rising: slowly (the softmax reference
    lags behind the true maximum and no rescale happens), at the threshold, and fast (a rescale in
    most tiles, for some rows of a 32-query block and not for others)."""
    from oracle import meta_swa_varlen_ref
    from vllm_hymeta import hip_ops

    H, Hkv, lens = 4, 2, [700, 130]
    T = sum(lens)
    g_ = gen(5)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    pos = torch.cat([torch.arange(L) / L for L in lens])
    k = (k.float() * (1 + ramp * pos)[:, None, None]).to(torch.bfloat16)
    q[::3] = (q[::3].float() * 0.2).to(torch.bfloat16)          # every third query: flat softmax, no growth
    cu = _cu(lens)
    for wl in (-1, 200):
        ref = meta_swa_varlen_ref(q, k, v, cu, cu, None, None, None, wl)
        out = hip_ops.meta_swa_prefill(q.cuda(), cu.cuda(), max(lens), key=k.cuda(), value=v.cuda(), cu_seqlens_k=cu.cuda(),
                                       window_left=wl)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), ref, f"prefill, scores growing x{ramp}, window {wl}")


@pytest.mark.parametrize("splits", [2, 3, 8, 0])
@pytest.mark.parametrize("wl,M,with_q2", [(-1, 32, True), (200, 0, False), (63, 32, False)])
def test_swa_prefill_key_slices(splits, wl, M, with_q2):
    """A short chunk of long sequences with its visible keys cut into slices (partials + merge)
    equals the oracle like the single pass does; slices may be empty (short sequences, early rows)."""
    from oracle import meta_swa_varlen_ref
    from vllm_hymeta import hip_ops

    H, Hkv, lens = 2, 1, [700, 3, 130]
    T = sum(lens)
    g_ = gen(9)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    k2 = randn_bf16(g_, M, Hkv, D) if M else None
    v2 = randn_bf16(g_, M, Hkv, D) if M else None
    q2 = randn_bf16(g_, M, H, D) if with_q2 else None
    cu = _cu(lens)
    ref = meta_swa_varlen_ref(q, k, v, cu, cu, k2, v2, q2, wl)
    dev = lambda t: None if t is None else t.cuda()
    out = hip_ops.meta_swa_prefill(q.cuda(), cu.cuda(), max(lens), key=k.cuda(), value=v.cuda(), cu_seqlens_k=cu.cuda(),
                                   meta_query=dev(q2), meta_key=dev(k2), meta_value=dev(v2), window_left=wl,
                                   max_seqlen_k=max(lens), num_splits=splits)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"swa prefill, {splits} key slices, wl={wl} M={M}")


def test_swa_prefill_key_slices_paged_chunk():
    """Chunked prefill through the block table with the library's own choice of slices."""
    from oracle.meta_swa import meta_swa_paged_prefill_ref
    from vllm_hymeta import hip_ops

    H, Hkv, bs, M, wl = 2, 1, 16, 32, 1000
    seq_lens, q_lens = [1500], [100]
    g_ = gen(10)
    nblk = (seq_lens[0] + bs - 1) // bs
    bt = torch.randperm(nblk + 2, generator=g_)[:nblk].to(torch.int32).view(1, -1)
    kc, vc = randn_bf16(g_, nblk + 2, bs, Hkv, D), randn_bf16(g_, nblk + 2, bs, Hkv, D)
    q = randn_bf16(g_, q_lens[0], H, D)
    k2, v2 = randn_bf16(g_, M, Hkv, D), randn_bf16(g_, M, Hkv, D)
    cuq = _cu(q_lens)
    ref = meta_swa_paged_prefill_ref(q, kc, vc, cuq, seq_lens, bt, k2, v2, wl)
    lib = hip_ops._C.load()
    assert lib.hymeta_meta_swa_prefill_splits(q_lens[0], H, M, 0, wl, seq_lens[0]) == 2      # 20 tiles at most, 8 or more per slice
    out = hip_ops.meta_swa_prefill(q.cuda(), cuq.cuda(), q_lens[0], key_cache=kc.cuda(), value_cache=vc.cuda(),
                                   block_tables=bt.cuda(), seq_lens_k=torch.tensor(seq_lens, dtype=torch.int32).cuda(),
                                   meta_key=k2.cuda(), meta_value=v2.cuda(), window_left=wl, max_seqlen_k=seq_lens[0])
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "paged chunk, key slices")
