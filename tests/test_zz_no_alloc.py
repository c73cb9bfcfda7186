"""The allocation-free paged decode entry point (hymeta_meta_swa_paged_decode_static, split-KV partials in a
file-scope device array of the library) against hymeta_meta_swa_paged_decode on the same inputs: the two
run the same kernels, so the outputs must agree bit for bit; the CPU oracle checks both.  Two calls in a row
on one GPU, the second smaller than the first (the array keeps the first call's partials)."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu

D = 128


def _problem(seed, seq_lens, H, Hkv, bs, M):
    g_ = gen(seed)
    nb_per = [(L + bs - 1) // bs for L in seq_lens]
    nblk = sum(nb_per) + 2
    perm = torch.randperm(nblk, generator=g_)
    bt = torch.zeros(len(seq_lens), max(nb_per), dtype=torch.int32)
    off = 0
    for i, n in enumerate(nb_per):
        bt[i, :n] = perm[off:off + n].to(torch.int32)
        off += n
    kc, vc = randn_bf16(g_, nblk, bs, Hkv, D), randn_bf16(g_, nblk, bs, Hkv, D)
    q = randn_bf16(g_, len(seq_lens), H, D)
    k2 = randn_bf16(g_, M, Hkv, D) if M else None
    v2 = randn_bf16(g_, M, Hkv, D) if M else None
    return q, kc, vc, bt, k2, v2


def _run_pair(seed, seq_lens, H, Hkv, bs, M, window_left, splits):
    from oracle import meta_swa_paged_decode_ref
    from vllm_hymeta import hip_ops

    q, kc, vc, bt, k2, v2 = _problem(seed, seq_lens, H, Hkv, bs, M)
    lens = torch.tensor(seq_lens, dtype=torch.int32)
    dev = [None if t is None else t.cuda() for t in (q, kc, vc, lens, bt, k2, v2)]
    out_static = torch.full((len(seq_lens), H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    print(f"no-alloc paged decode: seed={seed} lens={seq_lens} H={H} Hkv={Hkv} bs={bs} M={M} wl={window_left} splits={splits}",
          flush=True)
    hip_ops.meta_swa_paged_decode(*dev, window_left=window_left, num_splits=splits, out=out_static, static_scratch=True)
    out_ws = hip_ops.meta_swa_paged_decode(*dev, window_left=window_left, num_splits=splits)
    torch.cuda.synchronize()
    assert torch.equal(out_static.view(torch.int16), out_ws.view(torch.int16)), "static-scratch output differs from the workspace form"
    ref = meta_swa_paged_decode_ref(q, kc, vc, seq_lens, bt, k2, v2, window_left)
    assert_bf16_close(out_static.cpu(), ref, "static-scratch paged decode vs oracle")


def test_static_scratch_two_calls_second_smaller():
    # 70B shard geometry with meta keys, 6 forced key splits; then fewer, shorter sequences with 3 splits
    _run_pair(71, [900, 64, 1, 333, 1200], 7, 1, 16, 32, 255, 6)
    _run_pair(72, [130, 40], 7, 1, 16, 32, 255, 3)


def test_static_scratch_all_heads_shape_and_library_split_choice():
    # 7B geometry, 4 kv-heads, block 256: the all-heads workgroup shape at 48+ sequences, splits chosen by the library
    _run_pair(73, [4200] * 2 + [700] * 46, 28, 4, 256, 0, 4096, 0)
    _run_pair(74, [300, 5, 77], 28, 4, 256, 0, 4096, 0)


def test_static_scratch_rejects_what_does_not_fit():
    from vllm_hymeta import _C, hip_ops

    B, H, Hkv, bs, splits = 1200, 28, 4, 16, 8      # 1200 * 28 * 8 * 130 * 4 B = 140 MB of partials
    assert B * H * splits * (D + 2) * 4 > _C.STATIC_SCRATCH_BYTES
    kc = torch.zeros(4, bs, Hkv, D, dtype=torch.bfloat16, device="cuda")
    q = torch.zeros(B, H, D, dtype=torch.bfloat16, device="cuda")
    lens = torch.full((B,), 40, dtype=torch.int32, device="cuda")
    bt = torch.zeros(B, 3, dtype=torch.int32, device="cuda")
    with pytest.raises(_C.HymetaError, match="scratch"):
        hip_ops.meta_swa_paged_decode(q, kc, kc, lens, bt, window_left=-1, num_splits=splits, static_scratch=True)
    # unsplit calls need no scratch and are never refused
    out = hip_ops.meta_swa_paged_decode(q, kc, kc, lens, bt, window_left=-1, num_splits=1, static_scratch=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
