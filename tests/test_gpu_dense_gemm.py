"""bf16 linear for decode-sized batches (hymeta_bf16_gemm) vs an fp64 / fp32 matmul on the same
bf16 inputs; the K-slice decomposition is deterministic and does not change the result beyond
fp32 summation order."""
import pytest
import torch

from _util import assert_bf16_close, assert_bf16_ulp, gen, randn_bf16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N,bias,split_k", [
    (1, 256, 128, True, 1),
    (1, 3584, 4608, True, 0),          # 7B qkv projection, one token
    (7, 3584, 3584, False, 0),
    (16, 512, 64, True, 3),            # slices that do not divide the k-steps evenly
    (17, 4736, 3584, False, 0),        # 70B TP=4 down projection: K = 18.5 chunks of 256 (partial last chunk)
    (64, 896, 3584, True, 0),          # 70B TP=4 out_proj: 3.5 chunks, K slices that end inside a chunk
    (64, 3584, 9472, False, 0),        # 70B TP=4 gate_up: 148 column groups of 64 -> tiles of two widths... one width here
    (40, 320, 4160, True, 0),          # 65 groups < 256 workgroups, 1.25 chunks
    (64, 3584, 18944, False, 0),       # 296 groups: 256 tiles, 40 of two fragments per wave and 216 of one
    (33, 18944, 3584, False, 0),       # 7B down projection: few column tiles, deep K
    (64, 3584, 1024, True, 5),
    (64, 1024, 4096, True, 1),
    # the weight-streaming kernel (library's choice, K a multiple of 64): whole tiles of two widths, K slices + the reduce launch
    (64, 3584, 4608, True, 0),         # 7B qkv at the bench batch: 128-column tiles, K slices
    (48, 3584, 37888, False, 0),       # 7B gate_up: 592 groups -> 256 tiles, 80 of 192 columns and 176 of 128; rows past M masked
    (64, 18944, 3584, True, 0),        # 7B down projection: deep K in slices
    (5, 512, 320, True, 0),            # 64-column tiles (N not a multiple of 128), two chunks
    (20, 256, 64, False, 0),           # one tile, one chunk
])
def test_bf16_gemm(M, K, N, bias, split_k):
    from vllm_hymeta import hip_ops

    g_ = gen(M + K + N)
    x, w = randn_bf16(g_, M, K), randn_bf16(g_, N, K, scale=K ** -0.5)
    b = randn_bf16(g_, N) if bias else None
    ref = x.double() @ w.double().t() if M * K * N < 2e9 else (x.float() @ w.float().t()).double()
    if b is not None:
        ref = ref + b.double()
    out = hip_ops.bf16_gemm(x.cuda(), w.cuda(), None if b is None else b.cuda(), split_k=split_k)
    again = hip_ops.bf16_gemm(x.cuda(), w.cuda(), None if b is None else b.cuda(), split_k=split_k)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"bf16 gemm M={M} K={K} N={N} split={split_k}")
    assert torch.equal(out, again), "slice sums are taken in slice order: same bits on every run"
    one = hip_ops.bf16_gemm(x.cuda(), w.cuda(), None if b is None else b.cuda(), split_k=1)
    torch.cuda.synchronize()
    assert_bf16_ulp(out, one, "sliced vs unsliced", ulps=1)


def test_bf16_gemm_strided_views_and_linear_dispatch():
    """Rows of a wider buffer as input (the fused qkv buffer), a column slice as output; HymetaLinear
    takes the kernel up to 64 rows and the library beyond, with the same result within rounding."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(3)
    buf = randn_bf16(g_, 20, 1024).cuda()
    w, b = randn_bf16(g_, 256, 512, scale=512 ** -0.5).cuda(), randn_bf16(g_, 256).cuda()
    x = buf[:, 256:768]
    out_buf = torch.zeros(20, 640, dtype=torch.bfloat16, device="cuda")
    hip_ops.bf16_gemm(x, w, b, out=out_buf[:, 128:384])
    torch.cuda.synchronize()
    ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    assert_bf16_close(out_buf[:, 128:384].cpu(), ref, "strided bf16 gemm")
    assert float(out_buf[:, :128].abs().max()) == 0 and float(out_buf[:, 384:].abs().max()) == 0

    small, big = randn_bf16(g_, 64, 512).cuda(), randn_bf16(g_, 65, 512).cuda()
    assert hip_ops.bf16_gemm_supported(small, w) and not hip_ops.bf16_gemm_supported(big, w)
    deep = randn_bf16(g_, 128, 512, scale=512 ** -0.5).cuda()          # K = 4 N
    assert hip_ops.bf16_gemm_preferred(small, w) and hip_ops.bf16_gemm_preferred(small, deep), "every decode-sized linear runs on the kernel"
    assert not hip_ops.bf16_gemm_preferred(big, w)
    for wt, bias in ((w, b), (deep, None)):
        lin = HymetaLinear.from_weight(wt, bias)
        for xin in (small, big):
            ref = xin.double().cpu() @ wt.double().cpu().t() + (0 if bias is None else bias.double().cpu())
            assert_bf16_close(lin(xin).cpu(), ref, f"linear {tuple(wt.shape)}, {xin.shape[0]} rows")
    with pytest.raises(Exception):
        hip_ops.bf16_gemm(big, w, b)
