"""The 129..256-row weight-streaming GEMM (tall_gemm.hip) behind hymeta_bf16_gemm / hymeta_w8a8_gemm: bf16 against an
fp64 product of the same operands; int8 bit for bit against the decode-sized kernel run on the rows in two batches (exact
int32 sums, the same scaling) and against the CPU restatement."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N,bias", [
    (256, 18944, 3584, True),        # 7B down projection: 28 tiles of 8 fragments x 9 K slices
    (129, 18944, 3584, False),       # three row fragments per wave, all but one row of the last fragment masked
    (193, 9472, 2048 + 64, True),    # four row fragments per wave; 132 fragments in 17 tiles of 7 or 8
    (200, 18944, 1024, False),       # 8 tiles x 16 slices
    (256, 3584, 4608, True),         # 7B qkv: 36 tiles x 7 slices of 8 pairs
    (144, 3584, 37888, False),       # 256 whole tiles of 9 or 10 fragments, no slices
])
def test_bf16_rows_129_to_256(M, K, N, bias):
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(M + K + N)
    x, w = randn_bf16(g_, M, K).cuda(), randn_bf16(g_, N, K, scale=K ** -0.5).cuda()
    b = randn_bf16(g_, N).cuda() if bias else None
    assert hip_ops.bf16_gemm_supported(x, w), "a shape the 129..256-row kernel takes"
    ref = (x.double() @ w.double().t()).cpu()
    if b is not None:
        ref = ref + b.double().cpu()
    out = hip_ops.bf16_gemm(x, w, b)
    again = hip_ops.bf16_gemm(x, w, b)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"bf16 gemm M={M} K={K} N={N}")
    assert torch.equal(out, again), "slice sums are taken in slice order: same bits on every run"
    # rows of a wider buffer, output into columns of a wider one
    xw = torch.zeros(M, K + 64, dtype=torch.bfloat16, device="cuda")
    xw[:, :K] = x
    ow = torch.full((M, N + 128), 7.0, dtype=torch.bfloat16, device="cuda")
    hip_ops.bf16_gemm(xw[:, :K], w, b, out=ow[:, 64:64 + N])
    assert torch.equal(ow[:, 64:64 + N], out) and bool((ow[:, :64] == 7).all()) and bool((ow[:, 64 + N:] == 7).all())
    # the K slices left to the fused add + RMS norm
    lin = HymetaLinear.from_weight(w, b)
    nw = (1 + 0.1 * torch.randn(N, generator=g_)).to(torch.bfloat16).cuda()
    res = randn_bf16(g_, M, N).cuda()
    r1, r2 = res.clone(), res.clone()
    out1 = hip_ops.add_rmsnorm(lin(x), nw, 1e-6, residual=r1)
    part = lin(x, defer=True)
    if N >= 16384:
        assert torch.equal(part, out), "whole tiles: nothing to defer"
        return
    assert isinstance(part, hip_ops.PartialSums) and part.shape == (M, N)
    out2 = hip_ops.add_rmsnorm(part, nw, 1e-6, residual=r2)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2) and torch.equal(r1, r2)


@pytest.mark.parametrize("M,K,N,bias", [
    (256, 3584, 37888, False),       # 7B gate_up: 256 tiles of 9 or 10 fragments, whole K
    (130, 3584, 37888, True),
    (256, 18944, 3584, True),        # K slices
    (192, 3584, 4608, True),         # three row fragments per wave, exactly
    (201, 4736, 3584, False),        # 37 pairs in 9 slices of 4 or 5
    (160, 3584, 4608 + 64, False),   # tiles of 7 or 8 fragments
    (256, 512, 1024, False),         # 4 pairs: one slice
])
def test_w8a8_rows_129_to_256_bit_for_bit(M, K, N, bias):
    from oracle.quant import w8a8_gemm_ref
    from vllm_hymeta import hip_ops

    g_ = gen(M + K + N)
    x = randn_bf16(g_, M, K).cuda()
    qw = torch.randint(-127, 128, (N, K), generator=g_, dtype=torch.int8).cuda()
    ws = (torch.rand(N, generator=g_) * 0.01 + 0.001).cuda()
    b = randn_bf16(g_, N).cuda() if bias else None
    out = hip_ops.w8a8_gemm(x, qw, ws, bias=b)
    halves = torch.cat([hip_ops.w8a8_gemm(x[:128], qw, ws, bias=b), hip_ops.w8a8_gemm(x[128:], qw, ws, bias=b)])
    torch.cuda.synchronize()
    assert torch.equal(out, halves), "exact integer sums and one scaling: the row count does not change a bit"
    if M * K * N < 4e9:
        ref = w8a8_gemm_ref(x.cpu(), qw.cpu(), ws.cpu(), None if b is None else b.cpu())
        assert_bf16_close(out.cpu(), ref, f"w8a8 gemm M={M} K={K} N={N}")


@pytest.mark.parametrize("kind", ["bf16", "w8a8"])
@pytest.mark.parametrize("M,K,I", [(256, 3584, 18944), (150, 3584, 18944), (200, 1024, 4736)])
def test_gate_up_with_the_activation_in_the_epilogue_above_128_rows(kind, M, K, I):
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(M + K + I)
    x = randn_bf16(g_, M, K).cuda()
    if kind == "bf16":
        lin = HymetaLinear.from_weight(randn_bf16(g_, 2 * I, K, scale=K ** -0.5).cuda())
        assert hip_ops.gemm_silu_mul_supported(x, lin.weight) == (I == 18944), "narrow ones: too few tiles, a library GEMM"
    else:
        lin = HymetaLinear.from_w8a8(torch.randint(-127, 128, (2 * I, K), generator=g_, dtype=torch.int8).cuda(),
                                     (torch.rand(2 * I, generator=g_) * 0.01 + 0.001).cuda())
    fused = lin.forward_silu_mul(x)
    two = hip_ops.silu_and_mul(lin(x))
    torch.cuda.synchronize()
    assert fused.shape == (M, I) and torch.equal(fused, two)


@pytest.mark.parametrize("bits,group,scales_dtype,zero_offset", [(4, 128, torch.float16, 1), (4, 64, torch.bfloat16, 0),
                                                                 (8, 64, torch.float16, 1), (8, 128, torch.bfloat16, 0)])
@pytest.mark.parametrize("M,K,N,bias", [
    (256, 3584, 4608, True),         # K slices
    (200, 18944, 3584, False),       # deep K: 296 pairs in 9 slices
    (130, 3584, 37888, False),       # whole tiles of 9 or 10 fragments
    (193, 1024, 4608 + 64, True),    # tiles of 7 or 8 fragments, 16 pairs
    (100, 3584, 4608, True),         # 8-bit weights take this kernel from 65 rows on (two row fragments per wave)
    (65, 18944, 3584, False),
])
def test_gptq_rows_129_to_256(M, K, N, bias, bits, group, scales_dtype, zero_offset):
    """4- and 8-bit GPTQ linears at 129..256 rows: the packed words are expanded into the kernel's weight image as
    bf16(scale * (q - zero)) -- the values of the whole-matrix expansion -- so the result equals the bf16 kernel run on
    the expanded matrix bit for bit (same tiles, same order of the sums), and the fp64 restatement within tolerance."""
    from oracle.quant import gptq_gemm_ref
    from test_gpu_quant import _gptq_problem
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    qw, qz, scales = _gptq_problem(M + K + N, K, N, bits, group, scales_dtype)
    g_ = gen(M + N)
    x = randn_bf16(g_, M, K)
    b = randn_bf16(g_, N) if bias else None
    lin = HymetaLinear.from_gptq(qw.cuda(), qz.cuda(), scales.cuda(), bits, group, bias=None if b is None else b.cuda(), zero_offset=zero_offset)
    xd = x.cuda()
    y, again = lin(xd), lin(xd)
    torch.cuda.synchronize()
    assert torch.equal(y, again)
    if M > 128:
        w_t = hip_ops.gptq_dequant(lin.qweight, lin.qzeros, lin.scales, bits, group, zero_offset=zero_offset).t().contiguous()      # [N, K]
        assert hip_ops.bf16_gemm_supported(xd, w_t)
        same_tiles = hip_ops.bf16_gemm(xd, w_t, lin.bias)
        torch.cuda.synchronize()
        assert torch.equal(y, same_tiles), "the expanded pair in LDS holds the values of the expanded matrix"
    if M * K * N < 6e9:
        ref = gptq_gemm_ref(x, qw, qz, scales, bits, group, b, zero_offset=zero_offset)
        assert_bf16_close(y.cpu(), ref, f"gptq{bits} M={M} K={K} N={N}", rel=8e-3)
    if N < 16384 and (M > 128 or bits == 8):
        part = lin(xd, defer=True)
        assert isinstance(part, hip_ops.PartialSums), "cut into K slices"
        nw = (1 + 0.1 * torch.randn(N, generator=g_)).to(torch.bfloat16).cuda()
        res = randn_bf16(g_, M, N).cuda()
        r1, r2 = res.clone(), res.clone()
        out1 = hip_ops.add_rmsnorm(y, nw, 1e-6, residual=r1)
        out2 = hip_ops.add_rmsnorm(part, nw, 1e-6, residual=r2)
        torch.cuda.synchronize()
        assert torch.equal(out1, out2) and torch.equal(r1, r2)
