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
    # 65 to 128 rows: eight row fragments per wave, one x image (two barriers per chunk)
    (128, 3584, 4608, True, 0),        # K slices
    (100, 3584, 37888, False, 0),      # whole tiles of two widths, rows past M masked
    (65, 18944, 3584, True, 0),        # deep K in slices
    (128, 896, 3584, False, 0),        # partial last chunk
    (96, 256, 64, True, 0),            # one tile, one chunk
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
    if M > 64:       # an explicit slice count selects the older kernel, which stops at 64 rows
        with pytest.raises(Exception):
            hip_ops.bf16_gemm(x.cuda(), w.cuda(), None if b is None else b.cuda(), split_k=1)
        return
    one = hip_ops.bf16_gemm(x.cuda(), w.cuda(), None if b is None else b.cuda(), split_k=1)
    torch.cuda.synchronize()
    assert_bf16_ulp(out, one, "sliced vs unsliced", ulps=1)


def test_bf16_gemm_strided_views_and_linear_dispatch():
    """Rows of a wider buffer as input (the fused qkv buffer), a column slice as output; HymetaLinear
    takes the kernel up to 128 rows and the library beyond, with the same result within rounding."""
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

    small, big = randn_bf16(g_, 128, 512).cuda(), randn_bf16(g_, 129, 512).cuda()
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


@pytest.mark.parametrize("kind", ["bf16", "w8a8"])
@pytest.mark.parametrize("M,K,I", [(64, 3584, 18944), (1, 3584, 18944), (33, 3584, 4736), (16, 320, 64), (40, 1024, 8256),
                                   (128, 3584, 18944), (70, 1024, 8256)])
def test_gate_up_projection_with_the_activation_in_its_epilogue(kind, M, K, I):
    """7B / 70B TP=4 gate_up shapes and small ones (one tile; two tile widths): the fused launch gives the bits of the
    linear followed by silu_and_mul (where that linear sums in the same order), and matches the fp64 restatement within the bf16 tolerance."""
    from oracle.quant import w8a8_gemm_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(M + K + I)
    x = randn_bf16(g_, M, K)
    if kind == "bf16":
        w = randn_bf16(g_, 2 * I, K, scale=K ** -0.5)
        lin = HymetaLinear.from_weight(w.cuda())
        y = (x.double() @ w.double().t()) if M * K * I < 1e9 else (x.float() @ w.float().t()).double()
    else:
        if K % 128:
            pytest.skip("int8 rows of K bytes: K must be a multiple of 128")
        wq = torch.randint(-127, 128, (2 * I, K), generator=g_, dtype=torch.int8)
        ws = torch.rand(2 * I, generator=g_) * 0.02 + 0.001
        lin = HymetaLinear.from_w8a8(wq.cuda(), ws.cuda())
        y = w8a8_gemm_ref(x, wq, ws, None)
    xd = x.cuda()
    assert hip_ops.gemm_silu_mul_supported(xd, lin.weight if kind == "bf16" else lin.qweight)
    fused = lin.forward_silu_mul(xd)
    two_step = hip_ops.silu_and_mul(lin(xd))
    torch.cuda.synchronize()
    assert fused.shape == (M, I)
    if I == 18944 or kind == "w8a8":
        # the plain linear runs these shapes without K slices (or sums exact integers): the same bits
        assert torch.equal(fused, two_step)
    else:
        # K slices change the order of the fp32 sums: a factor may round to the neighbouring bf16 value
        assert_bf16_ulp(fused, two_step, "fused vs linear + silu_and_mul", ulps=4)
    yb = y.to(torch.bfloat16).double()
    ref = torch.nn.functional.silu(yb[:, :I]) * yb[:, I:]
    assert_bf16_close(fused.cpu(), ref, f"gated {kind} M={M} K={K} I={I}", rel=2e-2, frac=2.0 ** -6)


@pytest.mark.parametrize("kind", ["bf16", "w8a8"])
@pytest.mark.parametrize("M,K,N,bias", [(64, 3584, 3584, False), (33, 18944, 3584, True), (17, 896, 3584, False), (128, 18944, 3584, True)])
def test_k_slice_sum_left_to_the_fused_add_rmsnorm(kind, M, K, N, bias):
    """o_proj / down_proj shapes: the linear stops at its K slices' partial sums and the residual add + RMS norm adds
    them while reading; normed output, updated residual and the materialised linear equal the plain calls bit for bit."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(M + K + N)
    x = randn_bf16(g_, M, K).cuda()
    b = randn_bf16(g_, N).cuda() if bias else None
    if kind == "bf16":
        lin = HymetaLinear.from_weight(randn_bf16(g_, N, K, scale=K ** -0.5).cuda(), b)
    else:
        if K % 128:
            pytest.skip("int8 rows of K bytes: K must be a multiple of 128")
        lin = HymetaLinear.from_w8a8(torch.randint(-127, 128, (N, K), generator=g_, dtype=torch.int8).cuda(),
                                     (torch.rand(N, generator=g_) * 0.02 + 0.001).cuda(), b)
    w = (1 + 0.1 * torch.randn(N, generator=g_)).to(torch.bfloat16).cuda()
    res = randn_bf16(g_, M, N).cuda()
    y = lin(x)
    r1 = res.clone()
    out1 = hip_ops.add_rmsnorm(y, w, 1e-6, residual=r1)
    part = lin(x, defer=True)
    if kind == "w8a8" and K < 1024:
        # two chunks of int8 K: one slice, nothing to defer -- the call returns the finished tensor
        assert torch.equal(part, y)
        return
    assert isinstance(part, hip_ops.PartialSums) and part.shape == (M, N), "these shapes are cut into K slices"
    r2 = res.clone()
    out2 = hip_ops.add_rmsnorm(part, w, 1e-6, residual=r2)
    again = lin(x, defer=True).tensor()
    torch.cuda.synchronize()
    assert torch.equal(out1, out2) and torch.equal(r1, r2) and torch.equal(again, y)
    # partial sums overwritten by the next linear of the kind are refused, not read
    stale = lin(x, defer=True)
    lin(x)
    with pytest.raises(ValueError):
        hip_ops.add_rmsnorm(stale, w, 1e-6, residual=r2)


@pytest.mark.parametrize("kind", ["bf16", "w8a8"])
def test_qkv_projection_finished_by_rope_from_its_partial_sums(kind):
    """7B sliding-window layer: the K-sliced qkv projection's slice sum, bias and the rotation of q and k in one launch
    equal the plain projection followed by the in-place RoPE, bit for bit."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.window_attention import rope_cos_sin_cache

    g_ = gen(31)
    T, H, Hkv, D, hidden = 37, 28, 4, 128, 3584
    N = (H + 2 * Hkv) * D
    x = randn_bf16(g_, T, hidden).cuda()
    b = randn_bf16(g_, N).cuda()
    if kind == "bf16":
        lin = HymetaLinear.from_weight(randn_bf16(g_, N, hidden, scale=hidden ** -0.5).cuda(), b)
    else:
        lin = HymetaLinear.from_w8a8(torch.randint(-127, 128, (N, hidden), generator=g_, dtype=torch.int8).cuda(),
                                     (torch.rand(N, generator=g_) * 0.02 + 0.001).cuda(), b)
    table = rope_cos_sin_cache(D, 4096, 1e6, device=x.device)
    pos = torch.randint(0, 4096, (T,), generator=g_).cuda()
    ref = lin(x)
    hip_ops.rope_neox_(pos, ref[:, : H * D].view(T, H, D), ref[:, H * D:(H + Hkv) * D].view(T, Hkv, D), table)
    part = lin(x, defer=True)
    assert isinstance(part, hip_ops.PartialSums)
    got = hip_ops.rope_from_partials(part, pos, table, H, Hkv)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("M,K,N", [(1, 512, 1024), (20, 3584, 4096), (64, 3584, 151936), (130, 1024, 2048 + 64), (200, 3584, 151936),
                                   (300, 1024, 40960)])
def test_bf16_logits_are_the_fp32_product_of_the_widened_operands(M, K, N):
    """lm_head kept in bf16: x W^T with fp32 accumulation and an unrounded fp32 result, against the fp64 product of the
    same bf16 values (what lm_head.float() on hidden_states.float() computes, up to fp32 summation error) -- also more
    than 128 rows (128 at a time, or 256 where the 129..256-row kernel takes the matrix: the wide ones) and the 7B
    vocabulary; then the greedy token of each row."""
    from vllm_hymeta import hip_ops

    g_ = gen(41)
    x = randn_bf16(g_, M, K).cuda()
    w = randn_bf16(g_, N, K, scale=K ** -0.5).cuda()
    got = hip_ops.bf16_logits(x, w)
    assert got.dtype == torch.float32 and got.shape == (M, N)
    # the fp64 product of the large cases on a few rows: the first ones, one of every later 64, the last
    rows = torch.arange(M) if N <= 8192 else torch.tensor(sorted({0, 1, 2, 3, M - 1} | set(range(70, M, 64))))
    rows = rows.cuda()
    ref = x[rows].double() @ w.double().t()
    err = (got[rows].double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err
    # fp32 tolerance: bf16-rounded logits would be off by 2^-9 relative, far outside it
    tok = hip_ops.argmax_rows(got)
    torch.cuda.synchronize()
    assert torch.equal(tok, got.argmax(-1))
    ties = torch.zeros(3, 1024, device="cuda")
    ties[0, 700] = ties[0, 9] = 5.0
    ties[1, :] = -1.0
    ties[2, 1023] = 1.0
    assert hip_ops.argmax_rows(ties).tolist() == [9, 0, 1023]
