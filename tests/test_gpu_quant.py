"""Fused dequant-GEMMs (GPTQ 4/8-bit, W8A8) vs the CPU restatement of the tensor formats."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu


def _gptq_problem(seed, K, N, bits, group, scales_dtype):
    g_ = gen(seed)
    from oracle.quant import gptq_pack_ref

    w_int = torch.randint(0, 2 ** bits, (K, N), generator=g_)
    zeros = torch.randint(1, 2 ** bits, (K // group, N), generator=g_)   # stored as z - 1 >= 0
    scales = (torch.rand(K // group, N, generator=g_) * 0.02 + 0.005).to(scales_dtype)
    qw, qz = gptq_pack_ref(w_int, zeros, bits)
    return qw, qz, scales


@pytest.mark.parametrize("bits,group,M,K,N,scales_dtype,split_k", [
    (4, 128, 1, 512, 256, torch.float16, 0),
    (4, 64, 7, 1024, 128, torch.float16, 1),
    (4, 64, 16, 1024, 128, torch.bfloat16, 4),
    (8, 64, 3, 512, 192, torch.float16, 0),      # the reference recipe: 8-bit, group 64
    (8, 64, 40, 768, 64, torch.float16, 2),
    (4, 128, 100, 512, 128, torch.float16, 1),   # several 64-row passes
    (4, 32, 5, 256, 64, torch.float16, 0),
    # wider outputs, odd group counts per K slice, ragged rows, two row blocks
    (4, 128, 64, 1024, 512, torch.float16, 0),
    (4, 128, 33, 640, 256, torch.bfloat16, 3),   # odd group count per slice, ragged rows
    (4, 64, 17, 512, 256, torch.float16, 2),
    (8, 64, 50, 512, 512, torch.float16, 1),
    (4, 32, 100, 256, 256, torch.float16, 4),    # two row blocks
])
def test_gptq_gemm(bits, group, M, K, N, scales_dtype, split_k):
    from oracle.quant import gptq_gemm_ref
    from vllm_hymeta import hip_ops

    qw, qz, scales = _gptq_problem(0, K, N, bits, group, scales_dtype)
    g_ = gen(1)
    x = randn_bf16(g_, M, K)
    bias = randn_bf16(g_, N)
    ref = gptq_gemm_ref(x, qw, qz, scales, bits, group, bias)
    out = hip_ops.gptq_gemm(x.cuda(), qw.cuda(), qz.cuda(), scales.cuda(), bits, group, bias=bias.cuda(),
                            split_k=split_k)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"gptq gemm bits={bits} g={group} M={M} split={split_k}")


def test_gptq_gemm_7b_qkv_shape_and_strided_input():
    """qkv projection of the 7B layer (3584 -> 4608), input rows taken from a wider buffer."""
    from oracle.quant import gptq_gemm_ref
    from vllm_hymeta import hip_ops

    K, N, M = 3584, 4608, 4
    qw, qz, scales = _gptq_problem(2, K, N, 4, 128, torch.float16)
    xbuf = randn_bf16(gen(3), M, K + 64)
    x = xbuf[:, :K]
    ref = gptq_gemm_ref(x, qw, qz, scales, 4, 128)
    out = hip_ops.gptq_gemm(xbuf.cuda()[:, :K], qw.cuda(), qz.cuda(), scales.cuda(), 4, 128)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "gptq qkv 7B")


@pytest.mark.parametrize("M,K,N", [(1, 256, 64), (9, 1024, 192), (70, 512, 128), (4, 3584, 4608),
                                   (33, 256, 128), (64, 3584, 4608), (50, 768, 256), (200, 3584, 3584), (64, 18944, 128)])
def test_w8a8_gemm(M, K, N):
    from oracle.quant import w8a8_gemm_ref
    from vllm_hymeta import hip_ops

    g_ = gen(4)
    x = randn_bf16(g_, M, K)
    wq = torch.randint(-127, 128, (N, K), generator=g_).to(torch.int8)
    ws = torch.rand(N, generator=g_) * 0.01 + 0.001
    bias = randn_bf16(g_, N)
    ref = w8a8_gemm_ref(x, wq, ws, bias)
    out = hip_ops.w8a8_gemm(x.cuda(), wq.cuda(), ws.cuda(), bias=bias.cuda())
    torch.cuda.synchronize()
    # a product x/s that lands within fp32 rounding of .5 may quantise one step differently
    assert_bf16_close(out.cpu(), ref, f"w8a8 gemm M={M} K={K} N={N}", rel=6e-3)


def test_quant_gemm_rejects_unsupported():
    from vllm_hymeta import hip_ops

    x = torch.zeros(2, 128, dtype=torch.bfloat16, device="cuda")
    qw = torch.zeros(16, 48, dtype=torch.int32, device="cuda")      # N not a multiple of 64
    qz = torch.zeros(1, 6, dtype=torch.int32, device="cuda")
    sc = torch.ones(1, 48, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError):
        hip_ops.gptq_gemm(x, qw, qz, sc, 4, 128)


@pytest.mark.parametrize("bits,group,scales_dtype", [(4, 128, torch.float16), (8, 64, torch.float16), (4, 32, torch.bfloat16)])
def test_gptq_dequant_is_the_rounded_oracle_matrix(bits, group, scales_dtype):
    """Whole-matrix dequantisation: bit-exact bf16 rounding of s * (q - z) evaluated in fp32."""
    from oracle.quant import gptq_dequant_ref
    from vllm_hymeta import hip_ops

    K, N = 512, 192
    qw, qz, scales = _gptq_problem(5, K, N, bits, group, scales_dtype)
    w = hip_ops.gptq_dequant(qw.cuda(), qz.cuda(), scales.cuda(), bits, group)
    torch.cuda.synchronize()
    ref = gptq_dequant_ref(qw, qz, scales, bits, group).float().to(torch.bfloat16)   # s and (q - z) are exact in fp32
    assert torch.equal(w.cpu(), ref)
    # the transposed form (torch Linear layout, what the prefill-sized linear feeds the library GEMM): the same bits
    assert hip_ops.gptq_dequant_nk_supported(qw, bits)
    wt = hip_ops.gptq_dequant(qw.cuda(), qz.cuda(), scales.cuda(), bits, group, transposed=True)
    torch.cuda.synchronize()
    assert wt.shape == (N, K) and torch.equal(wt.cpu(), ref.t())


def test_gptq_transposed_expansion_covers_whole_tiles_only():
    """64 columns x 32 packed rows a tile: other shapes are refused by the entry point and the linear keeps the [K, N]
    expansion for them (same result either way)."""
    from oracle.quant import gptq_gemm_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta import linear as L

    K, N = 384, 96                                 # K % 256 != 0 and N % 64 != 0
    qw, qz, scales = _gptq_problem(9, K, N, 4, 128, torch.float16)
    assert not hip_ops.gptq_dequant_nk_supported(qw, 4)
    with pytest.raises(ValueError):
        hip_ops.gptq_dequant(qw.cuda(), qz.cuda(), scales.cuda(), 4, 128, transposed=True)
    out_nk = torch.empty(N, K, dtype=torch.bfloat16, device="cuda")
    from vllm_hymeta import _C
    assert _C.load().hymeta_gptq_dequant_nk(qw.cuda().data_ptr(), qz.cuda().data_ptr(), scales.cuda().data_ptr(), out_nk.data_ptr(),
                                            K, N, 4, 128, _C.HYMETA_F16, 1, None) == -2
    lin = L.HymetaLinear.from_gptq(qw.cuda(), qz.cuda(), scales.cuda(), 4, 128)
    x = randn_bf16(gen(10), L.GPTQ_EXPAND_ROWS + 3, K)
    out = lin(x.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), gptq_gemm_ref(x, qw, qz, scales, 4, 128, None), "gptq [K, N] expansion", rel=8e-3)


@pytest.mark.parametrize("bits,group,scales_dtype,zero_offset", [(4, 128, torch.float16, 1), (4, 128, torch.bfloat16, 0),
                                                                 (8, 64, torch.float16, 1), (8, 64, torch.bfloat16, 0),
                                                                 (4, 64, torch.float16, 1)])
@pytest.mark.parametrize("M,K,N,bias", [(64, 3584, 4608, True), (33, 18944, 3584, False), (1, 3584, 3584, False), (16, 256, 128, True),
                                        (20, 1024, 640, False), (24, 3584, 37888, False)])
def test_gptq_streaming_kernel_and_deferred_slice_sum(M, K, N, bias, bits, group, scales_dtype, zero_offset):
    """Group-128 4-bit and group-64 8-bit linears (the two recipes of the reference; 4-bit at group 64 as well) with few columns (qkv / o / down shapes;
    one tile; five 32-column blocks per tile) or few rows (gate_up) run on the weight-streaming kernel: against the fp64
    restatement, deterministic, and with the K slices' sum left to the fused add + RMS norm the same bits as the plain call
    followed by the norm."""
    from oracle.quant import gptq_gemm_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    qw, qz, scales = _gptq_problem(M + K + N, K, N, bits, group, scales_dtype)
    g_ = gen(M + N)
    x = randn_bf16(g_, M, K)
    b = randn_bf16(g_, N) if bias else None
    ref = gptq_gemm_ref(x, qw, qz, scales, bits, group, b, zero_offset=zero_offset)
    lin = HymetaLinear.from_gptq(qw.cuda(), qz.cuda(), scales.cuda(), bits, group, bias=None if b is None else b.cuda(), zero_offset=zero_offset)
    xd = x.cuda()
    y, again = lin(xd), lin(xd)
    torch.cuda.synchronize()
    assert_bf16_close(y.cpu(), ref, f"gptq{bits} streaming M={M} K={K} N={N}", rel=8e-3)
    assert torch.equal(y, again)
    part = lin(xd, defer=True)
    w = (1 + 0.1 * torch.randn(N, generator=g_)).to(torch.bfloat16).cuda()
    res = randn_bf16(g_, M, N).cuda()
    r1, r2 = res.clone(), res.clone()
    out1 = hip_ops.add_rmsnorm(y, w, 1e-6, residual=r1)
    if K >= 1024 and N < 16384:
        assert isinstance(part, hip_ops.PartialSums), "cut into K slices"
    out2 = hip_ops.add_rmsnorm(part, w, 1e-6, residual=r2)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2) and torch.equal(r1, r2)


def test_gptq_linear_large_m_uses_expanded_weights():
    """Above the row threshold the linear expands its weights and runs the library GEMM: same
    result as the oracle within the bf16 tolerance, and equal code path selection on both sides
    of the threshold."""
    from oracle.quant import gptq_gemm_ref
    from vllm_hymeta import linear as L

    K, N, bits, group = 512, 256, 4, 128
    qw, qz, scales = _gptq_problem(6, K, N, bits, group, torch.float16)
    g_ = gen(7)
    bias = randn_bf16(g_, N)
    lin = L.HymetaLinear.from_gptq(qw.cuda(), qz.cuda(), scales.cuda(), bits, group, bias=bias.cuda())
    for M in (L.GPTQ_EXPAND_ROWS, L.GPTQ_EXPAND_ROWS + 44):
        x = randn_bf16(g_, M, K)
        ref = gptq_gemm_ref(x, qw, qz, scales, bits, group, bias)
        out = lin(x.cuda())
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), ref, f"gptq linear M={M}", rel=8e-3)


@pytest.mark.parametrize("M,K,N", [(17, 3584, 4608), (33, 3584, 37888), (64, 18944, 3584), (64, 3584, 3584), (40, 512, 320),
                                   (64, 4736, 3584), (24, 896, 3584), (64, 3584, 9472),      # these three: 70B TP=4 shards
                                   (128, 3584, 4608), (100, 3584, 37888), (65, 18944, 3584), (1, 3584, 3584), (9, 18944, 3584)])
def test_w8a8_streaming_kernel_matches_the_library_path_exactly(M, K, N):
    """Up to 128 rows run on the weight-streaming int8 kernel (whole tiles or K slices + the reduce launch); the int32 sums
    are exact, so the result equals quantise -> library int8 GEMM -> scale bit for bit, and the oracle within tolerance."""
    from oracle.quant import w8a8_gemm_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear, w8a8_use_library

    assert not w8a8_use_library(M, K, N)
    g_ = gen(23)
    x = randn_bf16(g_, M, K)
    wq = torch.randint(-127, 128, (N, K), generator=g_, dtype=torch.int8)
    ws = torch.rand(N, generator=g_) * 0.02 + 0.001
    bias = randn_bf16(g_, N, scale=0.1)
    lin = HymetaLinear.from_w8a8(wq.cuda(), ws.cuda(), bias.cuda())
    out = lin(x.cuda())
    xq, xs = hip_ops.w8a8_quantize(x.cuda())
    lib = hip_ops.w8a8_scale_out(torch._int_mm(xq, wq.cuda().t()) if M > 32 else (xq.int().cpu() @ wq.int().t()).cuda().int(),
                                 xs, ws.cuda(), bias=bias.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out, lib)
    assert_bf16_close(out.cpu(), w8a8_gemm_ref(x, wq, ws, bias), f"w8a8 streaming M={M}", rel=6e-3)


@pytest.mark.parametrize("M,K,N", [(300, 512, 256), (1024, 3584, 4608), (129, 3584, 4608), (200, 2048, 256), (513, 3584, 3584),
                                   (8192, 3584, 4608), (1100, 18944, 256), (129, 3584, 37888), (300, 1024, 16384), (130, 320, 192)])
def test_w8a8_linear_prefill_regime_both_paths_give_the_same_bits(M, K, N):
    """Above 128 rows a W8A8 linear is the tiled int8-MFMA kernel (quantisation launch, then ONE GEMM launch with the
    scales and the bias in its epilogue) or quantise -> library int8 GEMM -> scale, whichever ``w8a8_use_library`` picks
    for the shape: the int32 sums are exact on both, the scaling rule is shared, so the two agree bit for bit (row counts
    that are no multiple of the tiles, 128 x 128 and 256 x 256 tiles, one K step that is not a multiple of 128: library
    only) and with the oracle within the bf16 tolerance."""
    from oracle.quant import w8a8_gemm_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear, w8a8_use_library

    assert not w8a8_use_library(128, K, N)
    g_ = gen(21)
    x = randn_bf16(g_, M, K)
    wq = torch.randint(-127, 128, (N, K), generator=g_, dtype=torch.int8)
    ws = torch.rand(N, generator=g_) * 0.02 + 0.001
    bias = randn_bf16(g_, N, scale=0.1)
    xd, wd, sd, bd = x.cuda(), wq.cuda(), ws.cuda(), bias.cuda()
    lin = HymetaLinear.from_w8a8(wd, sd, bd)
    out = lin(xd)
    xq, xs = hip_ops.w8a8_quantize(xd)
    library = hip_ops.w8a8_scale_out(torch._int_mm(xq, wd.t()), xs, sd, bias=bd)
    torch.cuda.synchronize()
    assert torch.equal(out, library)
    if K % 128 == 0 and N % 128 == 0:
        tiled = hip_ops.w8a8_gemm(xd, wd, sd, bias=bd)          # the entry point itself: the tiled kernel above 128 rows
        torch.cuda.synchronize()
        assert torch.equal(tiled, library)
    else:
        assert w8a8_use_library(M, K, N)
    if M <= 2048:
        assert_bf16_close(out.cpu(), w8a8_gemm_ref(x, wq, ws, bias), f"w8a8 linear M={M}", rel=6e-3)


@pytest.mark.parametrize("M,K,N", [(129, 512, 256), (300, 512, 37888), (1000, 1024, 1024), (2048, 3584, 9472), (257, 640, 384)])
def test_w8a8_gate_up_with_the_activation_in_the_tiled_epilogue(M, K, N):
    """Above the streaming kernel's rows the fused gate / up projection of a W8A8 MLP runs the tiled kernel with gate and
    up rows of the same output columns in one tile and silu(gate) * up in its store pass (128 x 128 and 256 x 256 tiles,
    ragged row counts, a 70B TP=4 shard's width): the bits of the linear followed by silu_and_mul."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(31)
    x = randn_bf16(g_, M, K).cuda()
    wq = torch.randint(-127, 128, (N, K), generator=g_, dtype=torch.int8).cuda()
    ws = (torch.rand(N, generator=g_) * 0.004 + 0.0005).cuda()
    lin = HymetaLinear.from_w8a8(wq, ws)
    assert hip_ops.gemm_silu_mul_supported(x, wq)
    fused = lin.forward_silu_mul(x)
    two = hip_ops.silu_and_mul(lin(x))
    torch.cuda.synchronize()
    assert fused.shape == (M, N // 2) and torch.equal(fused, two)
    assert float(fused.float().abs().max()) > 0.01


def test_w8a8_path_choice_above_the_streaming_rows():
    from vllm_hymeta.linear import w8a8_use_library

    assert not w8a8_use_library(129, 3584, 37888) and not w8a8_use_library(32768, 3584, 37888)      # gate_up: tiled from 129 rows
    assert w8a8_use_library(512, 3584, 4608) and not w8a8_use_library(1024, 3584, 4608)             # qkv / o: from 1024
    assert w8a8_use_library(2048, 18944, 3584) and not w8a8_use_library(4096, 18944, 3584)          # down: from 4096
    assert w8a8_use_library(4096, 3584 + 64, 4608) and w8a8_use_library(4096, 3584, 4608 + 64)      # shapes the tiles do not cut


@pytest.mark.parametrize("tp", [2, 4])
def test_tensor_parallel_gptq_mlp_shards_sum_to_the_unsharded_layer(tp):
    """gate_up (column parallel) and down (row parallel) cut for tp ranks by the checkpoint loader:
    the ranks' partial outputs, summed as the layer's all-reduce would, equal the unsharded MLP."""
    from oracle.moe import silu_and_mul_ref
    from oracle.quant import gptq_gemm_ref
    from vllm_hymeta import hip_ops, loader

    g_ = gen(31)
    hidden, inter, M, group = 512, 1024, 24, 128
    def mk(seed, K, N):
        qw, qz, sc = _gptq_problem(seed, K, N, 4, group, torch.float16)
        return {"qweight": qw, "qzeros": qz, "scales": sc}
    gate, up, down = mk(1, hidden, inter), mk(2, hidden, inter), mk(3, inter, hidden)
    x = randn_bf16(g_, M, hidden)
    h_ref = silu_and_mul_ref(torch.cat([gptq_gemm_ref(x, gate["qweight"], gate["qzeros"], gate["scales"], 4, group),
                                        gptq_gemm_ref(x, up["qweight"], up["qzeros"], up["scales"], 4, group)], dim=1).to(torch.bfloat16))
    ref = gptq_gemm_ref(h_ref, down["qweight"], down["qzeros"], down["scales"], 4, group)
    total = torch.zeros(M, hidden, dtype=torch.float32, device="cuda")
    for rank in range(tp):
        gu = loader.make_linear(loader.shard_gate_up(gate, up, tp, rank), 4, group, device="cuda")
        dn = loader.make_linear(loader.shard_row_parallel(down, tp, rank, 4, group), 4, group, device="cuda")
        total += dn(hip_ops.silu_and_mul(gu(x.cuda()))).float()
    torch.cuda.synchronize()
    assert_bf16_close(total.cpu(), ref, f"tp={tp} GPTQ MLP", rel=1e-2)


@pytest.mark.parametrize("partials", [False, True])
def test_norm_that_quantises_its_output_feeds_w8a8_linears(partials):
    """The block norm can write the int8 quantisation of its output in the same launch; the rows equal the stand-alone
    quantise kernel's bit for bit, and W8A8 linears (plain and gated) called on that tensor give the bits of the plain
    calls."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear

    g_ = gen(77)
    M, hidden, inter = 48, 3584, 1024
    w = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16).cuda()
    res = randn_bf16(g_, M, hidden).cuda()
    if partials:
        src = HymetaLinear.from_weight(randn_bf16(g_, hidden, 2048, scale=2048 ** -0.5).cuda())
        xin = randn_bf16(g_, M, 2048).cuda()
        make_x = lambda: src(xin, defer=True)
        assert isinstance(make_x(), hip_ops.PartialSums)
    else:
        x0 = randn_bf16(g_, M, hidden).cuda()
        make_x = lambda: x0
    r1, r2 = res.clone(), res.clone()
    plain = hip_ops.add_rmsnorm(make_x(), w, 1e-6, residual=r1)
    fused = hip_ops.add_rmsnorm(make_x(), w, 1e-6, residual=r2, int8=True)
    xq, xs = hip_ops.int8_rows_of(fused)
    xq_ref, xs_ref = hip_ops.w8a8_quantize(plain)
    torch.cuda.synchronize()
    assert torch.equal(plain, fused) and torch.equal(r1, r2) and hip_ops.int8_rows_of(plain) is None
    assert torch.equal(xq, xq_ref) and torch.equal(xs, xs_ref)
    qkv = HymetaLinear.from_w8a8(torch.randint(-127, 128, (4608, hidden), generator=g_, dtype=torch.int8).cuda(),
                                 (torch.rand(4608, generator=g_) * 0.02 + 0.001).cuda(), randn_bf16(g_, 4608).cuda())
    gate_up = HymetaLinear.from_w8a8(torch.randint(-127, 128, (2 * inter, hidden), generator=g_, dtype=torch.int8).cuda(),
                                     (torch.rand(2 * inter, generator=g_) * 0.02 + 0.001).cuda())
    a, b = qkv(fused), qkv(plain)
    c, d = gate_up.forward_silu_mul(fused), gate_up.forward_silu_mul(plain)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(c, d)
    # no residual (the first block's norm)
    first = hip_ops.add_rmsnorm(plain, w, 1e-6, int8=True)
    q2, s2 = hip_ops.w8a8_quantize(hip_ops.add_rmsnorm(plain, w, 1e-6))
    torch.cuda.synchronize()
    assert torch.equal(hip_ops.int8_rows_of(first)[0], q2) and torch.equal(hip_ops.int8_rows_of(first)[1], s2)
