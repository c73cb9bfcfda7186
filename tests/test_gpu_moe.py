"""Sparse MoE block (routing, counting sort, grouped expert GEMMs, activation, combine) and the
dense MLP vs the CPU restatement."""
import pytest
import torch

from _util import assert_bf16_close, assert_bf16_ulp, gen, randn_bf16

pytestmark = pytest.mark.gpu


def _problem(seed, T, H, inter, E):
    g_ = gen(seed)
    x = randn_bf16(g_, T, H)
    gate_w = randn_bf16(g_, E, H, scale=H ** -0.5)
    w13 = randn_bf16(g_, E, 2 * inter, H, scale=H ** -0.5)
    w2 = randn_bf16(g_, E, H, inter, scale=inter ** -0.5)
    logits = (x.float() @ gate_w.float().t()).to(torch.bfloat16)
    return x, gate_w, w13, w2, logits


def test_silu_and_mul_bit_exact():
    from oracle.moe import silu_and_mul_ref
    from vllm_hymeta import hip_ops

    x = randn_bf16(gen(0), 37, 2 * 320, scale=3.0)
    out = hip_ops.silu_and_mul(x.cuda())
    torch.cuda.synchronize()
    ref = silu_and_mul_ref(x)
    # exp / reciprocal differ from torch's by at most an fp32 ulp: at most one bf16 ulp, rarely
    diff = (out.cpu().view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.01


@pytest.mark.parametrize("T,E,top_k,renorm,with_shared", [
    (1, 16, 1, False, False),       # single decode token
    (64, 16, 1, False, True),       # decode batch, the 70B routing (top-1, softmax weight), shared expert
    (300, 16, 1, False, True),      # prefill chunk: 64-row tiles, ragged expert sizes
    (50, 8, 2, True, False),        # the eager module's top-2 with renormalised weights
    (40, 4, 2, False, True),
    (900, 4, 1, False, True),       # many rows per expert: the 128 x 128 tiled grouped GEMM, ragged last tiles
    (700, 4, 2, True, False),       # the same with two choices per token
])
def test_moe_forward(T, E, top_k, renorm, with_shared):
    from oracle.moe import moe_ref
    from vllm_hymeta import hip_ops

    H, inter = 256, 192
    x, gate_w, w13, w2, logits = _problem(1, T, H, inter, E)   # bf16 logits: ties between experts do occur
    shared = randn_bf16(gen(2), T, H) if with_shared else None
    ref = moe_ref(x, logits, w13, w2, top_k, renorm, shared)
    out = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda(), top_k=top_k, renormalize=renorm,
                              shared=None if shared is None else shared.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"moe T={T} E={E} k={top_k}", rel=8e-3)


@pytest.mark.parametrize("T,E,H,inter", [
    (64, 16, 1024, 320),     # whole-line weight loads: 16 and 5 pairs of k-steps over the 4 waves (uneven shares)
    (300, 16, 832, 448),     # the same with 64-row tiles: 13 and 7 pairs
    (64, 16, 256, 160),      # K = 160 is no multiple of 64: the down GEMM takes the fragment-shaped loads
    (3, 16, 64, 64),         # one pair: three of the four waves have no k-step at all
])
def test_moe_decode_gemm_k_shapes(T, E, H, inter):
    from oracle.moe import moe_ref
    from vllm_hymeta import hip_ops

    x, gate_w, w13, w2, logits = _problem(5, T, H, inter, E)
    ref = moe_ref(x, logits, w13, w2, 1, False, None)
    out = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda(), top_k=1)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"moe T={T} H={H} I={inter}", rel=8e-3)


@pytest.mark.parametrize("T,E,top_k,H", [(64, 16, 1, 3584), (1, 16, 1, 3584), (256, 16, 1, 512), (37, 8, 2, 320), (5, 64, 2, 256)])
def test_router_in_the_routing_launch(T, E, top_k, H):
    """Decode-sized batches: the router's linear layer runs inside the routing launch.  Its logits equal the bf16 linear
    layer's within one bf16 ulp (another order of the fp32 sums); routed with exactly those logits, the block equals
    the CPU restatement and, bit for bit, the block that is handed the same logits."""
    from oracle.moe import moe_ref
    from vllm_hymeta import hip_ops

    inter = 192
    x, gate_w, w13, w2, _ = _problem(6, T, H, inter, E)
    shared = randn_bf16(gen(7), T, H)
    xd, gd = x.cuda(), gate_w.cuda()
    assert hip_ops.moe_router_fused(xd, gd)
    logits = torch.empty(T, E, dtype=torch.bfloat16, device="cuda")
    for _ in range(2):       # twice: the launch leaves its ticket as it found it
        out = hip_ops.moe_forward(xd, None, w13.cuda(), w2.cuda(), top_k=top_k, shared=shared.cuda(), gate_weight=gd, logits_out=logits)
    torch.cuda.synchronize()
    lin = torch.nn.functional.linear(x.float(), gate_w.float())
    assert_bf16_ulp(logits.cpu(), lin, "router logits", 1)
    ref = moe_ref(x, logits.cpu(), w13, w2, top_k, False, shared)
    assert_bf16_close(out.cpu(), ref, f"fused router T={T} E={E} k={top_k}", rel=8e-3)
    same = hip_ops.moe_forward(xd, logits, w13.cuda(), w2.cuda(), top_k=top_k, shared=shared.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out, same)
    # a batch above the limit takes the linear layer and the separate routing launches
    xb = randn_bf16(gen(8), 300, H).cuda()
    assert not hip_ops.moe_router_fused(xb, gd)
    big = hip_ops.moe_forward(xb, None, w13.cuda(), w2.cuda(), top_k=top_k, gate_weight=gd)
    ref_big = hip_ops.moe_forward(xb, torch.nn.functional.linear(xb, gd), w13.cuda(), w2.cuda(), top_k=top_k)
    torch.cuda.synchronize()
    assert torch.equal(big, ref_big)


@pytest.mark.parametrize("T,E,top_k,with_shared,H", [
    (64, 16, 1, True, 256), (1, 16, 1, False, 256), (12, 4, 2, True, 256), (100, 16, 1, True, 256),
    (100, 4, 1, True, 192),       # 25 rows per expert with 192 columns: the 64-row tiles of the decode-sized kernel
])
def test_combine_inside_the_down_gemm_equals_the_separate_launch(T, E, top_k, with_shared, H):
    """Decode-sized batches with one choice per token fold the combine with the shared expert into the down GEMM's
    epilogue: the same bits as the separate launch (two choices per token keep the launch)."""
    from vllm_hymeta import hip_ops

    inter = 192
    x, gate_w, w13, w2, logits = _problem(9, T, H, inter, E)
    shared = randn_bf16(gen(10), T, H).cuda() if with_shared else None
    args = (x.cuda(), logits.cuda(), w13.cuda(), w2.cuda())
    assert hip_ops.MOE_FUSE_DOWN and _C_fuses(T * top_k, E, inter, H)
    fused = hip_ops.moe_forward(*args, top_k=top_k, shared=shared)
    hip_ops.MOE_FUSE_DOWN = False
    try:
        plain = hip_ops.moe_forward(*args, top_k=top_k, shared=shared)
    finally:
        hip_ops.MOE_FUSE_DOWN = True
    torch.cuda.synchronize()
    assert torch.equal(fused, plain)


def _C_fuses(rows, experts, K, N):
    from vllm_hymeta import _C
    return bool(_C.load().hymeta_moe_gemm_fuses(rows, experts, K, N))


def test_moe_expert_parallel_shares_sum_to_the_whole():
    """Two expert-parallel ranks (emulated in one process): each computes only its experts' pairs;
    the sum of the two outputs (the layer's all-reduce) equals the single-rank result, and each
    share equals the oracle's share."""
    from oracle.moe import moe_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.models.moe import expert_range

    T, H, inter, E = 90, 256, 128, 16
    x, gate_w, w13, w2, logits = _problem(3, T, H, inter, E)
    full = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda())
    parts = []
    for r in range(2):
        lo_hi = expert_range(E, 2, r)
        part = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda(), local_experts=lo_hi)
        torch.cuda.synchronize()
        assert_bf16_close(part.cpu(), moe_ref(x, logits, w13, w2, local_experts=lo_hi), f"EP share {r}", rel=8e-3)
        parts.append(part.float())
    # top-1: every token belongs to exactly one rank, so the sum is the whole -- up to the fp32
    # summation order of the GEMM variant that each call's rows-per-expert average selects
    assert_bf16_ulp((parts[0] + parts[1]).to(torch.bfloat16), full, "EP shares summed vs single rank", ulps=1)
    with pytest.raises(ValueError):
        expert_range(16, 3, 0)
    # a rank that holds ONLY its experts (what an expert-parallel rank of the model keeps): the same bits as its share
    # computed from the full tensors
    for r in range(2):
        lo, hi = expert_range(E, 2, r)
        whole = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda(), local_experts=(lo, hi))
        held = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13[lo:hi].contiguous().cuda(), w2[lo:hi].contiguous().cuda(),
                                   local_experts=(lo, hi))
        torch.cuda.synchronize()
        assert torch.equal(whole, held)
    with pytest.raises(ValueError):      # neither all experts nor exactly the local ones
        hip_ops.moe_forward(x.cuda(), logits.cuda(), w13[:5].contiguous().cuda(), w2[:5].contiguous().cuda(), local_experts=(0, 8))


@pytest.mark.parametrize("T", [8, 300])
def test_a_row_of_nan_logits_gives_a_nan_row_and_nothing_else(T):
    """Both routing kernels (router inside the launch up to 256 tokens; separate launches above): a token whose logits are
    NaN picks a valid expert with a NaN weight -- its output row is NaN, every other row is what it is without that
    token, and no counter is indexed out of range."""
    from vllm_hymeta import hip_ops

    H, inter, E = 256, 128, 16
    x, gate_w, w13, w2, _ = _problem(17, T, H, inter, E)
    bad = 3
    xn = x.clone()
    xn[bad] = float("nan")
    xd, gw, a, b = xn.cuda(), gate_w.cuda(), w13.cuda(), w2.cuda()
    shared = randn_bf16(gen(18), T, H).cuda()
    out = hip_ops.moe_forward(xd, None, a, b, gate_weight=gw, shared=shared)
    ref = hip_ops.moe_forward(x.cuda(), None, a, b, gate_weight=gw, shared=shared)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out[bad]).all())
    keep = [i for i in range(T) if i != bad]
    assert_bf16_ulp(out[keep], ref[keep], "rows beside the NaN row", ulps=1)


def test_moe_tiled_regime_expert_parallel_share():
    """Prefill-sized chunk on the tiled grouped GEMM, one expert-parallel rank's experts only."""
    from oracle.moe import moe_ref
    from vllm_hymeta import hip_ops

    T, H, inter, E = 1000, 256, 256, 4      # both GEMMs on the 256 x 256 tile
    x, gate_w, w13, w2, logits = _problem(5, T, H, inter, E)
    part = hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda(), local_experts=(2, 4))
    torch.cuda.synchronize()
    assert_bf16_close(part.cpu(), moe_ref(x, logits, w13, w2, local_experts=(2, 4)), "tiled EP share", rel=8e-3)


def test_hymeta_moe_module_and_dense_mlp():
    from oracle.moe import mlp_ref, moe_ref
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.models.moe import HymetaMLP, HymetaMoE

    T, H, inter, E = 33, 256, 192, 16
    x, gate_w, w13, w2, logits = _problem(4, T, H, inter, E)
    g_ = gen(5)
    s13 = randn_bf16(g_, 2 * inter, H, scale=H ** -0.5)
    s2 = randn_bf16(g_, H, inter, scale=inter ** -0.5)
    mlp = HymetaMLP(HymetaLinear.from_weight(s13.cuda()), HymetaLinear.from_weight(s2.cuda()))
    y = mlp(x.cuda())
    torch.cuda.synchronize()
    shared_ref = mlp_ref(x, s13, s2)
    assert_bf16_close(y.cpu(), shared_ref, "dense MLP", rel=8e-3)
    moe = HymetaMoE(gate_w.cuda(), w13.cuda(), w2.cuda(), shared_mlp=mlp)
    out = moe(x.cuda().view(3, 11, H))
    torch.cuda.synchronize()
    # route the oracle with the router output of the device's own bf16 linear (a library GEMM whose
    # rounding may differ from the host's in the last bit, which can flip a near-tie)
    logits = torch.nn.functional.linear(x.cuda(), gate_w.cuda()).cpu()
    ref = moe_ref(x, logits, w13, w2, 1, False, shared_ref)
    assert out.shape == (3, 11, H)
    assert_bf16_close(out.cpu().view(T, H), ref, "HymetaMoE + shared expert", rel=8e-3)


def _w8a8_experts(seed, E, H, inter):
    g_ = gen(seed)
    w13 = torch.randint(-127, 128, (E, 2 * inter, H), generator=g_, dtype=torch.int8)
    w2 = torch.randint(-127, 128, (E, H, inter), generator=g_, dtype=torch.int8)
    s13 = (torch.rand(E, 2 * inter, generator=g_) + 0.5) * (H ** -0.5 / 73.0)     # int8 weights of unit-variance rows
    s2 = (torch.rand(E, H, generator=g_) + 0.5) * (inter ** -0.5 / 73.0)
    return w13, s13, w2, s2


@pytest.mark.parametrize("inter", [320, 4736])
def test_silu_and_mul_int8_is_the_activation_then_the_row_quantisation(inter):
    from vllm_hymeta import hip_ops

    x = randn_bf16(gen(3), 41, 2 * inter, scale=2.0).cuda()
    q, sc = hip_ops.silu_and_mul_int8(x)
    q2, sc2 = hip_ops.w8a8_quantize(hip_ops.silu_and_mul(x))
    torch.cuda.synchronize()
    assert torch.equal(q, q2) and torch.equal(sc, sc2)


@pytest.mark.parametrize("T,E,top_k,renorm,with_shared,H,inter", [
    (1, 16, 1, False, False, 256, 192),      # one decode token; the down GEMM's K = 192 is no whole number of 128-byte lines
    (64, 16, 1, False, True, 1024, 384),     # decode batch: whole-line loads on both GEMMs, the combine inside the down GEMM
    (64, 16, 1, False, True, 512, 320),      # the down GEMM on fragment-shaped loads
    (100, 4, 2, True, False, 256, 128),      # two choices, renormalised, 64-row tiles
    (300, 16, 1, False, True, 512, 384),     # prefill chunk: 128 x 128 tiles, ragged expert sizes
    (900, 4, 1, False, True, 256, 256),      # many rows per expert: 256 x 256 tiles
])
def test_moe_forward_with_w8a8_experts(T, E, top_k, renorm, with_shared, H, inter):
    """Experts as a compressed-tensors W8A8 checkpoint holds them (int8 weights, per-output-channel scales): tokens and
    activated rows are quantised per row, the grouped GEMMs run on the int8 matrix cores with exact int32 sums and apply
    the scales once -- every kernel form (fragment loads, whole-line loads, both tiles) against the CPU restatement."""
    from oracle.moe import moe_w8a8_ref
    from vllm_hymeta import hip_ops

    x, gate_w, _, _, logits = _problem(5, T, H, 64, E)
    w13, s13, w2, s2 = _w8a8_experts(6, E, H, inter)
    shared = randn_bf16(gen(7), T, H) if with_shared else None
    ref = moe_w8a8_ref(x, logits, w13, s13, w2, s2, top_k, renorm, shared)
    args = (x.cuda(), logits.cuda(), w13.cuda(), w2.cuda())
    kw = dict(top_k=top_k, renormalize=renorm, shared=None if shared is None else shared.cuda(), w13_scale=s13.cuda(), w2_scale=s2.cuda())
    out, again = hip_ops.moe_forward(*args, **kw), hip_ops.moe_forward(*args, **kw)
    torch.cuda.synchronize()
    # Dynamic quantisation makes the comparison with an independent restatement loose by construction: when a last-bit
    # difference in the activated row (sigmoid, fp32 against fp64 scaling) moves the row's maximum by one bf16 ulp, the
    # row's scale moves with it and about a seventh of its int8 values change by one step -- as large as the
    # quantisation noise itself (about 1 % of the output).  The sharp check is the staged one below.
    assert_bf16_close(out.cpu(), ref, f"w8a8 moe T={T} E={E} k={top_k}", rel=2.5e-2, frac=0.08)
    assert torch.equal(out, again)
    if top_k == 1:
        # the same chain with the kernels' own quantisation launches and exact integer products on the CPU in between:
        # every rounding is the kernels' (float(acc) * s_x, * s_w, * routing weight, one bf16 rounding), so only the
        # routing weight's last fp32 bit (softmax on the GPU against torch's) can differ
        from oracle.moe import moe_route_ref
        ids, wts = moe_route_ref(logits, 1, renorm)
        xq, xs = hip_ops.w8a8_quantize(x.cuda())
        xq, xs = xq.cpu(), xs.cpu()
        staged = torch.zeros(T, H, dtype=torch.bfloat16)
        for e in range(E):
            rows = (ids[:, 0] == e).nonzero().flatten()
            if rows.numel() == 0:
                continue
            acc1 = (xq[rows].long() @ w13[e].long().t()).float()
            h = ((acc1 * xs[rows, None]) * s13[e][None, :]).to(torch.bfloat16)
            aq, a_s = hip_ops.w8a8_quantize(hip_ops.silu_and_mul(h.cuda()))
            acc2 = (aq.cpu().long() @ w2[e].long().t()).float()
            staged[rows] = (((acc2 * a_s.cpu()[:, None]) * s2[e][None, :]) * wts[rows, 0:1]).to(torch.bfloat16)
        if shared is not None:
            staged = (staged.float() + shared.float()).to(torch.bfloat16)
        assert_bf16_ulp(out.cpu(), staged, f"w8a8 moe staged T={T}", ulps=1)
        assert float((out.cpu() != staged).float().mean()) < 0.01
    if top_k == 1 and T <= 64:
        saved = hip_ops.MOE_FUSE_DOWN
        hip_ops.MOE_FUSE_DOWN = False
        try:
            plain = hip_ops.moe_forward(*args, **kw)
        finally:
            hip_ops.MOE_FUSE_DOWN = saved
        torch.cuda.synchronize()
        assert torch.equal(out, plain)          # the combine inside the down GEMM: the same bits


@pytest.mark.parametrize("T,E,top_k", [(64, 16, 1), (5, 8, 2), (256, 16, 1)])
def test_int8_router_in_the_routing_launch(T, E, top_k):
    """The router of a W8A8 checkpoint is int8 too: decode-sized batches route on the tokens' int8 rows inside the routing
    launch (integer dot products), with the logits of the W8A8 linear bit for bit; the module sends larger batches
    through that linear."""
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.models.moe import HymetaMoE

    H, inter = 512, 256
    x, gate_w, _, _, _ = _problem(11, T, H, 64, E)
    w13, s13, w2, s2 = _w8a8_experts(12, E, H, inter)
    gs = gate_w.float().abs().amax(dim=1) / 127.0
    gq = torch.round(gate_w.float() / gs[:, None]).to(torch.int8)
    pad = torch.zeros(64, H, dtype=torch.int8)
    pad[:E] = gq
    sc = torch.ones(64)
    sc[:E] = gs
    xd = x.cuda()
    ref_logits = HymetaLinear.from_w8a8(pad.cuda(), sc.cuda())(xd)[:, :E].contiguous()
    logits = torch.empty(T, E, dtype=torch.bfloat16, device="cuda")
    kw = dict(top_k=top_k, w13_scale=s13.cuda(), w2_scale=s2.cuda())
    out = hip_ops.moe_forward(xd, None, w13.cuda(), w2.cuda(), gate_weight=gq.cuda(), gate_scale=gs.cuda(), logits_out=logits, **kw)
    same = hip_ops.moe_forward(x.cuda(), ref_logits, w13.cuda(), w2.cuda(), **kw)
    torch.cuda.synchronize()
    assert torch.equal(logits, ref_logits) and torch.equal(out, same)
    moe = HymetaMoE(gq.cuda(), w13.cuda(), w2.cuda(), top_k=top_k, w13_scale=s13.cuda(), w2_scale=s2.cuda(), gate_scale=gs.cuda())
    assert moe.router_logits(xd) is None and torch.equal(moe(x.cuda()), out)
    xb = randn_bf16(gen(13), 300, H).cuda()
    big_logits = moe.router_logits(xb)
    assert big_logits is not None and big_logits.shape == (300, E)
    big = moe(xb)
    ref_big = hip_ops.moe_forward(xb, big_logits, w13.cuda(), w2.cuda(), **kw)
    torch.cuda.synchronize()
    assert torch.equal(big, ref_big)


def test_w8a8_experts_expert_parallel_shares():
    """A rank that holds experts [lo, hi) only (compact int8 tensors and scales) gives the bits of the full tensors with
    ``local_experts``; the shares add up to the whole block."""
    from oracle.moe import moe_w8a8_ref
    from vllm_hymeta import hip_ops

    T, E, H, inter = 64, 8, 512, 256
    x, gate_w, _, _, logits = _problem(8, T, H, 64, E)
    w13, s13, w2, s2 = _w8a8_experts(9, E, H, inter)
    xd, ld = x.cuda(), logits.cuda()
    total = torch.zeros(T, H)
    for lo, hi in ((0, 4), (4, 8)):
        whole = hip_ops.moe_forward(xd, ld, w13.cuda(), w2.cuda(), local_experts=(lo, hi), w13_scale=s13.cuda(), w2_scale=s2.cuda())
        held = hip_ops.moe_forward(xd, ld, w13[lo:hi].contiguous().cuda(), w2[lo:hi].contiguous().cuda(), local_experts=(lo, hi),
                                   w13_scale=s13[lo:hi].contiguous().cuda(), w2_scale=s2[lo:hi].contiguous().cuda())
        torch.cuda.synchronize()
        assert torch.equal(whole, held)
        assert_bf16_close(whole.cpu(), moe_w8a8_ref(x, logits, w13, s13, w2, s2, local_experts=(lo, hi)), f"EP share {lo}", rel=2.5e-2,
                          frac=0.08)
        total += whole.float().cpu()
    full = hip_ops.moe_forward(xd, ld, w13.cuda(), w2.cuda(), w13_scale=s13.cuda(), w2_scale=s2.cuda())
    torch.cuda.synchronize()
    assert torch.equal(total.to(torch.bfloat16), full.cpu())      # top-1: every token's row comes from exactly one share


def _gptq_experts(seed, E, K, N, bits, group, scales_dtype=torch.float16):
    """Per-expert GPTQ tensors of [N, K] weights of about unit-variance rows: (list for the oracle, stacked for the kernels)."""
    from oracle.quant import gptq_pack_ref
    from vllm_hymeta import hip_ops

    g_ = gen(seed)
    per = []
    for _ in range(E):
        wi = torch.randint(0, 2 ** bits, (K, N), generator=g_)
        z = torch.randint(1, 2 ** bits, (K // group, N), generator=g_)
        sc = ((torch.rand(K // group, N, generator=g_) * 0.5 + 0.75) * (3.4 / 2 ** bits) * K ** -0.5).to(scales_dtype)
        qw, qz = gptq_pack_ref(wi, z, bits)
        per.append((qw, qz, sc))
    stack = hip_ops.GptqExperts(torch.stack([p[0] for p in per]).cuda(), torch.stack([p[1] for p in per]).cuda(),
                                torch.stack([p[2] for p in per]).cuda(), bits, group)
    return per, stack


@pytest.mark.parametrize("T,E,top_k,with_shared,H,inter,bits,group", [
    (1, 16, 1, False, 256, 128, 4, 128),       # one decode token
    (64, 16, 1, True, 512, 384, 4, 128),       # decode batch, 16-row tiles, waves with uneven group counts (3 groups on 4 waves)
    (64, 4, 2, False, 256, 256, 8, 64),        # 8-bit, group 64, two choices: 32-row tiles walked inside the workgroup
    (400, 4, 1, True, 256, 128, 4, 32),        # group 32; 100 rows per expert on average: still the fused kernel
    (800, 4, 1, True, 512, 256, 4, 128),       # prefill-sized: experts expanded to bf16, tiled grouped GEMM
    (800, 4, 1, False, 512, 384, 4, 128),      # the same with a down projection whose K is three groups (no multiple of 256)
])
def test_moe_forward_with_gptq_experts(T, E, top_k, with_shared, H, inter, bits, group):
    """Experts as a GPTQ checkpoint holds them: decode-sized batches through the fused de-quantising grouped GEMM,
    prefill-sized ones through the one-launch expansion and the bf16 tiled kernel, against the CPU restatement."""
    from oracle.moe import moe_gptq_ref
    from vllm_hymeta import hip_ops

    x, gate_w, _, _, logits = _problem(15, T, H, 64, E)
    p13, s13 = _gptq_experts(16, E, H, 2 * inter, bits, group)
    p2, s2 = _gptq_experts(17, E, inter, H, bits, group, torch.bfloat16 if bits == 8 else torch.float16)
    shared = randn_bf16(gen(18), T, H) if with_shared else None
    ref = moe_gptq_ref(x, logits, p13, p2, bits, group, top_k, False, shared)
    out = hip_ops.moe_forward(x.cuda(), logits.cuda(), s13, s2, top_k=top_k, shared=None if shared is None else shared.cuda())
    again = hip_ops.moe_forward(x.cuda(), logits.cuda(), s13, s2, top_k=top_k, shared=None if shared is None else shared.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, f"gptq moe T={T} E={E} k={top_k} {bits}-bit", rel=8e-3)
    assert torch.equal(out, again)


def test_gptq_experts_expert_parallel_and_the_expansion():
    """Compact stacks of a rank's experts give the bits of the full stacks with ``local_experts`` (both regimes); the
    one-launch expansion is the dense expansion of every expert."""
    from oracle.moe import moe_gptq_ref
    from vllm_hymeta import hip_ops

    E, H, inter = 8, 512, 256
    p13, s13 = _gptq_experts(19, E, H, 2 * inter, 4, 128)
    p2, s2 = _gptq_experts(20, E, inter, H, 4, 128)
    dense = torch.stack([hip_ops.gptq_dequant(s13.qweight[e], s13.qzeros[e], s13.scales[e], 4, 128, transposed=True) for e in range(E)])
    assert torch.equal(hip_ops.moe_gptq_expand(s13), dense)
    cut = lambda w, lo, hi: w._replace(qweight=w.qweight[lo:hi].contiguous(), qzeros=w.qzeros[lo:hi].contiguous(),
                                       scales=w.scales[lo:hi].contiguous())
    for T in (64, 1400):      # both regimes: the fused kernel, and the expansion for the shares and for the whole
        x, gate_w, _, _, logits = _problem(21 + T, T, H, 64, E)
        xd, ld = x.cuda(), logits.cuda()
        total = torch.zeros(T, H)
        for lo, hi in ((0, 4), (4, 8)):
            whole = hip_ops.moe_forward(xd, ld, s13, s2, local_experts=(lo, hi))
            held = hip_ops.moe_forward(xd, ld, cut(s13, lo, hi), cut(s2, lo, hi), local_experts=(lo, hi))
            torch.cuda.synchronize()
            assert torch.equal(whole, held)
            assert_bf16_close(whole.cpu(), moe_gptq_ref(x, logits, p13, p2, local_experts=(lo, hi)), f"gptq EP share {lo} T={T}", rel=8e-3)
            total += whole.float().cpu()
        full = hip_ops.moe_forward(xd, ld, s13, s2)
        torch.cuda.synchronize()
        assert torch.equal(total.to(torch.bfloat16), full.cpu())


def test_moe_rejects_unsupported():
    from vllm_hymeta import hip_ops
    from vllm_hymeta._C import HymetaError

    x, gate_w, w13, w2, logits = _problem(6, 4, 256, 128, 16)
    with pytest.raises(HymetaError):
        hip_ops.moe_forward(x.cuda(), logits.cuda(), w13.cuda(), w2.cuda(), top_k=3)
