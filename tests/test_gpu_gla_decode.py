"""GLA decode kernel vs the CPU restatement, through the C ABI (ctypes)."""
import pytest
import torch

from _util import assert_bf16_close, assert_bf16_ulp, gen, randn_bf16

pytestmark = pytest.mark.gpu

D = 128


def _inputs(seed, B, H, Hkv, g_dtype, slots):
    g_ = gen(seed)
    q = randn_bf16(g_, B, H, D)
    k = randn_bf16(g_, B, Hkv, D)
    v = randn_bf16(g_, B, Hkv, D)
    # log-decay in (-0.2, 0): logsigmoid(N(0,1)) / 16, as the 7B gate produces
    g = (torch.nn.functional.logsigmoid(torch.randn(B, Hkv, D, generator=g_)) / 16).to(g_dtype)
    state = randn_bf16(g_, slots, H, D, D, scale=0.1)
    return q, k, v, g, state


@pytest.mark.parametrize("B,H,Hkv,g_dtype", [
    (3, 4, 4, torch.float32),     # 70B-style: k/v/g already per q-head, fp32 gate
    (3, 28, 4, torch.bfloat16),   # 7B: GQA 7:1, bf16 gate
    (5, 7, 1, torch.float32),     # 70B TP=4 shard
    (40, 28, 4, torch.bfloat16),  # >= 1024 (seq, head) units: unsplit kernel
])
def test_gla_decode_steps(B, H, Hkv, g_dtype):
    from oracle import gla_decode_step_ref
    from vllm_hymeta import hip_ops

    slots = B + 2
    q, k, v, g, state = _inputs(0, B, H, Hkv, g_dtype, slots)
    perm = torch.randperm(slots, generator=gen(1))[:B].to(torch.int32)
    perm[1] = -1  # one padded sequence
    state_free = state.clone()  # evolves on the oracle side only: bounds the 4-step drift
    state_dev = state.cuda()
    slot_dev = perm.cuda()
    for step in range(4):
        # each step is checked from identical inputs: the oracle starts from the device state
        state_ref = state_dev.cpu().clone()
        g_ = gen(100 + step)
        q = randn_bf16(g_, B, H, D)
        k = randn_bf16(g_, B, Hkv, D)
        v = randn_bf16(g_, B, Hkv, D)
        o_ref = gla_decode_step_ref(q, k, v, g, state_ref, perm)
        gla_decode_step_ref(q, k, v, g, state_free, perm)
        out = torch.full((B, H, D), 7.0, dtype=torch.bfloat16, device="cuda")
        hip_ops.gla_decode(q.cuda(), k.cuda(), v.cuda(), g.cuda(), state_dev, slot_dev, out=out)
        torch.cuda.synchronize()
        live = perm >= 0
        assert_bf16_close(out.cpu()[live], o_ref[live], f"gla_decode step {step}")
        # skipped sequence: output row untouched
        assert (out.cpu()[~live] == 7.0).all()
        assert_bf16_ulp(state_dev.cpu(), state_ref, f"gla state after step {step}")
    # independent 4-step evolution: bf16 re-rounding each step lets the two drift apart a little
    assert_bf16_close(state_dev.cpu(), state_free, "gla state drift over 4 steps", rel=2e-3, frac=2.0 ** -7)
    # untouched slots are bit-identical to the initial state
    used = set(int(s) for s in perm if s >= 0)
    for s in range(slots):
        if s not in used:
            assert torch.equal(state_dev.cpu()[s], state[s])


def test_gla_decode_strided_views_and_fp32_state():
    """q/k/v as views into one fused projection buffer; fp32 state option."""
    from oracle import gla_decode_step_ref
    from vllm_hymeta import hip_ops

    B, H, Hkv = 4, 28, 4
    g_ = gen(3)
    qkv = randn_bf16(g_, B, (H + 2 * Hkv) * D)
    q = qkv[:, : H * D].view(B, H, D)
    k = qkv[:, H * D:(H + Hkv) * D].view(B, Hkv, D)
    v = qkv[:, (H + Hkv) * D:].view(B, Hkv, D)
    g = (torch.nn.functional.logsigmoid(torch.randn(B, Hkv, D, generator=g_)) / 16).to(torch.bfloat16)
    state = (torch.randn(B, H, D, D, generator=g_) * 0.1)
    slot = torch.arange(B, dtype=torch.int32)
    ref_state = state.clone()
    o_ref = gla_decode_step_ref(q, k, v, g, ref_state, slot)
    qkv_d = qkv.cuda()
    qd = qkv_d[:, : H * D].view(B, H, D)
    kd = qkv_d[:, H * D:(H + Hkv) * D].view(B, Hkv, D)
    vd = qkv_d[:, (H + Hkv) * D:].view(B, Hkv, D)
    sd = state.cuda()
    out = hip_ops.gla_decode(qd, kd, vd, g.cuda(), sd, slot.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), o_ref, "gla_decode strided")
    assert torch.allclose(sd.cpu(), ref_state, rtol=1e-5, atol=1e-6)


def test_gla_decode_fused_7b_gate_and_norm():
    """Fused prologue (relu, logsigmoid/16) and epilogue (per-head RMS norm)."""
    from oracle import gla_decode_step_ref
    from oracle.layers import gate_7b_ref, rms_norm_ref
    from vllm_hymeta import hip_ops, _C

    B, H, Hkv = 6, 28, 4
    g_ = gen(5)
    q, k, v = randn_bf16(g_, B, H, D), randn_bf16(g_, B, Hkv, D), randn_bf16(g_, B, Hkv, D)
    gk = randn_bf16(g_, B, Hkv, D, scale=2.0)
    w = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16)
    state = randn_bf16(g_, B, H, D, D, scale=0.1)
    slot = torch.arange(B, dtype=torch.int32)
    qg, kg, gg = gate_7b_ref(q, k, gk)
    ref_state = state.clone()
    o_ref = rms_norm_ref(gla_decode_step_ref(qg, kg, v, gg, ref_state, slot), w, 1e-6)
    sd = state.cuda()
    out = hip_ops.gla_decode(q.cuda(), k.cuda(), v.cuda(), gk.cuda(), sd, slot.cuda(),
                             gate_mode=_C.GATE_7B, out_norm_weight=w.cuda(), out_norm_eps=1e-6)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), o_ref, "gla_decode fused 7B gate + g_norm", rel=8e-3)
    assert_bf16_ulp(sd.cpu(), ref_state, "state (fused 7B gate)", ulps=2)


@pytest.mark.parametrize("state_heads", ["kv", "query"])
def test_gla_decode_low_rank_gate_applied_in_the_kernel(state_heads):
    """The 7B gate given as its rank-16 input (rows of a wider, strided tensor, as they sit behind the qkv columns): the
    kernel's second linear reproduces hymeta_lowrank_gate's second stage, so outputs and states equal the run on the
    materialised gate bit for bit."""
    from vllm_hymeta import hip_ops, _C

    B, H, Hkv, hidden = 9, 28, 4, 512
    g_ = gen(6)
    q, k, v = randn_bf16(g_, B, H, D).cuda(), randn_bf16(g_, B, Hkv, D).cuda(), randn_bf16(g_, B, Hkv, D).cuda()
    x = randn_bf16(g_, B, hidden).cuda()
    w0 = randn_bf16(g_, 16, hidden, scale=hidden ** -0.5).cuda()
    w1, b1 = randn_bf16(g_, Hkv * D, 16, scale=0.5).cuda(), randn_bf16(g_, Hkv * D, scale=0.1).cuda()
    w = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16).cuda()
    heads = Hkv if state_heads == "kv" else H
    state = randn_bf16(g_, B, heads, D, D, scale=0.1).cuda()
    slot = torch.arange(B, dtype=torch.int32).cuda()
    kw = dict(gate_mode=_C.GATE_7B, out_norm_weight=w, out_norm_eps=1e-6)
    gk = hip_ops.lowrank_gate(x, w0, w1, b1).view(B, Hkv, D)
    s_a = state.clone()
    out_a = hip_ops.gla_decode(q, k, v, gk, s_a, slot, **kw)
    # the same rank-16 rows the two-stage kernel forms (an identity second stage exposes them), inside a wider tensor
    eye = torch.eye(16, dtype=torch.bfloat16, device="cuda")
    wide = torch.zeros(B, 64, dtype=torch.bfloat16, device="cuda")
    wide[:, 40:56] = hip_ops.lowrank_gate(x, w0, eye, None)
    s_b = state.clone()
    out_b = hip_ops.gla_decode(q, k, v, wide[:, 40:56], s_b, slot, g_w1=w1, g_bias=b1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(s_a, s_b)
    if state_heads == "kv":
        # both linears in the kernel, from the layer input (rows of a wider tensor): the same bits again
        xw = torch.zeros(B, hidden + 64, dtype=torch.bfloat16, device="cuda")
        xw[:, 8:8 + hidden] = x
        s_c = state.clone()
        out_c = hip_ops.gla_decode(q, k, v, xw[:, 8:8 + hidden], s_c, slot, g_w0=w0, g_w1=w1, g_bias=b1, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_c) and torch.equal(s_a, s_c)
    else:
        with pytest.raises(Exception):
            hip_ops.gla_decode(q, k, v, x, state.clone(), slot, g_w0=w0, g_w1=w1, g_bias=b1, **kw)


@pytest.mark.parametrize("first_layer", [True, False])
def test_gla_decode_fused_70b_gate(first_layer):
    from oracle import gla_decode_step_ref
    from oracle.layers import gate_70b_ref
    from vllm_hymeta import hip_ops, _C

    B, H, Hkv = 5, 7, 1
    g_ = gen(9)
    q, k, v = randn_bf16(g_, B, H, D), randn_bf16(g_, B, Hkv, D), randn_bf16(g_, B, Hkv, D)
    lb = torch.rand(H, D, generator=g_) * 0.5
    state = randn_bf16(g_, B, H, D, D, scale=0.1)
    slot = torch.arange(B, dtype=torch.int32)
    k_rep = k.repeat_interleave(H // Hkv, dim=1)
    v_rep = v.repeat_interleave(H // Hkv, dim=1)
    qg, kg, gg = gate_70b_ref(q, k_rep, lb, first_layer)
    ref_state = state.clone()
    o_ref = gla_decode_step_ref(qg, kg, v_rep, gg, ref_state, slot)
    sd = state.cuda()
    mode = _C.GATE_70B_CLAMP if first_layer else _C.GATE_70B_LOWER_BOUND
    out = hip_ops.gla_decode(q.cuda(), k.cuda(), v.cuda(), None, sd, slot.cuda(), gate_mode=mode,
                             lower_bound=lb.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), o_ref, "gla_decode fused 70B gate")
    assert_bf16_ulp(sd.cpu(), ref_state, "state (fused 70B gate)", ulps=2)


def test_gla_decode_rejects_bad_arguments():
    from vllm_hymeta import hip_ops

    q = torch.zeros(2, 4, 64, dtype=torch.bfloat16, device="cuda")  # head_dim 64 unsupported
    st = torch.zeros(2, 4, 64, 64, dtype=torch.bfloat16, device="cuda")
    slot = torch.zeros(2, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        hip_ops.gla_decode(q, q, q, q, st, slot)
    with pytest.raises(ValueError):
        hip_ops.gla_decode(q.float(), q, q, q, st, slot)


def test_gla_decode_rejects_inconsistent_partial_sum_arguments():
    """Through the C ABI: W8A8 scales of the partial sums come in pairs; int32 slabs need the state per key/value head."""
    import ctypes as C
    from vllm_hymeta import _C

    B, H, Hkv = 2, 28, 4
    dev = "cuda"
    out = torch.zeros(B, H, D, dtype=torch.bfloat16, device=dev)
    slabs = torch.zeros(2, B, (H + 2 * Hkv) * D, dtype=torch.int32, device=dev)
    scale = torch.ones((H + 2 * Hkv) * D, dtype=torch.float32, device=dev)
    w1 = torch.zeros(Hkv * D, 16, dtype=torch.bfloat16, device=dev)
    w0 = torch.zeros(16, 256, dtype=torch.bfloat16, device=dev)
    x = torch.zeros(B, 256, dtype=torch.bfloat16, device=dev)
    slot = torch.arange(B, dtype=torch.int32, device=dev)

    def call(state, **fields):
        a = _C.GlaDecodeArgs()
        a.out, a.state, a.slot_idx = out.data_ptr(), state.data_ptr(), slot.data_ptr()
        a.batch, a.num_heads, a.num_kv_heads, a.head_dim = B, H, Hkv, D
        a.out_stride_b, a.out_stride_h = out.stride(0), out.stride(1)
        a.state_stride_slot, a.state_stride_h, a.state_heads = state.stride(0), state.stride(1), state.shape[1]
        a.gate_mode, a.g_w1, a.g_w0, a.g_hidden, a.g, a.g_stride_b = _C.GATE_7B, w1.data_ptr(), w0.data_ptr(), 256, x.data_ptr(), 256
        a.qkv_slabs, a.qkv_splits, a.qkv_n = slabs.data_ptr(), 2, slabs.shape[2]
        a.q_col, a.k_col, a.v_col = 0, H * D, (H + Hkv) * D
        for k, v in fields.items():
            setattr(a, k, v)
        return _C.load().hymeta_gla_decode(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream))

    grouped = torch.zeros(B, Hkv, D, D, dtype=torch.bfloat16, device=dev)
    per_head = torch.zeros(B, H, D, D, dtype=torch.bfloat16, device=dev)
    assert call(grouped, qkv_x_scale=scale.data_ptr(), qkv_w_scale=scale.data_ptr()) == 0
    assert call(grouped, qkv_x_scale=scale.data_ptr()) == -1                  # one scale without the other
    assert call(per_head, qkv_x_scale=scale.data_ptr(), qkv_w_scale=scale.data_ptr()) == -2      # state per query head
    torch.cuda.synchronize()


def test_gla_decode_reads_the_projection_from_its_partial_sums():
    """7B GLA layer, pure decode: q, k, v and the gate's rank-16 input taken from the K-slice sums of the fused qkv (+ gate)
    projection give the outputs and states of the run on the finished projection, bit for bit."""
    from vllm_hymeta import hip_ops, _C
    from vllm_hymeta.linear import HymetaLinear

    B, H, Hkv, hidden = 23, 28, 4, 3584
    QKV = (H + 2 * Hkv) * D
    g_ = gen(8)
    x = randn_bf16(g_, B, hidden).cuda()
    lin = HymetaLinear.from_weight(randn_bf16(g_, QKV + 64, hidden, scale=hidden ** -0.5).cuda(), randn_bf16(g_, QKV + 64, scale=0.1).cuda())
    w1, b1 = randn_bf16(g_, Hkv * D, 16, scale=0.5).cuda(), randn_bf16(g_, Hkv * D, scale=0.1).cuda()
    w = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16).cuda()
    state = randn_bf16(g_, B + 2, Hkv, D, D, scale=0.1).cuda()
    slot = torch.randperm(B + 2, generator=g_)[:B].to(torch.int32).cuda()
    kw = dict(gate_mode=_C.GATE_7B, out_norm_weight=w, out_norm_eps=1e-6, g_w1=w1, g_bias=b1)
    qkv = lin(x)
    q, k, v = qkv[:, : H * D].view(B, H, D), qkv[:, H * D:(H + Hkv) * D].view(B, Hkv, D), qkv[:, (H + Hkv) * D:QKV].view(B, Hkv, D)
    s_a = state.clone()
    out_a = hip_ops.gla_decode(q, k, v, qkv[:, QKV:QKV + 16], s_a, slot, **kw)
    part = lin(x, defer=True)
    assert isinstance(part, hip_ops.PartialSums)
    s_b = state.clone()
    out_b = hip_ops.gla_decode((B, H, D), (B, Hkv, D), None, None, s_b, slot, qkv_partials=part,
                               qkv_cols=(0, H * D, (H + Hkv) * D, QKV), **kw)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(s_a, s_b)


@pytest.mark.parametrize("kind", ["w8a8", "gptq4", "gptq8"])
def test_gla_decode_reads_a_quantised_projection_from_its_partial_sums(kind):
    """7B GLA layer of a quantised checkpoint, pure decode: q, k, v from the K-slice sums of the quantised qkv projection
    (int32 slabs with the W8A8 scales, fp32 slabs for GPTQ) and both linears of the gate applied to the layer input inside
    the kernel give the outputs and states of the run on the finished projection, bit for bit."""
    from test_gpu_quant import _gptq_problem
    from vllm_hymeta import hip_ops, _C
    from vllm_hymeta.linear import HymetaLinear

    B, H, Hkv, hidden = 23, 28, 4, 3584
    QKV = (H + 2 * Hkv) * D
    g_ = gen(9)
    x = randn_bf16(g_, B, hidden).cuda()
    bias = randn_bf16(g_, QKV, scale=0.1).cuda()
    if kind == "w8a8":
        qw = torch.randint(-127, 128, (QKV, hidden), generator=g_, dtype=torch.int8).cuda()
        ws = (torch.rand(QKV, generator=g_) * (2.0 / 127) * hidden ** -0.5).float().cuda()
        lin = HymetaLinear.from_w8a8(qw, ws, bias=bias)
    else:
        bits, group = (4, 128) if kind == "gptq4" else (8, 64)
        qw, qz, scales = _gptq_problem(5, hidden, QKV, bits, group, torch.float16)
        lin = HymetaLinear.from_gptq(qw.cuda(), qz.cuda(), scales.cuda(), bits, group, bias=bias)
    w0 = randn_bf16(g_, 16, hidden, scale=hidden ** -0.5).cuda()
    w1, b1 = randn_bf16(g_, Hkv * D, 16, scale=0.5).cuda(), randn_bf16(g_, Hkv * D, scale=0.1).cuda()
    w = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16).cuda()
    state = randn_bf16(g_, B + 2, Hkv, D, D, scale=0.1).cuda()
    slot = torch.randperm(B + 2, generator=g_)[:B].to(torch.int32).cuda()
    kw = dict(gate_mode=_C.GATE_7B, out_norm_weight=w, out_norm_eps=1e-6, g_w0=w0, g_w1=w1, g_bias=b1)
    qkv = lin(x)
    q, k, v = qkv[:, : H * D].view(B, H, D), qkv[:, H * D:(H + Hkv) * D].view(B, Hkv, D), qkv[:, (H + Hkv) * D:QKV].view(B, Hkv, D)
    s_a = state.clone()
    out_a = hip_ops.gla_decode(q, k, v, x, s_a, slot, **kw)
    part = lin(x, defer=True)
    assert isinstance(part, hip_ops.PartialSums) and bool(part.desc.is_int32) == (kind == "w8a8")
    s_b = state.clone()
    out_b = hip_ops.gla_decode((B, H, D), (B, Hkv, D), None, x, s_b, slot, qkv_partials=part,
                               qkv_cols=(0, H * D, (H + Hkv) * D, 0), **kw)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b) and torch.equal(s_a, s_b)


def test_gla_decode_follows_the_reference_kernels_rounding_points():
    """The kernel's bf16 outputs against the restatement that the reference's own Triton kernel pins bit for bit
    (tests/test_ref_fixtures.py): q * scale rounded to bf16 with the scale itself rounded to bf16
    (my_fused_recurrent.py:59), k v^T rounded to bf16 (:65), fp32 update and contraction (:72-76).  What is left
    between the two is the order of the fp32 sum and the last bit of exp: a few outputs one bf16 ulp apart.  A kernel
    that multiplied by the unrounded scale in fp32 (the arithmetic before the fixture existed) differs on 40-60 % of
    the outputs; the bound below is far from that."""
    from oracle import gla_decode_step_ref
    from vllm_hymeta import hip_ops

    B, H, Hkv = 16, 8, 8
    q, k, v, g, state = _inputs(11, B, H, Hkv, torch.float32, B)
    slot = torch.arange(B, dtype=torch.int32)
    ref_state = state.clone()
    o_ref = gla_decode_step_ref(q, k, v, g, ref_state, slot)
    sd = state.cuda()
    out = hip_ops.gla_decode(q.cuda(), k.cuda(), v.cuda(), g.cuda(), sd, slot.cuda()).cpu()
    torch.cuda.synchronize()
    differ = (out.view(torch.int16) != o_ref.view(torch.int16)).float().mean().item()
    assert differ < 0.08, f"{differ:.3f} of the outputs differ from the pinned restatement"
    assert_bf16_ulp(out, o_ref, "gla_decode output vs pinned restatement")
    state_differ = (sd.cpu().view(torch.int16) != ref_state.view(torch.int16)).float().mean().item()
    assert state_differ < 0.01, f"{state_differ:.4f} of the state elements differ"
    # the unrounded-scale arithmetic, for the record of what the bound separates
    qs_old = (q.float() * (D ** -0.5))
    s_new = (k[:, :, :, None] * v[:, :, None, :]).float() + torch.exp(g)[:, :, :, None] * state[slot.long()].float()
    o_old = (qs_old[:, :, :, None] * s_new).sum(dim=2).to(torch.bfloat16)
    old_differ = (o_old.view(torch.int16) != o_ref.view(torch.int16)).float().mean().item()
    assert old_differ > 3 * max(differ, 0.02), (old_differ, differ)


def test_stale_struct_is_refused_before_any_field_is_read():
    """A binding compiled against an older header passes a shorter struct: every entry point compares the leading
    struct_size with its own sizeof and returns HYMETA_ERR_INVALID_ARG (-1) without touching the rest."""
    import ctypes as C

    from vllm_hymeta import _C

    lib = _C.load()
    stream = torch.cuda.current_stream().cuda_stream

    class OldGlaDecodeArgs(C.Structure):      # the struct as a first-cut binding declared it: no size, eleven fields short
        _fields_ = [f for f in _C.GlaDecodeArgs._fields_[1:] if f[0] not in (
            "state_heads", "g_w1", "g_bias", "qkv_slabs", "qkv_bias", "qkv_splits", "qkv_n", "q_col", "k_col", "v_col", "g_col",
            "g_w0", "g_hidden")]

    q = torch.zeros(1, 1, D, dtype=torch.bfloat16, device="cuda")
    old = OldGlaDecodeArgs(q=q.data_ptr(), k=q.data_ptr(), v=q.data_ptr(), g=q.data_ptr(), out=q.data_ptr(), batch=1, num_heads=1,
                           num_kv_heads=1, head_dim=D)
    fn = lib.hymeta_gla_decode
    saved = fn.argtypes
    fn.argtypes = [C.c_void_p, C.c_void_p]
    try:
        assert fn(C.addressof(old), stream) == -1
    finally:
        fn.argtypes = saved
    torch.cuda.synchronize()
