"""GLA state kept once per key/value head (state [slots, Hkv, D, D]) against the per-query-head
layout of the reference ([slots, H, D, D]): where the gate does not depend on the query head, the
query heads of a group receive identical updates, so outputs and states must agree bit for bit
with the per-head kernels (which the other GLA tests hold against the CPU restatement)."""
import numpy as np
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128


def _expand(state_kv, rep):
    return state_kv.repeat_interleave(rep, dim=1).contiguous()


@pytest.mark.parametrize("B,H,Hkv,state_dtype,mode", [
    (3, 28, 4, torch.bfloat16, "plain"),        # few units: two column halves per group
    (6, 28, 4, torch.bfloat16, "7b_norm"),      # fused 7B gate and per-head norm: whole rows per workgroup
    (5, 8, 2, torch.float32, "plain"),          # fp32 state, 4 query heads per group
    (5, 7, 1, torch.bfloat16, "70b_clamp"),     # first-layer 70B gate on a TP=4 shard
    (160, 28, 4, torch.bfloat16, "plain"),      # many units: whole rows per workgroup, no norm
    (4, 8, 4, torch.bfloat16, "7b_norm"),       # 2 query heads per group
    (3, 16, 2, torch.float32, "7b_norm"),       # 8 query heads per group, fp32 state with the fused norm
    (3, 6, 2, torch.bfloat16, "plain"),         # 3 query heads per group
])
def test_gla_decode_group_state_equals_per_head_state(B, H, Hkv, state_dtype, mode):
    from vllm_hymeta import _C, hip_ops

    rep = H // Hkv
    g_ = gen(21)
    slots = B + 2
    state_kv = (torch.randn(slots, Hkv, D, D, generator=g_) * 0.1).to(state_dtype)
    slot = torch.randperm(slots, generator=g_)[:B].to(torch.int32)
    slot[1] = -1
    kw = {}
    if mode == "7b_norm":
        kw = dict(gate_mode=_C.GATE_7B, out_norm_weight=(1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16).cuda())
    elif mode == "70b_clamp":
        kw = dict(gate_mode=_C.GATE_70B_CLAMP)
    s_group = state_kv.cuda()
    s_head = _expand(state_kv, rep).cuda()
    for step in range(3):
        q, k, v = randn_bf16(g_, B, H, D), randn_bf16(g_, B, Hkv, D), randn_bf16(g_, B, Hkv, D)
        if mode == "plain":
            g = (torch.nn.functional.logsigmoid(torch.randn(B, Hkv, D, generator=g_)) / 16).to(torch.bfloat16)
        elif mode == "7b_norm":
            g = randn_bf16(g_, B, Hkv, D, scale=2.0)
        else:
            g = None
        args = (q.cuda(), k.cuda(), v.cuda(), None if g is None else g.cuda())
        o_group = torch.full((B, H, D), 7.0, dtype=torch.bfloat16, device="cuda")
        o_head = torch.full((B, H, D), 7.0, dtype=torch.bfloat16, device="cuda")
        hip_ops.gla_decode(*args, s_group, slot.cuda(), out=o_group, **kw)
        hip_ops.gla_decode(*args, s_head, slot.cuda(), out=o_head, **kw)
        torch.cuda.synchronize()
        assert torch.equal(o_group, o_head), f"step {step}: outputs differ"
        assert torch.equal(_expand(s_group, rep), s_head), f"step {step}: states differ"
    assert (o_group[1] == 7.0).all()      # padded sequence untouched


def test_gla_group_state_rejects_the_lower_bound_gate():
    from vllm_hymeta import _C, hip_ops

    B, H, Hkv = 2, 7, 1
    q = torch.zeros(B, H, D, dtype=torch.bfloat16, device="cuda")
    kv = torch.zeros(B, Hkv, D, dtype=torch.bfloat16, device="cuda")
    st = torch.zeros(B, Hkv, D, D, dtype=torch.bfloat16, device="cuda")
    lb = torch.zeros(H * D, dtype=torch.float32, device="cuda")
    slot = torch.arange(B, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        hip_ops.gla_decode(q, kv, kv, None, st, slot, gate_mode=_C.GATE_70B_LOWER_BOUND, lower_bound=lb)
    with pytest.raises(ValueError):
        hip_ops.gla_prefill_varlen(q, kv, kv, None, st, slot[:1], torch.tensor([0, B], dtype=torch.int32, device="cuda"), B,
                                   gate_mode=_C.GATE_70B_LOWER_BOUND, lower_bound=lb)


@pytest.mark.parametrize("lens,H,Hkv,nseg,state_dtype,mode", [
    ([33, 64, 5], 28, 4, 1, torch.bfloat16, "plain"),      # single runs
    ([257, 40], 7, 1, 1, torch.float32, "plain"),
    ([600, 90], 4, 2, 3, torch.bfloat16, "plain"),         # segmented runs
    ([300, 31, 64], 28, 4, 1, torch.bfloat16, "7b_norm"),  # fused 7B gate and per-head norm
    ([700], 14, 2, 4, torch.bfloat16, "7b_norm"),
    ([130, 257], 7, 1, 2, torch.bfloat16, "70b_clamp"),    # first-layer 70B gate, first chunks start from the meta state
])
def test_gla_prefill_group_state_equals_per_head_state(lens, H, Hkv, nseg, state_dtype, mode):
    from vllm_hymeta import _C, hip_ops

    rep = H // Hkv
    T, nseq = sum(lens), len(lens)
    g_ = gen(22)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    kw_group, kw_head = {}, {}
    if mode == "plain":
        g = (torch.nn.functional.logsigmoid(torch.randn(T, Hkv, D, generator=g_)) / 16).to(torch.bfloat16).cuda()
    elif mode == "7b_norm":
        g = randn_bf16(g_, T, Hkv, D, scale=2.0).cuda()
        kw_group = dict(gate_mode=_C.GATE_7B, out_norm_weight=(1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16).cuda())
        kw_head = kw_group
    else:
        g = None
        meta = (torch.randn(Hkv, D, D, generator=g_) * 0.1).to(state_dtype)
        kw_group = dict(gate_mode=_C.GATE_70B_CLAMP, meta_state=meta.cuda())
        kw_head = dict(gate_mode=_C.GATE_70B_CLAMP, meta_state=_expand(meta[None], rep)[0].cuda())
    cu = torch.tensor(np.cumsum([0] + lens), dtype=torch.int32).cuda()
    slots = nseq + 2
    state_kv = (torch.randn(slots, Hkv, D, D, generator=g_) * 0.1).to(state_dtype)
    slot = torch.randperm(slots, generator=g_)[:nseq].to(torch.int32).cuda()
    # sequence 0 resumes from its state, the others start from zero (or the meta state)
    ctx = torch.tensor([17] + [0] * (nseq - 1), dtype=torch.int32).cuda()
    s_group, s_head = state_kv.cuda(), _expand(state_kv, rep).cuda()
    args = (q.cuda(), k.cuda(), v.cuda(), g)
    o_group = hip_ops.gla_prefill_varlen(*args, s_group, slot, cu, max(lens), context_lens=ctx, num_segments=nseg, **kw_group)
    o_head = hip_ops.gla_prefill_varlen(*args, s_head, slot, cu, max(lens), context_lens=ctx, num_segments=nseg, **kw_head)
    torch.cuda.synchronize()
    assert torch.equal(o_group, o_head)
    assert torch.equal(_expand(s_group, rep), s_head)
    if mode != "plain":
        return
    # and a decode step continues from either layout alike
    B = nseq
    qd, kd, vd = randn_bf16(g_, B, H, D).cuda(), randn_bf16(g_, B, Hkv, D).cuda(), randn_bf16(g_, B, Hkv, D).cuda()
    gd = (torch.nn.functional.logsigmoid(torch.randn(B, Hkv, D, generator=g_)) / 16).to(torch.bfloat16).cuda()
    d_group = hip_ops.gla_decode(qd, kd, vd, gd, s_group, slot)
    d_head = hip_ops.gla_decode(qd, kd, vd, gd, s_head, slot)
    torch.cuda.synchronize()
    assert torch.equal(d_group, d_head)


def test_gla_group_state_against_the_restatement():
    """One direct check of the per-kv-head layout against the CPU restatement (prefill, then a decode step)."""
    from oracle import gla_decode_step_ref, gla_prefill_varlen_ref
    from vllm_hymeta import hip_ops

    H, Hkv, T = 8, 2, 75
    rep = H // Hkv
    g_ = gen(23)
    q, k, v = randn_bf16(g_, T + 1, H, D), randn_bf16(g_, T + 1, Hkv, D), randn_bf16(g_, T + 1, Hkv, D)
    g = (torch.nn.functional.logsigmoid(torch.randn(T + 1, Hkv, D, generator=g_)) / 16).to(torch.bfloat16)
    slot = torch.tensor([1], dtype=torch.int32)
    ref_state = torch.zeros(2, H, D, D, dtype=torch.bfloat16)
    o_ref = gla_prefill_varlen_ref(q[:T], k[:T], v[:T], g[:T], ref_state, slot, [0, T], [0], None)
    sd = torch.zeros(2, Hkv, D, D, dtype=torch.bfloat16, device="cuda")
    out = hip_ops.gla_prefill_varlen(q[:T].cuda(), k[:T].cuda(), v[:T].cuda(), g[:T].cuda(), sd, slot.cuda(),
                                     torch.tensor([0, T], dtype=torch.int32).cuda(), T,
                                     context_lens=torch.tensor([0], dtype=torch.int32).cuda())
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), o_ref, "prefill, state per kv-head")
    assert_bf16_close(_expand(sd, rep).cpu(), ref_state, "prefill state per kv-head", rel=6e-3, frac=2.0 ** -5)
    ref_state = _expand(sd, rep).cpu()
    d_ref = gla_decode_step_ref(q[T:], k[T:], v[T:], g[T:], ref_state, slot)
    d = hip_ops.gla_decode(q[T:].cuda(), k[T:].cuda(), v[T:].cuda(), g[T:].cuda(), sd, slot.cuda())
    torch.cuda.synchronize()
    assert_bf16_close(d.cpu(), d_ref, "decode, state per kv-head")
