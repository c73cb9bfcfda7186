"""GLA varlen chunked prefill kernel vs the CPU restatement (fp64 recurrence), through the C ABI."""
import numpy as np
import os
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gate_inputs(seed, T, H, Hkv, g_dtype, strong=False):
    g_ = gen(seed)
    q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
    pre = torch.randn(T, Hkv, D, generator=g_)
    if strong:   # 70B-style decays: log(1 - min(sigmoid, .95)) in (-3, 0)
        g = torch.log(1 - torch.clamp(torch.sigmoid(pre * 2), max=0.95))
    else:        # 7B-style: logsigmoid / 16 in (-0.2, 0)
        g = torch.nn.functional.logsigmoid(pre) / 16
    return q, k, v, g.to(g_dtype)


def _state_close(dev, ref, what):
    # states are bf16 roundings of fp32 / fp64 accumulations of bf16 products
    assert_bf16_close(dev, ref, what, rel=6e-3, frac=2.0 ** -5)


@pytest.mark.parametrize("lens,H,Hkv,g_dtype,strong", [
    ([1, 16, 17, 130], 4, 4, torch.float32, False),
    ([1, 16, 17, 130], 4, 4, torch.float32, True),
    ([33, 64, 5], 28, 4, torch.bfloat16, False),
    ([257], 7, 1, torch.float32, True),
])
def test_gla_prefill_varlen(lens, H, Hkv, g_dtype, strong):
    from oracle import gla_prefill_varlen_ref
    from vllm_hymeta import hip_ops

    T = sum(lens)
    q, k, v, g = _gate_inputs(0, T, H, Hkv, g_dtype, strong)
    nseq = len(lens)
    cu = torch.tensor(np.cumsum([0] + lens), dtype=torch.int32)
    slots = nseq + 2
    g_ = gen(1)
    state = randn_bf16(g_, slots, H, D, D, scale=0.1)
    slot = torch.randperm(slots, generator=g_)[:nseq].to(torch.int32)
    if nseq > 2:
        slot[2] = -1
    ref_state = state.clone()
    o_ref = gla_prefill_varlen_ref(q, k, v, g, ref_state, slot, cu)   # resume from state[slot]
    sd = state.cuda()
    out = hip_ops.gla_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), g.cuda(), sd, slot.cuda(), cu.cuda(), max(lens))
    torch.cuda.synchronize()
    live = torch.ones(T, dtype=torch.bool)
    for i in range(nseq):
        if slot[i] < 0:
            live[cu[i]:cu[i + 1]] = False
    assert_bf16_close(out.cpu()[live], o_ref[live], "gla prefill out")
    _state_close(sd.cpu(), ref_state, "gla prefill state")


def test_gla_prefill_first_chunk_init_and_chunked_resume():
    """context_lens == 0 -> zeros (7B) or the shared meta state (70B); a prefill cut in two chunks
    equals the single pass."""
    from oracle import gla_prefill_varlen_ref
    from vllm_hymeta import hip_ops

    H, Hkv, T = 4, 2, 100
    q, k, v, g = _gate_inputs(3, T, H, Hkv, torch.float32, True)
    g_ = gen(4)
    meta = randn_bf16(g_, H, D, D, scale=0.1)
    stale = randn_bf16(g_, 2, H, D, D, scale=3.0)        # must be ignored on a first chunk
    slot = torch.tensor([1], dtype=torch.int32)
    for meta_state in (None, meta):
        ref_state = stale.clone()
        o_ref = gla_prefill_varlen_ref(q, k, v, g, ref_state, slot, [0, T], [0], meta_state)
        # single pass
        sd = stale.cuda()
        out = hip_ops.gla_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), g.cuda(), sd, slot.cuda(),
                                         torch.tensor([0, T], dtype=torch.int32).cuda(), T,
                                         context_lens=torch.tensor([0], dtype=torch.int32).cuda(),
                                         meta_state=None if meta_state is None else meta_state.cuda())
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu(), o_ref, "first chunk init")
        _state_close(sd.cpu()[1], ref_state[1], "first chunk state")
        assert torch.equal(sd.cpu()[0], stale[0])
        # two chunks: 37 + 63 tokens
        sd2 = stale.cuda()
        outs = []
        for (s, e, ctx) in ((0, 37, 0), (37, T, 37)):
            outs.append(hip_ops.gla_prefill_varlen(
                q[s:e].cuda(), k[s:e].cuda(), v[s:e].cuda(), g[s:e].cuda(), sd2, slot.cuda(),
                torch.tensor([0, e - s], dtype=torch.int32).cuda(), e - s,
                context_lens=torch.tensor([ctx], dtype=torch.int32).cuda(),
                meta_state=None if meta_state is None else meta_state.cuda()))
        torch.cuda.synchronize()
        assert_bf16_close(torch.cat(outs).cpu(), o_ref, "chunked resume", rel=8e-3)
        _state_close(sd2.cpu()[1], ref_state[1], "chunked resume state")


def test_gla_prefill_then_decode_continues_the_sequence():
    from oracle import gla_recurrent_ref
    from vllm_hymeta import hip_ops

    H, Hkv, T = 4, 4, 70
    q, k, v, g = _gate_inputs(5, T + 3, H, Hkv, torch.float32, False)
    o_ref, _ = gla_recurrent_ref(q, k, v, g)
    sd = torch.zeros(1, H, D, D, dtype=torch.float32, device="cuda")   # fp32 state option
    slot = torch.zeros(1, dtype=torch.int32, device="cuda")
    o_pre = hip_ops.gla_prefill_varlen(q[:T].cuda(), k[:T].cuda(), v[:T].cuda(), g[:T].cuda(), sd, slot,
                                       torch.tensor([0, T], dtype=torch.int32).cuda(), T)
    outs = [o_pre]
    for t in range(T, T + 3):
        outs.append(hip_ops.gla_decode(q[t:t + 1].cuda(), k[t:t + 1].cuda(), v[t:t + 1].cuda(),
                                       g[t:t + 1].cuda(), sd, slot))
    torch.cuda.synchronize()
    assert_bf16_close(torch.cat(outs).cpu(), o_ref, "prefill + decode continuation", rel=8e-3)


def test_gla_prefill_golden_fused_gates():
    from vllm_hymeta import hip_ops, _C

    z = np.load(os.path.join(GOLD, "gla_prefill.npz"))

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16)

    q, k, v, gk = (bf(z[n]).cuda() for n in ("q", "k", "v", "gk"))
    cu = torch.from_numpy(z["cu_seqlens"]).cuda()
    slot = torch.from_numpy(z["slot_idx"]).cuda()
    ctx = torch.from_numpy(z["context_lens"]).cuda()
    live = torch.ones(q.shape[0], dtype=torch.bool)
    live[int(z["cu_seqlens"][2]):int(z["cu_seqlens"][3])] = False   # slot -1
    maxlen = int(np.diff(z["cu_seqlens"]).max())
    # 7B gate
    sd = bf(z["state0"]).cuda()
    out = hip_ops.gla_prefill_varlen(q, k, v, gk, sd, slot, cu, maxlen, context_lens=ctx, gate_mode=_C.GATE_7B)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu()[live], bf(z["o_7b"])[live], "golden 7B gate")
    _state_close(sd.cpu(), bf(z["state_7b"]), "golden 7B state")
    # 70B gates with the shared meta state
    lb = torch.from_numpy(z["lower_bound"]).cuda()
    meta = bf(z["meta_state"]).cuda()
    for name, mode in (("clamp", _C.GATE_70B_CLAMP), ("lb", _C.GATE_70B_LOWER_BOUND)):
        sd = bf(z["state0"]).cuda()
        out = hip_ops.gla_prefill_varlen(q, k, v, None, sd, slot, cu, maxlen, context_lens=ctx, meta_state=meta,
                                         gate_mode=mode, lower_bound=lb)
        torch.cuda.synchronize()
        assert_bf16_close(out.cpu()[live], bf(z[f"o_70b_{name}"])[live], f"golden 70B {name} gate")
        _state_close(sd.cpu(), bf(z[f"state_70b_{name}"]), f"golden 70B {name} state")


def test_gla_prefill_fused_norm():
    from oracle import gla_prefill_varlen_ref
    from oracle.layers import rms_norm_ref
    from vllm_hymeta import hip_ops

    H, Hkv, T = 4, 2, 45
    q, k, v, g = _gate_inputs(7, T, H, Hkv, torch.bfloat16, False)
    w = (1 + 0.1 * torch.randn(D, generator=gen(8))).to(torch.bfloat16)
    st = torch.zeros(1, H, D, D, dtype=torch.bfloat16)
    o_ref = rms_norm_ref(gla_prefill_varlen_ref(q, k, v, g, st, torch.tensor([0]), [0, T]), w, 1e-6)
    out = hip_ops.gla_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), g.cuda(), torch.zeros_like(st).cuda(),
                                     torch.zeros(1, dtype=torch.int32).cuda(),
                                     torch.tensor([0, T], dtype=torch.int32).cuda(), T,
                                     out_norm_weight=w.cuda(), out_norm_eps=1e-6)
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), o_ref, "gla prefill + g_norm", rel=8e-3)


@pytest.mark.parametrize("nseg", [2, 5, 0])
@pytest.mark.parametrize("mode", ["plain", "meta_norm_f32state"])
def test_gla_prefill_segmented_runs_equal_oracle(nseg, mode):
    """Sequences cut into parallel segments (state contributions, scan, outputs): ragged lengths so
    that some sequences own fewer segments than the grid provides, one skipped slot, resume from a
    stored state / the meta state, fused 7B gate + norm."""
    from oracle import gla_prefill_varlen_ref
    from oracle.layers import gate_7b_ref, rms_norm_ref
    from vllm_hymeta import _C, hip_ops

    lens = [700, 33, 1, 1024, 290]
    H, Hkv = 4, 2
    T, nseq = sum(lens), len(lens)
    cu = torch.tensor(np.cumsum([0] + lens), dtype=torch.int32)
    g_ = gen(11)
    slot = torch.tensor([3, 0, -1, 5, 1], dtype=torch.int32)
    if mode == "plain":
        q, k, v, g = _gate_inputs(10, T, H, Hkv, torch.float32, True)
        state = randn_bf16(g_, 6, H, D, D, scale=0.1)
        ref_state = state.clone()
        o_ref = gla_prefill_varlen_ref(q, k, v, g, ref_state, slot, cu)
        sd = state.cuda()
        out = hip_ops.gla_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), g.cuda(), sd, slot.cuda(), cu.cuda(), max(lens),
                                         num_segments=nseg)
    else:
        q, k, v = randn_bf16(g_, T, H, D), randn_bf16(g_, T, Hkv, D), randn_bf16(g_, T, Hkv, D)
        gk = randn_bf16(g_, T, Hkv, D)
        w = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16)
        meta = (0.1 * torch.randn(H, D, D, generator=g_)).float()
        state = torch.full((6, H, D, D), 7.0)
        ctx = torch.zeros(nseq, dtype=torch.int32)
        rep = H // Hkv
        qg, kg, gg = gate_7b_ref(q, k.repeat_interleave(rep, 1), gk.repeat_interleave(rep, 1))
        ref_state = state.clone()
        o_ref = gla_prefill_varlen_ref(qg, kg, v.repeat_interleave(rep, 1), gg, ref_state, slot, cu, ctx, meta)
        o_ref = rms_norm_ref(o_ref.to(torch.bfloat16), w, 1e-5).float()
        sd = state.cuda()
        out = hip_ops.gla_prefill_varlen(q.cuda(), k.cuda(), v.cuda(), gk.cuda(), sd, slot.cuda(), cu.cuda(), max(lens),
                                         context_lens=ctx.cuda(), meta_state=meta.cuda(), gate_mode=_C.GATE_7B,
                                         out_norm_weight=w.cuda(), out_norm_eps=1e-5, num_segments=nseg)
    torch.cuda.synchronize()
    live = torch.ones(T, dtype=torch.bool)
    live[cu[2]:cu[3]] = False
    assert_bf16_close(out.cpu()[live].float(), o_ref[live].float(), f"segmented gla prefill out nseg={nseg}", rel=8e-3)
    _state_close(sd.cpu().float(), ref_state.float(), "segmented gla prefill state")
    assert torch.equal(sd.cpu()[2], state[2]) and torch.equal(sd.cpu()[4], state[4])   # untouched slots
