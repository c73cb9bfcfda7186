"""BASELINE-sized cases (8k / 32k contexts, 7B and 70B-TP4 head geometry) checked through
size-independent properties, where the fp64 oracle would take minutes: invariance under
segmentation / chunking / split count, prefill-decode consistency, and linearity in V."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128


def _cuda_randn(g_, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g_, device="cuda") * scale).to(torch.bfloat16)


def test_gla_prefill_32k_segmentation_invariance_and_decode_continuation():
    """7B geometry, one 32k prompt: 9 parallel segments == single run (fp32 summation order only),
    a prompt cut into 4 chunked-prefill calls == one call, and the decode step that follows the
    prefill == one more prefill token."""
    from vllm_hymeta import _C, hip_ops

    H, Hkv, L = 28, 4, 32768
    g_ = torch.Generator(device="cuda").manual_seed(0)
    q, k, v, gk = (_cuda_randn(g_, L + 1, h, D) for h in (H, Hkv, Hkv, Hkv))
    w = (1 + 0.1 * torch.randn(D, generator=g_, device="cuda")).to(torch.bfloat16)
    slot = torch.zeros(1, dtype=torch.int32, device="cuda")
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")

    def run(n_tok, nseg, state, ctx, t0=0):
        cu = torch.tensor([0, n_tok], dtype=torch.int32, device="cuda")
        return hip_ops.gla_prefill_varlen(q[t0:t0 + n_tok], k[t0:t0 + n_tok], v[t0:t0 + n_tok], gk[t0:t0 + n_tok], state,
                                          slot, cu, n_tok, context_lens=ctx, gate_mode=_C.GATE_7B,
                                          out_norm_weight=w, out_norm_eps=1e-6, num_segments=nseg)

    s_one = torch.full((1, H, D, D), 3.0, dtype=torch.bfloat16, device="cuda")     # stale: first chunk ignores it
    o_one = run(L, 1, s_one, zero)
    s_seg = torch.full((1, H, D, D), 3.0, dtype=torch.bfloat16, device="cuda")
    o_seg = run(L, 9, s_seg, zero)
    assert_bf16_close(o_seg.float().cpu(), o_one.float().cpu(), "32k: 9 segments vs single run", rel=2e-3)
    assert_bf16_close(s_seg.float().cpu(), s_one.float().cpu(), "32k: state, segments vs single", rel=2e-3, frac=2.0 ** -6)

    s_chk = torch.zeros(1, H, D, D, dtype=torch.bfloat16, device="cuda")
    outs = []
    for i in range(4):
        ctx = torch.tensor([i * 8192], dtype=torch.int32, device="cuda")
        outs.append(run(8192, 0, s_chk, ctx, t0=i * 8192))
    # a chunk boundary rounds the carried state to bf16, the single run keeps it in fp32
    assert_bf16_close(torch.cat(outs).float().cpu(), o_one.float().cpu(), "32k: 4 chunks vs single run", rel=6e-3)

    # decode step L == prefill of L + 1 tokens, last row
    s_ref = torch.zeros(1, H, D, D, dtype=torch.bfloat16, device="cuda")
    o_ref = run(L + 1, 1, s_ref, zero)
    o_dec = hip_ops.gla_decode(q[L:], k[L:], v[L:], gk[L:], s_one, slot, gate_mode=_C.GATE_7B, out_norm_weight=w,
                               out_norm_eps=1e-6)
    torch.cuda.synchronize()
    assert_bf16_close(o_dec.float().cpu(), o_ref[L:].float().cpu(), "decode after 32k prefill", rel=6e-3)
    assert_bf16_close(s_one.float().cpu(), s_ref.float().cpu(), "state after decode", rel=6e-3, frac=2.0 ** -5)


@pytest.mark.parametrize("H,Hkv,M,window", [(28, 4, 0, 4097), (7, 1, 128, 4096)])   # 7B TP1; 70B TP4 (window incl. self)
def test_swa_32k_chunked_paged_prefill_and_decode_agree_with_single_pass(H, Hkv, M, window):
    """One 32k prompt: un-paged single pass == 4 chunks attended through the block table (context +
    chunk, bottom-right aligned); the paged decode of position L (any split count) == the prefill
    row of that position; attention is linear in V."""
    from vllm_hymeta import hip_ops

    L, bs = 32768, 256
    wl = window - 1
    g_ = torch.Generator(device="cuda").manual_seed(1)
    q, k, v = _cuda_randn(g_, L + 1, H, D), _cuda_randn(g_, L + 1, Hkv, D), _cuda_randn(g_, L + 1, Hkv, D)
    k2 = _cuda_randn(g_, M, Hkv, D) if M else None
    v2 = _cuda_randn(g_, M, Hkv, D) if M else None
    cu = torch.tensor([0, L + 1], dtype=torch.int32, device="cuda")
    ref = hip_ops.meta_swa_prefill(q, cu, L + 1, key=k, value=v, cu_seqlens_k=cu, meta_key=k2, meta_value=v2,
                                   window_left=wl)
    ref = ref[0] if isinstance(ref, tuple) else ref
    ref = ref.view(-1, H, D)[-(L + 1):]

    nblk = (L + 1 + bs - 1) // bs
    perm = torch.randperm(nblk + 2, generator=torch.Generator().manual_seed(2))[:nblk].to(torch.int32)
    bt = perm.view(1, -1).cuda()
    kv = torch.zeros(2, nblk + 2, bs, Hkv, D, dtype=torch.bfloat16, device="cuda")
    pos = torch.arange(L + 1, device="cuda")
    slots = bt[0].long()[pos // bs] * bs + pos % bs
    outs = []
    for i in range(4):
        lo, hi = i * 8192, (i + 1) * 8192
        hip_ops.reshape_and_cache(k[lo:hi], v[lo:hi], kv[0], kv[1], slots[lo:hi])
        cuq = torch.tensor([0, hi - lo], dtype=torch.int32, device="cuda")
        o = hip_ops.meta_swa_prefill(q[lo:hi], cuq, hi - lo, key_cache=kv[0], value_cache=kv[1], block_tables=bt,
                                     seq_lens_k=torch.tensor([hi], dtype=torch.int32, device="cuda"),
                                     meta_key=k2, meta_value=v2, window_left=wl)
        o = o[0] if isinstance(o, tuple) else o
        outs.append(o.view(-1, H, D))
    got = torch.cat(outs)
    torch.cuda.synchronize()
    # same keys, same tiles relative to position 0 -> the two paths differ only in tile phase
    assert_bf16_close(got.float().cpu(), ref[:L].float().cpu(), "32k chunked paged prefill vs single pass", rel=2e-3)

    lens = torch.tensor([L + 1], dtype=torch.int32, device="cuda")
    for splits in (1, 8, 0):
        kv_d = kv.clone()
        o_dec = hip_ops.meta_swa_paged_decode(q[L:], kv_d[0], kv_d[1], lens, bt, meta_key=k2, meta_value=v2, window_left=wl,
                                              num_splits=splits, new_key=k[L:], new_value=v[L:], slot_mapping=slots[L:])
        torch.cuda.synchronize()
        # two bf16 kernels against each other (each within 5e-3 of the fp64 oracle at test sizes): both
        # round their P and their output, so the pair sits near sqrt(2) x one kernel's error
        assert_bf16_close(o_dec.float().cpu(), ref[L:].float().cpu(), f"decode at 32k, splits={splits}", rel=4e-3)
    assert torch.equal(kv_d[0].view(-1, Hkv, D)[slots[L]], k[L])      # the fused write landed

    # linearity in V: attn(V1 + V2) == attn(V1) + attn(V2) up to bf16 rounding of the three outputs
    va, vb = v, _cuda_randn(g_, L + 1, Hkv, D)
    def attn(vv, vm):
        o = hip_ops.meta_swa_prefill(q[-2048:], torch.tensor([0, 2048], dtype=torch.int32, device="cuda"), 2048,
                                     key=k, value=vv, cu_seqlens_k=cu, meta_key=k2, meta_value=vm, window_left=wl)
        o = o[0] if isinstance(o, tuple) else o
        return o.view(-1, H, D)[-2048:].float()
    vsum = (va.float() + vb.float()).to(torch.bfloat16)
    v2b = _cuda_randn(g_, M, Hkv, D) if M else None
    v2s = (v2.float() + v2b.float()).to(torch.bfloat16) if M else None
    lhs = attn(vsum, v2s)
    rhs = attn(va, v2) + attn(vb, v2b)
    torch.cuda.synchronize()
    assert_bf16_close(lhs.cpu(), rhs.cpu(), "linearity in V", rel=8e-3, frac=2.0 ** -5)


@pytest.mark.parametrize("T", [64, 256])
def test_expert_mixture_at_the_tp4_rank_size_against_fp32_torch(T):
    """The sparse mixture at the size one TP=4 rank of the 70B model runs it (16 experts, hidden 3584, a quarter of the
    18944-wide experts, top-1 with the softmax probability as weight, shared expert added; 64 tokens = the 16-row tiles of
    the whole-line expert GEMM, 256 tokens = its 64-row tiles), router inside the routing launch, against plain fp32
    torch GEMMs on the GPU with the block's rounding points (every projection, the activation and the weighted expert
    output rounded to bf16; the routing on the logits the block itself routed)."""
    from vllm_hymeta import hip_ops

    dev = torch.device("cuda")
    g_ = torch.Generator(device=dev).manual_seed(77 + T)
    E, H, inter = 16, 3584, 18944 // 4
    rnd = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g_, device=dev) * scale).to(torch.bfloat16)
    x, gate = rnd(T, H), rnd(E, H, scale=0.02)
    w13, w2 = rnd(E, 2 * inter, H, scale=H ** -0.5), rnd(E, H, inter, scale=inter ** -0.5)
    shared = rnd(T, H)
    logits = torch.empty(T, E, dtype=torch.bfloat16, device=dev)
    assert hip_ops.moe_router_fused(x, gate)
    out = hip_ops.moe_forward(x, None, w13, w2, top_k=1, shared=shared, gate_weight=gate, logits_out=logits)
    torch.cuda.synchronize()

    lin = torch.nn.functional.linear(x.float(), gate.float())
    assert float((logits.float() - lin).abs().max()) <= 2.0 ** -7 * float(lin.abs().max()) + 2e-5      # one bf16 ulp
    probs = torch.softmax(logits.float(), dim=-1)
    w, ids = probs.max(dim=-1)
    assert ids.unique().numel() >= 8, "the test is meant to spread its tokens over the experts"
    ref = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
    for e in ids.unique().tolist():
        rows = (ids == e).nonzero().squeeze(1)
        h = (x[rows].float() @ w13[e].float().t()).to(torch.bfloat16)
        act = (torch.nn.functional.silu(h[:, :inter].float()) * h[:, inter:].float()).to(torch.bfloat16)
        y = ((act.float() @ w2[e].float().t()) * w[rows].unsqueeze(1)).to(torch.bfloat16)
        ref[rows] = (y.float() + shared[rows].float()).to(torch.bfloat16)
    assert_bf16_close(out.cpu(), ref.cpu(), f"expert mixture, {T} tokens at the TP=4 rank size", rel=8e-3)
