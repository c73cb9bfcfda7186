"""Whole attention layers on the GPU vs the eager restatement: the 7B GLA and sliding-window layers
(BASELINE config 1: random-init single layer, seq_len 512), a GPTQ-quantised GLA layer, and the
70B intra-hybrid block."""
import types

import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128


def _lin(g_, out_f, in_f, bias):
    from oracle.layers import xavier_linear_params_ref
    return xavier_linear_params_ref(g_, out_f, in_f, bias)


def _md(dev, *, prefill_lens=(), context_lens=(), decode_seq_lens=(), block_tables=None, slot_mapping=None):
    """Minimal step metadata (fields of MetaAttentionMetadata) for a batch of whole prompts and/or decodes."""
    from vllm_hymeta.attention.meta_attention import MetaAttentionMetadata

    n_pre, n_dec = len(prefill_lens), len(decode_seq_lens)
    qlens = list(prefill_lens) + [1] * n_dec
    ctx = list(context_lens) + [L - 1 for L in decode_seq_lens]
    seq_lens = [c + q for c, q in zip(ctx, qlens)]
    cu = lambda xs: torch.tensor([0] + list(torch.tensor(xs).cumsum(0)) if xs else [0], dtype=torch.int32, device=dev)
    T = sum(qlens)
    return MetaAttentionMetadata(
        num_prefills=n_pre, num_prefill_tokens=sum(prefill_lens), num_decode_tokens=n_dec,
        slot_mapping=slot_mapping if slot_mapping is not None else torch.full((T,), -1, dtype=torch.long, device=dev),
        multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=True, seq_lens=seq_lens,
        seq_lens_tensor=torch.tensor(seq_lens, dtype=torch.int32, device=dev),
        max_prefill_seq_len=max([c + q for c, q in zip(ctx[:n_pre], qlens[:n_pre])], default=0),
        max_decode_seq_len=max(decode_seq_lens, default=0),
        context_lens_tensor=torch.tensor(ctx, dtype=torch.int32, device=dev),
        block_tables=block_tables if block_tables is not None else torch.zeros(len(qlens), 0, dtype=torch.int32, device=dev),
        use_cuda_graph=False, max_query_len=max(qlens), max_decode_query_len=1,
        query_start_loc=cu(qlens), seq_start_loc=cu(seq_lens))


def _gla_params(g_, hidden, H, Hkv):
    p = {}
    for name, out_f, bias in (("q_proj", H * D, True), ("k_proj", Hkv * D, True), ("v_proj", Hkv * D, True)):
        for k, v in _lin(g_, out_f, hidden, bias).items():
            p[f"{name}.{k}"] = v
    p["gk_proj.0.weight"] = _lin(g_, 16, hidden, False)["weight"]
    gk1 = _lin(g_, Hkv * D, 16, True)
    p["gk_proj.1.weight"], p["gk_proj.1.bias"] = gk1["weight"], gk1["bias"]
    p["g_norm.weight"] = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16)
    p["o_proj.weight"] = _lin(g_, hidden, H * D, False)["weight"]
    # non-zero biases exercise the bias path (the eager init zeroes them)
    for n in ("q_proj.bias", "k_proj.bias", "v_proj.bias", "gk_proj.1.bias"):
        p[n] = randn_bf16(g_, *p[n].shape, scale=0.1)
    return p


@pytest.mark.parametrize("hidden,H,Hkv", [(512, 4, 2), (3584, 28, 4)])
def test_config1_7b_gla_layer_prefill_512_then_decode(hidden, H, Hkv):
    from oracle.layers import gla_layer_7b_ref
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.gla_attention import GatedLinearAttention

    dev = torch.device("cuda")
    g_ = gen(0)
    T = 512
    p = _gla_params(g_, hidden, H, Hkv)
    x = randn_bf16(g_, T + 1, hidden)
    ref, _ = gla_layer_7b_ref(x, p, H, Hkv)
    qkv_w = torch.cat([p["q_proj.weight"], p["k_proj.weight"], p["v_proj.weight"]]).to(dev)
    qkv_b = torch.cat([p["q_proj.bias"], p["k_proj.bias"], p["v_proj.bias"]]).to(dev)
    layer = GatedLinearAttention(
        hidden, H, Hkv, HymetaLinear.from_weight(qkv_w, qkv_b), HymetaLinear.from_weight(p["gk_proj.0.weight"].to(dev)),
        HymetaLinear.from_weight(p["gk_proj.1.weight"].to(dev), p["gk_proj.1.bias"].to(dev)),
        HymetaLinear.from_weight(p["o_proj.weight"].to(dev)), p["g_norm.weight"].to(dev))
    cache = torch.full((3, H, D, D), 9.0, dtype=torch.bfloat16, device=dev)   # stale: must be re-initialised
    slots = torch.tensor([2], dtype=torch.int32, device=dev)
    out_pre = layer(x[:T].to(dev), cache, slots, _md(dev, prefill_lens=[T], context_lens=[0]))
    out_dec = layer(x[T:].to(dev), cache, slots, _md(dev, decode_seq_lens=[T + 1]))
    torch.cuda.synchronize()
    assert_bf16_close(torch.cat([out_pre, out_dec]).cpu(), ref, f"7B GLA layer hidden={hidden}", rel=1e-2)


@pytest.mark.parametrize("hidden,H,Hkv,W", [(512, 4, 2, 64), (3584, 28, 4, 200)])
def test_config1_7b_swa_layer_prefill_512_then_paged_decode(hidden, H, Hkv, W):
    from oracle.layers import rope_cos_sin_cache_ref, swa_layer_7b_ref
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.window_attention import SlidingWindowAttention, rope_cos_sin_cache

    dev = torch.device("cuda")
    g_ = gen(1)
    T, bs = 512, 16
    p = {}
    for name, out_f, bias in (("q_proj", H * D, True), ("k_proj", Hkv * D, True), ("v_proj", Hkv * D, True),
                              ("o_proj", hidden, False)):
        for k, v in _lin(g_, out_f, H * D if name == "o_proj" else hidden, bias).items():
            p[f"{name}.{k}"] = v
    for n in ("q_proj.bias", "k_proj.bias", "v_proj.bias"):
        p[n] = randn_bf16(g_, *p[n].shape, scale=0.1)
    x = randn_bf16(g_, T + 1, hidden)
    pos = torch.arange(T + 1)
    ref = swa_layer_7b_ref(x, pos, p, H, Hkv, W, max_position=1024)
    table = rope_cos_sin_cache(D, 1024, 1e6, device=dev)
    assert torch.equal(table.cpu(), rope_cos_sin_cache_ref(D, 1024, 1e6))
    qkv_w = torch.cat([p["q_proj.weight"], p["k_proj.weight"], p["v_proj.weight"]]).to(dev)
    qkv_b = torch.cat([p["q_proj.bias"], p["k_proj.bias"], p["v_proj.bias"]]).to(dev)
    layer = SlidingWindowAttention(hidden, H, Hkv, W, HymetaLinear.from_weight(qkv_w, qkv_b),
                                   HymetaLinear.from_weight(p["o_proj.weight"].to(dev)), table)
    nblk = (T + 1 + bs - 1) // bs + 1
    kv = torch.zeros(2, nblk, bs, Hkv, D, dtype=torch.bfloat16, device=dev)
    bt = torch.randperm(nblk, generator=g_)[: nblk - 1].to(torch.int32).view(1, -1).to(dev)
    slot = lambda t: (bt[0, t // bs].long() * bs + t % bs)
    slots = torch.stack([slot(t) for t in range(T + 1)])
    md_pre = _md(dev, prefill_lens=[T], context_lens=[0], block_tables=torch.zeros(1, 0, dtype=torch.int32, device=dev),
                 slot_mapping=slots[:T])
    out_pre = layer(x[:T].to(dev), pos[:T].to(dev), kv, md_pre)
    md_dec = _md(dev, decode_seq_lens=[T + 1], block_tables=bt, slot_mapping=slots[T:])
    out_dec = layer(x[T:].to(dev), pos[T:].to(dev), kv, md_dec)
    torch.cuda.synchronize()
    assert_bf16_close(torch.cat([out_pre, out_dec]).cpu(), ref, f"7B SWA layer hidden={hidden}", rel=1e-2)


def test_7b_gla_layer_with_gptq_linears():
    """Same layer with 4-bit GPTQ qkv / o projections (gk_proj stays bf16, as in the reference's
    module list): compared with the eager layer run on the de-quantised weights."""
    from oracle.layers import gla_layer_7b_ref
    from oracle.quant import gptq_dequant_ref, gptq_pack_ref
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.gla_attention import GatedLinearAttention

    dev = torch.device("cuda")
    g_ = gen(2)
    hidden, H, Hkv, T, group = 512, 4, 2, 70, 128
    p = _gla_params(g_, hidden, H, Hkv)

    def quantise(w):   # w [out, in] -> GPTQ tensors of W^T [in, out]
        K, N = w.shape[1], w.shape[0]
        wi = torch.randint(0, 16, (K, N), generator=g_)
        z = torch.randint(1, 16, (K // group, N), generator=g_)
        s = (torch.rand(K // group, N, generator=g_) * 0.01 + 0.002).to(torch.float16)
        qw, qz = gptq_pack_ref(wi, z, 4)
        wd = gptq_dequant_ref(qw, qz, s, 4, group).t().contiguous().to(torch.bfloat16)
        return (qw, qz, s), wd

    qkv_w = torch.cat([p["q_proj.weight"], p["k_proj.weight"], p["v_proj.weight"]])
    (qw1, qz1, s1), qkv_d = quantise(qkv_w)
    (qw2, qz2, s2), o_d = quantise(p["o_proj.weight"])
    p["q_proj.weight"], p["k_proj.weight"], p["v_proj.weight"] = qkv_d[: H * D], qkv_d[H * D:(H + Hkv) * D], qkv_d[(H + Hkv) * D:]
    p["o_proj.weight"] = o_d
    x = randn_bf16(g_, T, hidden)
    ref, _ = gla_layer_7b_ref(x, p, H, Hkv)
    qkv_b = torch.cat([p["q_proj.bias"], p["k_proj.bias"], p["v_proj.bias"]]).to(dev)
    layer = GatedLinearAttention(
        hidden, H, Hkv, HymetaLinear.from_gptq(qw1.to(dev), qz1.to(dev), s1.to(dev), 4, group, bias=qkv_b),
        HymetaLinear.from_weight(p["gk_proj.0.weight"].to(dev)),
        HymetaLinear.from_weight(p["gk_proj.1.weight"].to(dev), p["gk_proj.1.bias"].to(dev)),
        HymetaLinear.from_gptq(qw2.to(dev), qz2.to(dev), s2.to(dev), 4, group), p["g_norm.weight"].to(dev))
    cache = torch.zeros(1, H, D, D, dtype=torch.bfloat16, device=dev)
    out = layer(x.to(dev), cache, torch.zeros(1, dtype=torch.int32, device=dev), _md(dev, prefill_lens=[T], context_lens=[0]))
    torch.cuda.synchronize()
    assert_bf16_close(out.cpu(), ref, "7B GLA layer, GPTQ linears", rel=1.5e-2)


@pytest.mark.parametrize("layer_idx,window", [(0, 48), (1, 48), (3, None)])
def test_70b_intra_hybrid_block(layer_idx, window):
    from oracle.layers import hybrid_layer_70b_ref, lower_bounds_ref
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.window_attention import rope_cos_sin_cache
    from vllm_hymeta.models.hybrid_attention import HLinearAttention, IntraHybridAttention, MetaFlashAttention

    dev = torch.device("cuda")
    g_ = gen(3)
    hidden, H, Hkv, M, T, bs = 512, 4, 1, 32, 90, 16
    p = {}
    for br in ("vanilla_attn", "linear_attn"):
        for name, out_f in (("q_proj", H * D), ("k_proj", Hkv * D), ("v_proj", Hkv * D)):
            lin = _lin(g_, out_f, hidden, True)
            p[f"{br}.{name}.weight"] = lin["weight"]
            p[f"{br}.{name}.bias"] = randn_bf16(g_, out_f, scale=0.1)
    p["out_proj.weight"] = _lin(g_, hidden, H * D, False)["weight"]
    p["norm1.weight"] = 1 + 0.1 * torch.randn(hidden, generator=g_)
    p["norm2.weight"] = 1 + 0.1 * torch.randn(hidden, generator=g_)
    lbs = lower_bounds_ref(torch.randn(7, hidden, generator=g_))
    lb = lbs[layer_idx % 7].view(H, D)
    x, meta_x = randn_bf16(g_, T, hidden), randn_bf16(g_, M, hidden)
    pos = torch.arange(T)
    ref = hybrid_layer_70b_ref(x, meta_x, pos, p, H, Hkv, window, lb, first_layer=(layer_idx % 7 == 0), max_position=1024)

    def fused(br):
        w = torch.cat([p[f"{br}.q_proj.weight"], p[f"{br}.k_proj.weight"], p[f"{br}.v_proj.weight"]]).to(dev)
        b = torch.cat([p[f"{br}.q_proj.bias"], p[f"{br}.k_proj.bias"], p[f"{br}.v_proj.bias"]]).to(dev)
        return HymetaLinear.from_weight(w, b)

    table = rope_cos_sin_cache(D, 1024, 1e6, device=dev)
    attn = MetaFlashAttention(hidden, H, Hkv, fused("vanilla_attn"), table, window, num_meta_tokens=M)
    lin = HLinearAttention(hidden, H, Hkv, fused("linear_attn"), layer_idx, num_meta_tokens=M)
    block = IntraHybridAttention(attn, lin, HymetaLinear.from_weight(p["out_proj.weight"].to(dev)),
                                 p["norm1.weight"].to(dev), p["norm2.weight"].to(dev))
    meta_kv = torch.zeros(2, M, Hkv, D, dtype=torch.bfloat16, device=dev)
    meta_lin = torch.zeros(H, D, D, dtype=torch.bfloat16, device=dev)
    attn.prepare_meta_state(meta_x.to(dev), meta_kv)
    lin.prepare_meta_state(meta_x.to(dev), meta_lin, lb.to(dev))
    nblk = (T + bs - 1) // bs
    kv = torch.zeros(2, nblk + 1, bs, Hkv, D, dtype=torch.bfloat16, device=dev)
    bt = torch.randperm(nblk + 1, generator=g_)[:nblk].to(torch.int32).view(1, -1).to(dev)
    slots = torch.stack([bt[0, t // bs].long() * bs + t % bs for t in range(T)])
    # first chunk: 60 tokens as a whole prompt; second chunk: 30 tokens of chunked prefill through the cache
    state = torch.full((2, H, D, D), 5.0, dtype=torch.bfloat16, device=dev)
    sidx = torch.tensor([1], dtype=torch.int32, device=dev)
    md1 = _md(dev, prefill_lens=[60], context_lens=[0], slot_mapping=slots[:60],
              block_tables=torch.zeros(1, 0, dtype=torch.int32, device=dev))
    y1 = block(x[:60].to(dev), pos[:60].to(dev), kv, state, meta_lin, meta_kv, sidx, md1, lb.to(dev))
    md2 = _md(dev, prefill_lens=[30], context_lens=[60], slot_mapping=slots[60:], block_tables=bt)
    y2 = block(x[60:].to(dev), pos[60:].to(dev), kv, state, meta_lin, meta_kv, sidx, md2, lb.to(dev))
    torch.cuda.synchronize()
    assert_bf16_close(torch.cat([y1, y2]).cpu(), ref, f"70B hybrid block layer {layer_idx} window {window}", rel=1.2e-2)


def test_70b_block_two_head_shards_equal_unsharded():
    """The block run as two tensor-parallel shards (one process, the variance and out-projection
    reductions done by hand) equals the unsharded block: checks the head / weight / lower-bound
    slicing and the statistics path of the combine kernel."""
    from oracle.layers import lower_bounds_ref
    from vllm_hymeta import hip_ops
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.window_attention import rope_cos_sin_cache
    from vllm_hymeta.models.hybrid_attention import HLinearAttention, IntraHybridAttention, MetaFlashAttention
    from vllm_hymeta.parallel import shard_heads

    dev = torch.device("cuda")
    g_ = gen(4)
    hidden, H, Hkv, M, T, bs, layer_idx, window = 512, 4, 2, 32, 50, 16, 2, 24
    w = {br: {n: _lin(g_, o, hidden, True)["weight"].to(dev) for n, o in (("q", H * D), ("k", Hkv * D), ("v", Hkv * D))}
         for br in ("vanilla_attn", "linear_attn")}
    b = {br: {n: randn_bf16(g_, o, scale=0.1).to(dev) for n, o in (("q", H * D), ("k", Hkv * D), ("v", Hkv * D))}
         for br in ("vanilla_attn", "linear_attn")}
    w_out = _lin(g_, hidden, H * D, False)["weight"].to(dev)
    n1 = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(dev)
    n2 = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(dev)
    lb = lower_bounds_ref(torch.randn(7, hidden, generator=g_))[layer_idx].to(dev)
    x, meta_x = randn_bf16(g_, T, hidden).to(dev), randn_bf16(g_, M, hidden).to(dev)
    pos = torch.arange(T, device=dev)
    table = rope_cos_sin_cache(D, 1024, 1e6, device=dev)
    slots = torch.arange(T, device=dev)
    md = _md(dev, prefill_lens=[T], context_lens=[0], slot_mapping=slots)

    def build(tp, rank):
        s = shard_heads(H, Hkv, tp, rank)
        hs = slice(s.head_start * D, (s.head_start + s.num_heads) * D)
        ks = slice(s.kv_head_start * D, (s.kv_head_start + s.num_kv_heads) * D)
        fused = lambda br: HymetaLinear.from_weight(
            torch.cat([w[br]["q"][hs], w[br]["k"][ks], w[br]["v"][ks]]), torch.cat([b[br]["q"][hs], b[br]["k"][ks], b[br]["v"][ks]]))
        attn = MetaFlashAttention(hidden, H, Hkv, fused("vanilla_attn"), table, window, num_meta_tokens=M, shard=s)
        lin = HLinearAttention(hidden, H, Hkv, fused("linear_attn"), layer_idx, num_meta_tokens=M, shard=s)
        blk = IntraHybridAttention(attn, lin, HymetaLinear.from_weight(w_out[:, hs].contiguous()), n1[hs], n2[hs])
        meta_kv = torch.zeros(2, M, s.num_kv_heads, D, dtype=torch.bfloat16, device=dev)
        meta_lin = torch.zeros(s.num_heads, D, D, dtype=torch.bfloat16, device=dev)
        attn.prepare_meta_state(meta_x, meta_kv)
        lin.prepare_meta_state(meta_x, meta_lin, lb)
        kv = torch.zeros(2, (T + bs - 1) // bs, bs, s.num_kv_heads, D, dtype=torch.bfloat16, device=dev)
        state = torch.zeros(1, s.num_heads, D, D, dtype=torch.bfloat16, device=dev)
        sidx = torch.zeros(1, dtype=torch.int32, device=dev)
        o_a = attn(x, pos, kv, meta_kv, md)
        o_l = lin(x, state, meta_lin, sidx, md, lb)
        return blk, o_a, o_l

    blk, o_a, o_l = build(1, 0)
    y_full = blk.out_proj(blk.combine(o_a, o_l)).float()

    parts = [build(2, r) for r in range(2)]
    stats = sum(hip_ops.rms_sumsq2(o_a, o_l) for _, o_a, o_l in parts) / 2        # the variance all-reduce
    y_tp = sum(blk.out_proj(hip_ops.hybrid_combine(o_a, o_l, blk.norm1_weight, blk.norm2_weight, blk.norm_eps,
                                                   stats=stats)).float() for blk, o_a, o_l in parts)
    torch.cuda.synchronize()
    # per-shard attention outputs are bit-identical to the unsharded heads
    assert torch.equal(torch.cat([p_[1] for p_ in parts], dim=1), o_a)
    assert torch.equal(torch.cat([p_[2] for p_ in parts], dim=1), o_l)
    assert_bf16_close(y_tp.cpu(), y_full.cpu(), "two shards vs unsharded", rel=1e-2)


def test_7b_decoder_block_gla_with_dense_mlp():
    """Pre-norm, GLA attention, norm with the residual add, gated MLP, residual: one block of the 7B
    model on a 512-token prompt."""
    from oracle.layers import decoder_block_ref, gla_layer_7b_ref
    from oracle.moe import mlp_ref
    from vllm_hymeta.linear import HymetaLinear
    from vllm_hymeta.model_for_7B.gla_attention import GatedLinearAttention
    from vllm_hymeta.models.block import GLAswaBlock, HymetaRMSNorm
    from vllm_hymeta.models.moe import HymetaMLP

    dev = torch.device("cuda")
    g_ = gen(7)
    hidden, H, Hkv, inter, T = 512, 4, 2, 1024, 512
    p = _gla_params(g_, hidden, H, Hkv)
    w_gu, w_dn = _lin(g_, 2 * inter, hidden, False)["weight"], _lin(g_, hidden, inter, False)["weight"]
    n1 = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16)
    n2 = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16)
    x = randn_bf16(g_, T, hidden)
    ref = decoder_block_ref(x, lambda h: gla_layer_7b_ref(h, p, H, Hkv)[0], lambda h: mlp_ref(h, w_gu, w_dn), n1, n2)

    qkv_w = torch.cat([p["q_proj.weight"], p["k_proj.weight"], p["v_proj.weight"]]).to(dev)
    qkv_b = torch.cat([p["q_proj.bias"], p["k_proj.bias"], p["v_proj.bias"]]).to(dev)
    attn = GatedLinearAttention(
        hidden, H, Hkv, HymetaLinear.from_weight(qkv_w, qkv_b), HymetaLinear.from_weight(p["gk_proj.0.weight"].to(dev)),
        HymetaLinear.from_weight(p["gk_proj.1.weight"].to(dev), p["gk_proj.1.bias"].to(dev)),
        HymetaLinear.from_weight(p["o_proj.weight"].to(dev)), p["g_norm.weight"].to(dev))
    mlp = HymetaMLP(HymetaLinear.from_weight(w_gu.to(dev)), HymetaLinear.from_weight(w_dn.to(dev)))
    block = GLAswaBlock(HymetaRMSNorm(hidden, 1e-6, n1.to(dev)), attn, HymetaRMSNorm(hidden, 1e-6, n2.to(dev)), mlp)
    cache = torch.zeros(2, H, D, D, dtype=torch.bfloat16, device=dev)
    xd = x.to(dev)
    out, second = block(xd, gla_cache=cache, state_indices_tensor=torch.tensor([1], dtype=torch.int32, device=dev),
                        attn_metadata=_md(dev, prefill_lens=[T], context_lens=[0]))
    torch.cuda.synchronize()
    assert second is None
    assert torch.equal(xd.cpu(), x), "the block must not modify its input"
    assert_bf16_close(out.cpu(), ref, "7B decoder block", rel=1e-2)

    # the residual stream carried beside the block's output (what the model classes do): same values,
    # and a second block fed with the pair equals a second block fed with the sum
    md = _md(dev, prefill_lens=[T], context_lens=[0])
    slot = torch.tensor([1], dtype=torch.int32, device=dev)
    delta, stream = block(xd, carry_residual=True, gla_cache=cache.zero_(), state_indices_tensor=slot, attn_metadata=md)
    assert torch.equal((stream + delta), out) and torch.equal(xd.cpu(), x)
    out2, _ = block(out, gla_cache=cache.zero_(), state_indices_tensor=slot, attn_metadata=md)
    delta2, stream2 = block(delta, carry_residual=True, residual=stream, gla_cache=cache.zero_(), state_indices_tensor=slot,
                            attn_metadata=md)
    torch.cuda.synchronize()
    assert torch.equal(stream2 + delta2, out2)
