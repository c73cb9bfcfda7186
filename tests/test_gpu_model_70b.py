"""The 70B-family model class end to end on a tiny configuration (4 layers: dense and expert
feed-forward, windowed and full attention, clamp and lower-bound gates, 32 meta tokens):
checkpoint-name loading, the meta tokens' one-off pass, a batch of two prompts, then a decode step
of both, against the CPU restatement that carries the meta rows in front of each whole sequence."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16
from test_gpu_layers import _md

pytestmark = pytest.mark.gpu
D = 128


def _checkpoint(g_, cfg):
    from oracle.layers import xavier_linear_params_ref as lin
    hidden, H, Hkv, inter = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    norm = lambda n: (1 + 0.1 * torch.randn(n, generator=g_)).to(torch.bfloat16)
    w = {"model.embeddings.weight": randn_bf16(g_, cfg.vocab_size, hidden),
         "mode.norm.weight": norm(hidden),                      # the published checkpoint's spelling
         "lm_head.weight": randn_bf16(g_, cfg.vocab_size, hidden, scale=0.05),
         "model.meta_tokens": randn_bf16(g_, cfg.num_meta_tokens, hidden),
         "model.lower_bounds": torch.randn(cfg.num_hidden_layers, hidden, generator=g_)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for br in ("vanilla_attn", "linear_attn"):
            for n, o in (("q_proj", H * D), ("k_proj", Hkv * D), ("v_proj", Hkv * D)):
                w[p + f"attn.{br}.{n}.weight"] = lin(g_, o, hidden, False)["weight"]
                w[p + f"attn.{br}.{n}.bias"] = randn_bf16(g_, o, scale=0.1)
        w[p + "attn.vanilla_attn.rotary_emb.inv_freq"] = torch.zeros(D // 2)       # routed, never used
        w[p + "attn.out_proj.weight"] = lin(g_, hidden, H * D, False)["weight"]
        w[p + "attn.norm1.weight"], w[p + "attn.norm2.weight"] = norm(hidden).float(), norm(hidden).float()
        w[p + "attn_norm.weight"], w[p + "mlp_norm.weight"] = norm(hidden), norm(hidden)

        def mlp(name, size):
            w[p + name + ".gate_proj.weight"] = lin(g_, size, hidden, False)["weight"]
            w[p + name + ".up_proj.weight"] = lin(g_, size, hidden, False)["weight"]
            w[p + name + ".down_proj.weight"] = lin(g_, hidden, size, False)["weight"]

        if cfg.num_layer_experts[i] == 1:
            mlp("mlp", inter)
        else:
            # a wide router keeps the experts' probabilities apart, so that the two sides pick the same expert
            w[p + "block_sparse_moe.gate.weight"] = randn_bf16(g_, cfg.num_layer_experts[i], hidden, scale=0.5)
            for e in range(cfg.num_layer_experts[i]):
                mlp(f"block_sparse_moe.experts.{e}", inter)
            mlp("shared_mlp", cfg.shared_intermediate_size)
    return w


@pytest.mark.parametrize("engine_style", [False, True])
def test_hymeta_for_causal_lm_prefill_then_decode(engine_style):
    """``engine_style``: the model is built from a (duck-typed) ``VllmConfig``, its paged KV caches are bound to the
    registered attention layers by name and the step's metadata comes from the forward context -- the way vLLM
    constructs and drives the class -- instead of being passed to ``forward``."""
    from oracle.model import hymeta_model_ref
    from vllm_hymeta.models import attention
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

    dev = torch.device("cuda")
    g_ = gen(12)
    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    assert cfg.num_layer_experts == [1, 1, 1, 4] and cfg.interleaved_sliding_window == [48, 48, 48, 0]
    ckpt = _checkpoint(g_, cfg)
    if engine_style:
        from test_vllm_boundary import vllm_config_like
        attention.release_layers("model.")
        model = HymetaForCausalLM(vllm_config=vllm_config_like(cfg, max_num_seqs=4, device=dev), prefix="")
    else:
        model = HymetaForCausalLM(cfg, max_num_seqs=4, device=dev)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.long, device=dev), attn_metadata=object())
    loaded = model.load_weights(ckpt.items())
    assert "model.norm.weight" in loaded and "model.layers.3.block_sparse_moe.experts.w13_weight" in loaded
    assert "model.layers.0.attn.linear_attn.qkv_proj.bias" in loaded
    assert model.model.has_meta_cache

    lens = [75, 40]
    ids = [torch.randint(0, cfg.vocab_size, (L + 1,), generator=g_) for L in lens]
    refs = [hymeta_model_ref(x, ckpt, cfg) for x in ids]

    bs = 16
    per_seq = (max(lens) + 1 + bs - 1) // bs
    kv_caches = [torch.zeros(2, 2 * per_seq + 1, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev)
                 for _ in range(cfg.num_hidden_layers)]
    bt = torch.randperm(2 * per_seq + 1, generator=g_)[: 2 * per_seq].to(torch.int32).view(2, per_seq).to(dev)
    slot = lambda s, t: int(bt[s, t // bs]) * bs + t % bs
    run = dict(request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])

    pre_ids = torch.cat([x[:L] for x, L in zip(ids, lens)]).to(dev)
    pre_pos = torch.cat([torch.arange(L) for L in lens]).to(dev)
    pre_slots = torch.tensor([slot(s, t) for s, L in enumerate(lens) for t in range(L)], device=dev)
    md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=pre_slots)
    if engine_style:
        layers = attention.registered_layers()
        for i, kv in enumerate(kv_caches):
            attention.bind_kv_cache(layers[f"model.layers.{i}.attn.vanilla_attn.attn"], kv)

        def step(ids_, pos_, md_):
            with attention.forward_context(md_):
                return model(ids_, pos_, **run)
    else:
        def step(ids_, pos_, md_):
            return model(ids_, pos_, attn_metadata=md_, kv_caches=kv_caches, **run)
    h_pre = step(pre_ids, pre_pos, md)

    dec_ids = torch.stack([x[L] for x, L in zip(ids, lens)]).to(dev)
    dec_pos = torch.tensor(lens, device=dev)
    dec_slots = torch.tensor([slot(s, L) for s, L in enumerate(lens)], device=dev)
    md = _md(dev, decode_seq_lens=[L + 1 for L in lens], block_tables=bt, slot_mapping=dec_slots)
    h_dec = step(dec_ids, dec_pos, md)
    logits = model.compute_logits(h_dec)
    torch.cuda.synchronize()

    ref_pre = torch.cat([r[0][:L] for r, L in zip(refs, lens)])
    ref_dec = torch.stack([r[0][L] for r, L in zip(refs, lens)])
    ref_logits = torch.stack([r[1][L] for r, L in zip(refs, lens)])
    assert h_pre.shape == (sum(lens), cfg.hidden_size), "a serving step carries no meta rows"
    assert_bf16_close(h_pre.cpu(), ref_pre, "70B model, prefill hidden states", rel=2e-2, frac=2.0 ** -4)
    assert_bf16_close(h_dec.cpu(), ref_dec, "70B model, decode hidden states", rel=2e-2, frac=2.0 ** -4)
    assert_bf16_close(logits.cpu(), ref_logits, "70B model, decode logits", rel=2e-2, frac=2.0 ** -4)
    if engine_style:
        attention.release_layers("model.")


def _w8a8_recipe(ckpt):
    """The checkpoint as ``hf_model/quant_w8a8.py:121-132`` would store it: every Linear but the lm_head as int8 weights
    with one fp32 scale per output channel (``weight_scale [out, 1]``) -- the routed experts and the router included."""
    out = {}
    for name, t in ckpt.items():
        is_linear = name.endswith("_proj.weight") or name.endswith("block_sparse_moe.gate.weight")
        if not is_linear:
            out[name] = t
            continue
        w = t.float()
        scale = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-8) / 127.0
        out[name] = torch.clamp(torch.round(w / scale), -127, 127).to(torch.int8)
        out[name + "_scale"] = scale
    return out


def test_w8a8_checkpoint_with_quantised_experts_and_router():
    """A W8A8 checkpoint of the 70B family holds int8 experts and an int8 router.  The model keeps them int8 (grouped int8
    GEMMs, the router as a W8A8 linear) and agrees with the same checkpoint whose EXPERT weights were expanded to bf16
    -- same routing (the router is identical on both sides), so the difference is the experts' activation quantisation
    and the bf16 rounding of the expanded weights."""
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM
    from vllm_hymeta.models.moe import HymetaMoE

    dev = torch.device("cuda")
    g_ = gen(21)
    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    quantised = _w8a8_recipe(_checkpoint(g_, cfg))
    expanded = {}
    for name, t in quantised.items():
        if ".experts." in name:
            if name.endswith("weight_scale"):
                continue
            t = (t.float() * quantised[name + "_scale"]).to(torch.bfloat16)
        expanded[name] = t

    lens = [60, 33]
    ids = [torch.randint(0, cfg.vocab_size, (L,), generator=g_) for L in lens]
    bs = 16
    per_seq = (max(lens) + bs - 1) // bs
    bt = torch.arange(2 * per_seq, dtype=torch.int32).view(2, per_seq).to(dev)
    slots = torch.tensor([int(bt[s, t // bs]) * bs + t % bs for s, L in enumerate(lens) for t in range(L)], device=dev)
    outs = []
    for ckpt in (quantised, expanded):
        model = HymetaForCausalLM(cfg, max_num_seqs=4, device=dev)
        loaded = model.load_weights(ckpt.items())
        moe = [m for m in model.modules() if isinstance(m, HymetaMoE)]
        assert len(moe) == 1 and moe[0].gate_weight.dtype == torch.int8 and moe[0].w8a8_gate is not None
        if ckpt is quantised:
            assert "model.layers.3.block_sparse_moe.experts.w13_weight_scale" in loaded
            assert moe[0].w13.dtype == torch.int8 and moe[0].w13.shape == (4, 2 * cfg.intermediate_size, cfg.hidden_size)
            assert moe[0].w13_scale.shape == (4, 2 * cfg.intermediate_size) and moe[0].w2_scale.shape == (4, cfg.hidden_size)
        else:
            assert moe[0].w13.dtype == torch.bfloat16 and moe[0].w13_scale is None
        kv = [torch.zeros(2, 2 * per_seq, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev)
              for _ in range(cfg.num_hidden_layers)]
        md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=slots)
        h = model(torch.cat(ids).to(dev), torch.cat([torch.arange(L) for L in lens]).to(dev), attn_metadata=md, kv_caches=kv,
                  request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])
        torch.cuda.synchronize()
        outs.append(h.float().cpu())
    assert torch.isfinite(outs[0]).all()
    assert_bf16_close(outs[0], outs[1], "70B W8A8 checkpoint: int8 experts against the same experts expanded to bf16", rel=3e-2,
                      frac=0.15)


def _gptq_recipe(ckpt, g_, bits=4, group=128):
    """The checkpoint as gptqmodel stores it with the module list of ``hf_model/hymeta_gptq.py:22-35``: attention, MLP,
    shared-expert and routed-expert projections as ``qweight / qzeros / scales / g_idx`` (round-to-nearest here; the
    recipe's error compensation does not change the format); the router stays floating point.  Also returns the same
    checkpoint with every quantised projection expanded to the bf16 weight it stands for."""
    from oracle.quant import gptq_dequant_ref, gptq_pack_ref
    quantised, expanded = {}, {}
    for name, t in ckpt.items():
        if not name.endswith("_proj.weight"):
            quantised[name] = expanded[name] = t
            continue
        w = t.float().t().contiguous()                      # [K, N]
        K, N = w.shape
        wg = w.view(K // group, group, N)
        lo, hi = wg.amin(dim=1), wg.amax(dim=1)
        scale = ((hi - lo).clamp_min(1e-6) / (2 ** bits - 1)).to(torch.float16)
        zero = torch.clamp(torch.round(-lo / scale.float()), 1, 2 ** bits - 1).long()
        q = torch.clamp(torch.round(wg / scale.float()[:, None, :]) + zero[:, None, :], 0, 2 ** bits - 1).long().view(K, N)
        qw, qz = gptq_pack_ref(q, zero, bits)
        base = name[: -len("weight")]
        quantised[base + "qweight"], quantised[base + "qzeros"], quantised[base + "scales"] = qw, qz, scale
        quantised[base + "g_idx"] = (torch.arange(K, dtype=torch.int32) // group)
        expanded[name] = gptq_dequant_ref(qw, qz, scale, bits, group).t().to(torch.bfloat16).contiguous()
    return quantised, expanded


def test_gptq_checkpoint_with_quantised_experts():
    """A GPTQ checkpoint of the 70B family holds GPTQ experts.  The model keeps them packed (fused de-quantising grouped
    GEMM on decode-sized steps, expansion per call on prefill-sized ones) and agrees with the bf16 model built from the
    weights the packed tensors stand for."""
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM
    from vllm_hymeta.models.moe import HymetaMoE
    from vllm_hymeta import hip_ops

    dev = torch.device("cuda")
    g_ = gen(31)
    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    quantised, expanded = _gptq_recipe(_checkpoint(g_, cfg), g_)
    bs = 16
    outs = {}
    for lens in ([60, 33], [400, 420]):          # 93 pairs on 4 experts: the fused kernel; 820: the expansion
        ids = [torch.randint(0, cfg.vocab_size, (L,), generator=g_) for L in lens]
        per_seq = (max(lens) + bs - 1) // bs
        bt = torch.arange(2 * per_seq, dtype=torch.int32).view(2, per_seq).to(dev)
        slots = torch.tensor([int(bt[s, t // bs]) * bs + t % bs for s, L in enumerate(lens) for t in range(L)], device=dev)
        for kind, ckpt in (("gptq", quantised), ("bf16", expanded)):
            model = HymetaForCausalLM(cfg, max_num_seqs=4, device=dev,
                                      quant={"bits": 4, "group_size": 128, "zero_offset": 1} if kind == "gptq" else None)
            loaded = model.load_weights(ckpt.items())
            moe = [m for m in model.modules() if isinstance(m, HymetaMoE)]
            assert len(moe) == 1
            if kind == "gptq":
                assert "model.layers.3.block_sparse_moe.experts.w13_qweight" in loaded
                assert isinstance(moe[0].w13, hip_ops.GptqExperts) and tuple(moe[0].w13.shape) == (4, 2 * cfg.intermediate_size, cfg.hidden_size)
                assert tuple(moe[0].w2.shape) == (4, cfg.hidden_size, cfg.intermediate_size)
            kv = [torch.zeros(2, 2 * per_seq, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev)
                  for _ in range(cfg.num_hidden_layers)]
            md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=slots)
            h = model(torch.cat(ids).to(dev), torch.cat([torch.arange(L) for L in lens]).to(dev), attn_metadata=md, kv_caches=kv,
                      request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])
            torch.cuda.synchronize()
            outs[kind] = h.float().cpu()
        assert torch.isfinite(outs["gptq"]).all()
        assert_bf16_close(outs["gptq"], outs["bf16"], f"70B GPTQ checkpoint, {sum(lens)} tokens: packed against expanded weights",
                          rel=2e-2, frac=0.1)


@pytest.mark.parametrize("stages", [2, 3])
def test_pipeline_stages_equal_the_whole_model(stages):
    """The reference's 8-GPU recipe is TP4 x PP2: the model cut into pipeline stages (here all in one process, the
    hand-off carried by the test).  The meta rows pass from stage to stage once, at load time; every serving step hands
    ``IntermediateTensors`` on; prefill and decode of two sequences give the whole model's hidden states and logits bit
    for bit (the hand-off sum is the one the next block's fused add + norm forms)."""
    from vllm_hymeta import _compat
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

    dev = torch.device("cuda")
    g_ = gen(41)
    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    ckpt = _checkpoint(g_, cfg)
    whole = HymetaForCausalLM(cfg, max_num_seqs=4, device=dev)
    whole.load_weights(ckpt.items())
    pipe = [HymetaForCausalLM(cfg, max_num_seqs=4, device=dev, pp_rank=r, pp_size=stages) for r in range(stages)]
    rows = None
    for st in pipe:
        st.load_weights(ckpt.items())
        if st.model.pp_rank > 0:
            assert not st.model.has_meta_cache
            with pytest.raises(RuntimeError, match="meta caches are empty"):
                st(None, torch.zeros(1, dtype=torch.long, device=dev), attn_metadata=object(),
                   intermediate_tensors=st.make_empty_intermediate_tensors(1, torch.bfloat16, dev))
            st.model.prepare_meta_caches(rows)
        rows = st.model.meta_hidden_out
        assert st.model.has_meta_cache and rows.shape == (cfg.num_meta_tokens, cfg.hidden_size)
    with pytest.raises(RuntimeError, match="stage before it"):
        HymetaForCausalLM(cfg, max_num_seqs=4, device=dev, pp_rank=1, pp_size=2).model.prepare_meta_caches()

    lens = [75, 40]
    ids = [torch.randint(0, cfg.vocab_size, (L + 1,), generator=g_) for L in lens]
    bs = 16
    per_seq = (max(lens) + 1 + bs - 1) // bs
    bt = torch.randperm(2 * per_seq + 1, generator=g_)[: 2 * per_seq].to(torch.int32).view(2, per_seq).to(dev)
    slot = lambda s_, t: int(bt[s_, t // bs]) * bs + t % bs
    run = dict(request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])

    def caches(model):
        return [torch.zeros(2, 2 * per_seq + 1, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev)
                for _ in model.model.attn_layers]

    kv_whole, kv_pipe = caches(whole), [caches(st) for st in pipe]

    def through_pipe(ids_, pos_, md_):
        x = None
        for st, kv in zip(pipe, kv_pipe):
            x = st(ids_ if st.model.is_first_stage else None, pos_, intermediate_tensors=x, attn_metadata=md_, kv_caches=kv, **run)
            if not st.model.is_last_stage:
                assert isinstance(x, _compat.IntermediateTensors) and not bool(x["residual"].any())
        return x

    pre_ids = torch.cat([x[:L] for x, L in zip(ids, lens)]).to(dev)
    pre_pos = torch.cat([torch.arange(L) for L in lens]).to(dev)
    pre_slots = torch.tensor([slot(s_, t) for s_, L in enumerate(lens) for t in range(L)], device=dev)
    md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=pre_slots)
    h_whole = whole(pre_ids, pre_pos, attn_metadata=md, kv_caches=kv_whole, **run)
    h_pipe = through_pipe(pre_ids, pre_pos, md)
    torch.cuda.synchronize()
    assert torch.equal(h_whole, h_pipe)

    dec_ids = torch.stack([x[L] for x, L in zip(ids, lens)]).to(dev)
    dec_pos = torch.tensor(lens, device=dev)
    dec_slots = torch.tensor([slot(s_, L) for s_, L in enumerate(lens)], device=dev)
    md = _md(dev, decode_seq_lens=[L + 1 for L in lens], block_tables=bt, slot_mapping=dec_slots)
    d_whole = whole(dec_ids, dec_pos, attn_metadata=md, kv_caches=kv_whole, **run)
    d_pipe = through_pipe(dec_ids, dec_pos, md)
    torch.cuda.synchronize()
    assert torch.equal(d_whole, d_pipe)
    assert torch.equal(whole.compute_logits(d_whole), pipe[-1].compute_logits(d_pipe))
