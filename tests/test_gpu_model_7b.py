"""The 7B model class end to end on a tiny configuration: checkpoint-name loading, a batch of two
prompts, then a decode step of both, against the CPU restatement run on each whole sequence."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16
from test_gpu_layers import _md

pytestmark = pytest.mark.gpu
D = 128


def _checkpoint(g_, cfg):
    from oracle.layers import xavier_linear_params_ref as lin
    hidden, H, Hkv, inter = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    w = {"model.embeddings.weight": randn_bf16(g_, cfg.vocab_size, hidden),
         "model.norm.weight": (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16),
         "lm_head.weight": randn_bf16(g_, cfg.vocab_size, hidden, scale=0.05)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for n, o in (("q_proj", H * D), ("k_proj", Hkv * D), ("v_proj", Hkv * D)):
            w[p + f"attn.{n}.weight"] = lin(g_, o, hidden, False)["weight"]
            w[p + f"attn.{n}.bias"] = randn_bf16(g_, o, scale=0.1)
        w[p + "attn.o_proj.weight"] = lin(g_, hidden, H * D, False)["weight"]
        if i not in cfg.attn_layers:
            w[p + "attn.gk_proj.0.weight"] = lin(g_, 16, hidden, False)["weight"]
            w[p + "attn.gk_proj.1.weight"] = lin(g_, Hkv * D, 16, False)["weight"]
            w[p + "attn.gk_proj.1.bias"] = randn_bf16(g_, Hkv * D, scale=0.1)
            w[p + "attn.g_norm.weight"] = (1 + 0.1 * torch.randn(D, generator=g_)).to(torch.bfloat16)
        for n in ("attn_norm", "mlp_norm"):
            w[p + n + ".weight"] = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16)
        w[p + "mlp.gate_proj.weight"] = lin(g_, inter, hidden, False)["weight"]
        w[p + "mlp.up_proj.weight"] = lin(g_, inter, hidden, False)["weight"]
        w[p + "mlp.down_proj.weight"] = lin(g_, hidden, inter, False)["weight"]
    # a tensor of a layer the configuration does not have must be dropped, not loaded
    w[f"model.layers.{cfg.num_hidden_layers}.mlp.down_proj.weight"] = torch.zeros(hidden, inter, dtype=torch.bfloat16)
    return w


@pytest.mark.parametrize("state_per_kv_head,engine_style", [(True, False), (False, False), (True, True)])
def test_glaswa_for_causal_lm_prefill_then_decode(state_per_kv_head, engine_style):
    """``engine_style``: built from a (duck-typed) ``VllmConfig``, paged KV caches bound to the registered
    sliding-window attention layers by name, step metadata from the forward context (how vLLM drives the class)."""
    from oracle.model import glaswa_model_ref
    from vllm_hymeta.models import attention
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    dev = torch.device("cuda")
    g_ = gen(11)
    cfg = GLAswaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=1024, sliding_window=48, max_position_embeddings=1024)
    ckpt = _checkpoint(g_, cfg)
    if engine_style:
        from test_vllm_boundary import vllm_config_like
        attention.release_layers("model.")
        model = GLAswaForCausalLM(vllm_config=vllm_config_like(cfg, max_num_seqs=4, device=dev), prefix="")
    else:
        model = GLAswaForCausalLM(cfg, max_num_seqs=4, device=dev, state_per_kv_head=state_per_kv_head)
    loaded = model.load_weights(ckpt.items())
    assert "model.layers.0.attn.qkv_proj.weight" in loaded and "model.layers.1.mlp.gate_up_proj.weight" in loaded
    assert not any(".layers.4." in n for n in loaded)
    # one GLA state per key/value head (2), or per query head (4) as the reference caches them
    assert model.model.cache_shape == (2, 4, 2 if state_per_kv_head else 4, D, D)

    lens = [70, 41]
    ids = [torch.randint(0, cfg.vocab_size, (L + 1,), generator=g_) for L in lens]     # prompt + the decode step's token
    refs = [glaswa_model_ref(x, ckpt, cfg) for x in ids]

    bs = 16
    per_seq = (max(lens) + 1 + bs - 1) // bs
    n_swa = len(cfg.attn_layers)
    kv_caches = [torch.zeros(2, 2 * per_seq + 1, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev)
                 for _ in range(n_swa)]
    bt = torch.randperm(2 * per_seq + 1, generator=g_)[: 2 * per_seq].to(torch.int32).view(2, per_seq).to(dev)
    slot = lambda s, t: int(bt[s, t // bs]) * bs + t % bs
    run = dict(request_ids_to_seq_ids={"a": [0], "b": [1]}, finished_requests_ids=[])

    pre_ids = torch.cat([x[:L] for x, L in zip(ids, lens)]).to(dev)
    pre_pos = torch.cat([torch.arange(L) for L in lens]).to(dev)
    pre_slots = torch.tensor([slot(s, t) for s, L in enumerate(lens) for t in range(L)], device=dev)
    md = _md(dev, prefill_lens=lens, context_lens=[0, 0], slot_mapping=pre_slots)
    if engine_style:
        layers = attention.registered_layers()
        for i, kv in zip(cfg.attn_layers, kv_caches):
            attention.bind_kv_cache(layers[f"model.layers.{i}.attn.attn"], kv)

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
    assert_bf16_close(h_pre.cpu(), ref_pre, "7B model, prefill hidden states", rel=2e-2, frac=2.0 ** -4)
    assert_bf16_close(h_dec.cpu(), ref_dec, "7B model, decode hidden states", rel=2e-2, frac=2.0 ** -4)
    assert logits.dtype == torch.float32
    assert_bf16_close(logits.cpu(), ref_logits, "7B model, decode logits", rel=2e-2, frac=2.0 ** -4)

    # the same request ids keep their slots; a finished request frees its slot
    slots_before = [model.model.gla_cache.slots.slot_of("a", 0), model.model.gla_cache.slots.slot_of("b", 1)]
    assert len(set(slots_before)) == 2
    assert model(None, dec_pos, attn_metadata=None) is None          # profiling run: no metadata, no work
    with pytest.raises(ValueError):
        model.model(None, dec_pos, attn_metadata=md)
    if engine_style:
        attention.release_layers("model.")


def test_glaswa_for_causal_lm_gptq_checkpoint():
    """The same model from a 4-bit GPTQ checkpoint (qweight / qzeros / scales / g_idx per projection, the
    low-rank gate projections left in bf16 as quant_7B.py leaves them): loaded by name, run on the
    fused de-quantising GEMMs, against the restatement run on the de-quantised weights."""
    import re

    from oracle.model import glaswa_model_ref
    from oracle.quant import gptq_dequant_ref, gptq_pack_ref
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    dev = torch.device("cuda")
    g_ = gen(13)
    group = 128
    cfg = GLAswaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=1024, sliding_window=48, max_position_embeddings=1024)
    dense = _checkpoint(g_, cfg)
    quantised = {}
    for name, w in dense.items():
        m = re.match(r"(.*\.(?:q_proj|k_proj|v_proj|o_proj|gate_proj|up_proj|down_proj))\.weight$", name)
        if m is None or f"layers.{cfg.num_hidden_layers}." in name:
            quantised[name] = w
            continue
        N, K = w.shape
        wi = torch.randint(0, 16, (K, N), generator=g_)
        z = torch.randint(1, 16, (K // group, N), generator=g_)
        s = (torch.rand(K // group, N, generator=g_) * 0.002 + 0.0005).to(torch.float16)
        qw, qz = gptq_pack_ref(wi, z, 4)
        quantised[m.group(1) + ".qweight"], quantised[m.group(1) + ".qzeros"], quantised[m.group(1) + ".scales"] = qw, qz, s
        quantised[m.group(1) + ".g_idx"] = torch.arange(K, dtype=torch.int32) // group
        dense[name] = gptq_dequant_ref(qw, qz, s, 4, group).t().contiguous().to(torch.bfloat16)
    model = GLAswaForCausalLM(cfg, max_num_seqs=2, quant={"bits": 4, "group_size": group}, device=dev)
    loaded = model.load_weights(quantised.items())
    assert "model.layers.0.attn.qkv_proj.qweight" in loaded and "model.layers.1.mlp.gate_up_proj.scales" in loaded
    assert model.model.layers[0].mlp.down_proj.kind == "gptq" and model.model.layers[0].attn.gk_proj0.kind == "bf16"

    L = 50
    ids = torch.randint(0, cfg.vocab_size, (L + 1,), generator=g_)
    ref_h, ref_logits = glaswa_model_ref(ids, dense, cfg)
    bs = 16
    nblk = (L + 1 + bs - 1) // bs
    kv_caches = [torch.zeros(2, nblk, bs, cfg.num_key_value_heads, D, dtype=torch.bfloat16, device=dev)]
    bt = torch.arange(nblk, dtype=torch.int32, device=dev).view(1, -1)
    run = dict(request_ids_to_seq_ids={"r": [0]}, finished_requests_ids=[])
    pos = torch.arange(L + 1, device=dev)
    h_pre = model(ids[:L].to(dev), pos[:L], attn_metadata=_md(dev, prefill_lens=[L], context_lens=[0], slot_mapping=pos[:L]),
                  kv_caches=kv_caches, **run)
    h_dec = model(ids[L:].to(dev), pos[L:], attn_metadata=_md(dev, decode_seq_lens=[L + 1], block_tables=bt, slot_mapping=pos[L:]),
                  kv_caches=kv_caches, **run)
    logits = model.compute_logits(h_dec)
    torch.cuda.synchronize()
    assert_bf16_close(torch.cat([h_pre, h_dec]).cpu(), ref_h, "7B model from a GPTQ checkpoint, hidden states", rel=2e-2, frac=2.0 ** -4)
    assert_bf16_close(logits.cpu(), ref_logits[L:], "7B model from a GPTQ checkpoint, logits", rel=2e-2, frac=2.0 ** -4)


@pytest.mark.parametrize("stages", [2, 3])
def test_pipeline_stages_equal_the_whole_model(stages):
    """The model cut into pipeline stages (all in one process, the hand-off carried by the test): every step hands
    ``IntermediateTensors`` (``hidden_states`` = the residual stream, ``residual`` = zeros, as the reference's stages
    do) to the next stage; prefill and decode of two sequences give the whole model's hidden states and logits bit for
    bit.  With three stages of four layers the last stage holds one sliding-window layer and no linear-attention state."""
    from vllm_hymeta import _compat
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    dev = torch.device("cuda")
    g_ = gen(43)
    cfg = GLAswaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=1024, sliding_window=48, max_position_embeddings=1024)
    ckpt = _checkpoint(g_, cfg)
    whole = GLAswaForCausalLM(cfg, max_num_seqs=4, device=dev)
    whole.load_weights(ckpt.items())
    pipe = [GLAswaForCausalLM(cfg, max_num_seqs=4, device=dev, pp_rank=r, pp_size=stages) for r in range(stages)]
    for st in pipe:
        st.load_weights(ckpt.items())
    assert sum(len(st.model.layers) for st in pipe) == 4 and sum(st.model.cache_shape[0] for st in pipe) == 2
    assert [st.lm_head is not None for st in pipe] == [False] * (stages - 1) + [True]

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
    with pytest.raises(ValueError, match="stage before it"):
        pipe[1](None, dec_pos, attn_metadata=md, kv_caches=kv_pipe[1], **run)
