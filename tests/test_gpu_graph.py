"""A decode step captured into a HIP graph and replayed: the reference relies on CUDA-graph capture
of the decode path (``use_cuda_graph``, ``vllm_hymeta/attention/meta_attention.py:163-167,557-562``;
capture buffers of the state cache, ``vllm_hymeta/models/hymeta_cache.py:56-62``).  Every entry
point on that path must therefore be stream-ordered, allocation-free and free of host reads; the
replays must reproduce what the same calls produce eagerly, on inputs changed between replays."""
import pytest
import torch

from _util import gen, randn_bf16

pytestmark = pytest.mark.gpu
D = 128


def _gptq_tensors(g_, K, N, bits, group):
    pack = 32 // bits
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // pack, N), generator=g_, dtype=torch.int64).to(torch.int32)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // group, N // pack), generator=g_, dtype=torch.int64).to(torch.int32)
    scales = (torch.rand(K // group, N, generator=g_) * 0.02 + 0.005).to(torch.float16)
    return qweight, qzeros, scales


def test_decode_step_replays_from_a_graph():
    from vllm_hymeta import _C, hip_ops

    dev = torch.device("cuda")
    g_ = gen(11)
    B, H, Hkv, hidden, bs, W, M = 8, 7, 1, 896, 16, 64, 32
    E, inter = 4, 256
    nblk_seq = 12
    nblk = B * nblk_seq + 1
    qkv_n = (H + 2 * Hkv) * D

    # weights and caches (static across replays)
    norm_w = (1 + 0.1 * torch.randn(hidden, generator=g_)).to(torch.bfloat16).to(dev)
    qw, qz, sc = (t.to(dev) for t in _gptq_tensors(g_, hidden, qkv_n, 4, 64))
    lower = torch.rand(H * D, generator=g_).mul(0.5).to(dev)
    meta_k, meta_v = randn_bf16(g_, M, Hkv, D).to(dev), randn_bf16(g_, M, Hkv, D).to(dev)
    w13 = randn_bf16(g_, E, 2 * inter, hidden, scale=0.05).to(dev)
    w2 = randn_bf16(g_, E, hidden, inter, scale=0.05).to(dev)
    gate_w = randn_bf16(g_, E, hidden, scale=0.05).to(dev)
    kv0 = randn_bf16(g_, 2, nblk, bs, Hkv, D).to(dev)
    state0 = randn_bf16(g_, B + 1, H, D, D, scale=0.1).to(dev)
    bt = torch.randperm(nblk - 1, generator=g_)[: B * nblk_seq].view(B, nblk_seq).to(torch.int32).to(dev)

    # per-step inputs live in static buffers that the graph reads
    x_buf = torch.zeros(B, hidden, dtype=torch.bfloat16, device=dev)
    res_buf = torch.zeros(B, hidden, dtype=torch.bfloat16, device=dev)
    seqlens = torch.zeros(B, dtype=torch.int32, device=dev)
    slots = torch.zeros(B, dtype=torch.int64, device=dev)
    state_idx = torch.zeros(B, dtype=torch.int32, device=dev)

    def step(kv, state):
        normed = hip_ops.add_rmsnorm(x_buf, norm_w, 1e-6, residual=res_buf)
        qkv = hip_ops.gptq_gemm(normed, qw, qz, sc, 4, 64)
        q = qkv[:, : H * D].view(B, H, D)
        k = qkv[:, H * D:(H + Hkv) * D].view(B, Hkv, D)
        v = qkv[:, (H + Hkv) * D:].view(B, Hkv, D)
        o_lin = hip_ops.gla_decode(q, k, v, None, state, state_idx, gate_mode=_C.GATE_70B_LOWER_BOUND, lower_bound=lower)
        o_att = hip_ops.meta_swa_paged_decode(q, kv[0], kv[1], seqlens, bt, meta_k, meta_v, window_left=W - 1,
                                              max_seqlen_k=nblk_seq * bs, new_key=k, new_value=v, slot_mapping=slots)
        h = (o_lin.view(B, H * D)[:, :hidden] + o_att.view(B, H * D)[:, :hidden])
        logits = torch.nn.functional.linear(h, gate_w)
        return hip_ops.moe_forward(h, logits, w13, w2, top_k=1)

    def set_inputs(seed, lens, pad_last):
        gi = gen(seed)
        x_buf.copy_(randn_bf16(gi, B, hidden))
        res_buf.copy_(randn_bf16(gi, B, hidden))
        seqlens.copy_(torch.tensor(lens, dtype=torch.int32))
        pos = torch.tensor(lens) - 1
        sl = bt.cpu()[torch.arange(B), pos // bs].long() * bs + pos % bs
        idx = torch.arange(1, B + 1, dtype=torch.int32)
        if pad_last:        # a padded graph slot: no KV write, no state update
            sl[-1] = -1
            idx[-1] = -1
        slots.copy_(sl)
        state_idx.copy_(idx)

    cases = [(1, [5, 17, 64, 65, 100, 130, 191, 33], False), (2, [6, 18, 65, 66, 101, 131, 192, 1], True)]

    # eager results, each case from the same initial caches
    eager = []
    for seed, lens, pad in cases:
        kv, state = kv0.clone(), state0.clone()
        set_inputs(seed, lens, pad)
        out = step(kv, state)
        torch.cuda.synchronize()
        eager.append((out.clone(), kv.clone(), state.clone(), res_buf.clone()))

    # capture once (on a side stream, after the warm-up above sized every workspace), replay per case
    kv_g, state_g = kv0.clone(), state0.clone()
    set_inputs(*cases[0])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = step(kv_g, state_g)
    for (seed, lens, pad), (out_e, kv_e, state_e, res_e) in zip(cases, eager):
        kv_g.copy_(kv0)
        state_g.copy_(state0)
        set_inputs(seed, lens, pad)
        graph.replay()
        torch.cuda.synchronize()
        rows = B - 1 if pad else B      # the padded row's linear-attention output is left unwritten
        assert torch.equal(out_g[:rows], out_e[:rows]), "graph replay differs from the eager step"
        assert torch.equal(kv_g, kv_e) and torch.equal(state_g, state_e) and torch.equal(res_buf, res_e)


@pytest.mark.parametrize("quant", ["bf16", "w8a8"])
def test_whole_model_decode_step_replays_from_a_graph(quant):
    """The 7B model class's decode step (streaming linears with K slices summed by the norms, the gate folded into qkv and
    gla_decode, int8 rows written by the norms for W8A8) captured after an eager warm-up and replayed three times with
    eager work and new inputs between the replays: every replay equals the eager step on the same inputs and states,
    the fp32 logits and the greedy tokens included."""
    from test_gpu_layers import _md
    from test_gpu_model_7b import _checkpoint
    from vllm_hymeta import hip_ops
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    dev = torch.device("cuda")
    g_ = gen(21)
    cfg = GLAswaConfig(vocab_size=256, hidden_size=1024, num_hidden_layers=4, num_attention_heads=8, num_key_value_heads=2,
                       intermediate_size=2048, sliding_window=40, max_position_embeddings=512)
    ckpt = _checkpoint(g_, cfg)
    if quant == "w8a8":
        import re
        for name in list(ckpt):
            if re.search(r"\.(q_proj|k_proj|v_proj|o_proj|gate_proj|up_proj|down_proj)\.weight$", name):
                w = ckpt[name].float()
                s = w.abs().amax(1, keepdim=True).clamp_min(1e-8) / 127
                ckpt[name] = torch.clamp(torch.round(w / s), -127, 127).to(torch.int8)
                ckpt[name + "_scale"] = s.float()
    B, bs, ctx = 20, 16, 70          # K-sliced streaming linears (bf16 and W8A8): the deferred sums are in the graph
    model = GLAswaForCausalLM(cfg, max_num_seqs=B, device=dev)
    model.load_weights(ckpt.items())
    lin = model.model.layers[0].mlp.down_proj
    assert lin.kind == quant and isinstance(lin(randn_bf16(g_, B, 2048).cuda(), defer=True), hip_ops.PartialSums)
    nblk = (ctx + 8 + bs - 1) // bs
    kv = [torch.randn(2, B * nblk, bs, 2, D, device=dev).to(torch.bfloat16) for _ in cfg.attn_layers]
    bt = torch.randperm(B * nblk, generator=g_).to(torch.int32).view(B, nblk).to(dev)
    ids = torch.zeros(B, dtype=torch.int64, device=dev)
    pos = torch.zeros(B, dtype=torch.int64, device=dev)
    seq_lens = torch.zeros(B, dtype=torch.int32, device=dev)
    slots = torch.zeros(B, dtype=torch.int64, device=dev)
    md = _md(dev, decode_seq_lens=[ctx + 8] * B, block_tables=bt, slot_mapping=slots)
    md.seq_lens_tensor, md.seq_lens, md.use_cuda_graph = seq_lens, None, True
    cache, idx = model.get_seqlen_agnostic_capture_inputs(B)
    idx.copy_(torch.arange(B, dtype=torch.int32, device=dev))

    def set_inputs(step):
        gi = gen(100 + step)
        ids.copy_(torch.randint(0, cfg.vocab_size, (B,), generator=gi).to(dev))
        p = torch.tensor([ctx + step - i for i in range(B)], dtype=torch.int64, device=dev)
        pos.copy_(p)
        seq_lens.copy_((p + 1).to(torch.int32))
        slots.copy_(torch.gather(bt.long(), 1, (p // bs).unsqueeze(1)).squeeze(1) * bs + p % bs)

    tok = torch.zeros(B, dtype=torch.int64, device=dev)
    logits_buf = torch.zeros(B, cfg.vocab_size, dtype=torch.float32, device=dev)

    def run():
        # the step as a greedy decode loop runs it: layers, fp32 logits of the bf16 lm_head, their argmax
        h = model(ids, pos, attn_metadata=md, kv_caches=kv, seqlen_agnostic_capture_inputs=(cache, idx))
        logits_buf.copy_(model.compute_logits(h))
        hip_ops.argmax_rows(logits_buf, out=tok)
        return h

    state0 = (torch.randn(cache[0].shape, device=dev) * 0.1).to(torch.bfloat16)
    kv0 = [k.clone() for k in kv]

    def reset():
        cache[0].copy_(state0)
        for k, k0 in zip(kv, kv0):
            k.copy_(k0)

    set_inputs(0)
    reset()
    run()
    torch.cuda.synchronize()
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
        with torch.cuda.graph(graph, stream=side):
            out = run()
    torch.cuda.synchronize()
    for step in (1, 2, 3):
        set_inputs(step)
        reset()
        eager = run().clone()
        tok_e, logits_e = tok.clone(), logits_buf.clone()
        assert torch.equal(tok_e, logits_e.argmax(-1))
        state_e, kv_e = cache[0].clone(), [k.clone() for k in kv]
        lin(randn_bf16(g_, B, 2048).cuda())          # eager work of the same ops between replays
        reset()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), f"replay {step}: hidden states"
        assert torch.equal(logits_buf, logits_e) and torch.equal(tok, tok_e), f"replay {step}: logits / greedy tokens"
        assert torch.equal(cache[0], state_e) and all(torch.equal(a, b) for a, b in zip(kv, kv_e)), f"replay {step}: caches"
