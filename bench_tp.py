#!/usr/bin/env python3
"""Tensor-parallel mode of bench.py (N > 1): python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N.

Workload (BASELINE.json configs[3] / [4]): the 70B family's decode step at 32k context, one process per GPU, tensor
parallelism over heads inside a group of TP = min(N, 4) ranks (28 query heads and 4 kv-heads admit 2 and 4; the
reference's sharding rules refuse 8, vllm_hymeta/models/modeling_hymeta.py:401-420); N = 8 is two independent TP=4
replicas, each with its own batch.  One step = one token for every sequence of a replica through all 28 layers of
``HymetaForCausalLM``: on every layer the softmax branch (paged window / full attention over the sequence plus the 128
meta keys, fused KV write) and the GLA branch on this rank's 28/TP heads, the ``[T, 2]`` variance all-reduce of the two
group norms (:151-173), ``out_proj`` + all-reduce (:858-861), then the dense MLP or the 16-expert top-1 mixture with
its shared expert + all-reduce (:312-324).  The collectives are torch.distributed on the "nccl" backend (= RCCL over
xGMI), recorded into the step's HIP graph with the kernels.

Random-init weights of the published geometry are generated identically on every rank (same seed) under the checkpoint's
names and cut for the rank by the model's own ``load_weights``.  Rank 0 prints ONE JSON line: ``value`` = output tokens/s
of the whole job (all replicas), the step time is the max over ranks.
"""
from __future__ import annotations

import json
import os
import time
import sys

import torch
import torch.distributed as dist

import bench as B1

FULL_LAYERS = (3, 10, 17, 24)


def config_70b(layers=28, vocab=B1.VOCAB):
    """The published 70B geometry, vocabulary included.  ``layers`` below 28 keeps the first layers (a rehearsal on a
    GPU that cannot hold a 1/TP share of all experts)."""
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    return HymetaConfig(vocab_size=vocab, num_hidden_layers=layers)


def synthetic_checkpoint_70b(cfg, dev, quant, seed=0):
    """(name, tensor) pairs under the published names (hf_model/modeling_hymeta.py parameter names; ``mode.norm`` as the
    checkpoint spells it), one tensor at a time on the device.  ``quant``: the linears as bf16 / W8A8
    (hf_model/quant_w8a8.py:121-132: every Linear but the lm_head, i.e. the routed experts and the router too) / GPTQ-4
    (hf_model/hymeta_gptq.py:22-35: the attention, MLP, shared-expert and routed-expert projections; the router stays
    floating point)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    hidden, inter = cfg.hidden_size, cfg.intermediate_size
    H, HKV, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim

    def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, device=dev, generator=g, dtype=torch.float32) * scale).to(dtype)

    def linear(prefix, out_f, in_f, bias=False, quantise=True):
        if bias:
            yield prefix + ".bias", rnd(out_f, scale=0.02)
        if quant == "bf16" or not quantise:
            yield prefix + ".weight", rnd(out_f, in_f, scale=in_f ** -0.5)
        elif quant == "w8a8":
            yield prefix + ".weight", torch.randint(-127, 128, (out_f, in_f), dtype=torch.int8, device=dev, generator=g)
            yield prefix + ".weight_scale", (torch.rand(out_f, 1, device=dev, generator=g) * (2.0 / 127) * in_f ** -0.5).float()
        else:
            yield prefix + ".qweight", torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), dtype=torch.int32, device=dev, generator=g)
            # zero points 7 or 8 with equal odds (stored as 6 / 7): around them the random nibbles (mean 7.5) give
            # zero-mean weights.  With a mean of -0.5 steps per weight a linear amplifies the common component of its input
            # 6-fold, the hidden states of all sequences end up parallel and the router sends every token to one expert
            coin = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // B1.GROUP, out_f // 8), dtype=torch.int32, device=dev, generator=g)
            yield prefix + ".qzeros", (coin & 0x11111111) + 0x66666666
            # nibbles around the zero point have a standard deviation of 4.6: scales of 0.217 / sqrt(in) on average keep
            # the variance of the activations, as the bf16 and W8A8 weights of this checkpoint do
            yield prefix + ".scales", ((torch.rand(in_f // B1.GROUP, out_f, device=dev, generator=g) * 0.5 + 0.75)
                                       * (0.217 * in_f ** -0.5)).to(torch.float16)
            yield prefix + ".g_idx", (torch.arange(in_f, dtype=torch.int32, device=dev) // B1.GROUP)

    def mlp(prefix, size, quantise=True):
        yield from linear(prefix + ".gate_proj", size, hidden, quantise=quantise)
        yield from linear(prefix + ".up_proj", size, hidden, quantise=quantise)
        yield from linear(prefix + ".down_proj", hidden, size, quantise=quantise)

    yield "model.embeddings.weight", rnd(cfg.vocab_size, hidden, scale=0.5)
    yield "mode.norm.weight", torch.ones(hidden, dtype=torch.bfloat16, device=dev)
    yield "lm_head.weight", rnd(cfg.vocab_size, hidden, scale=0.02)
    yield "model.meta_tokens", rnd(cfg.num_meta_tokens, hidden, scale=0.5)
    yield "model.lower_bounds", rnd(cfg.num_hidden_layers, hidden, dtype=torch.float32)
    ones = lambda n, dt=torch.bfloat16: torch.ones(n, dtype=dt, device=dev)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for br in ("vanilla_attn", "linear_attn"):
            yield from linear(p + f"attn.{br}.q_proj", H * D, hidden, bias=True)
            yield from linear(p + f"attn.{br}.k_proj", HKV * D, hidden, bias=True)
            yield from linear(p + f"attn.{br}.v_proj", HKV * D, hidden, bias=True)
        yield from linear(p + "attn.out_proj", hidden, H * D)
        yield p + "attn.norm1.weight", ones(hidden, torch.float32)
        yield p + "attn.norm2.weight", ones(hidden, torch.float32)
        yield p + "attn_norm.weight", ones(hidden)
        yield p + "mlp_norm.weight", ones(hidden)
        if cfg.num_layer_experts[i] == 1:
            yield from mlp(p + "mlp", inter)
        else:
            if quant == "w8a8":
                gate = rnd(cfg.num_layer_experts[i], hidden, scale=0.5).float()
                gs = gate.abs().amax(dim=1, keepdim=True) / 127.0
                yield p + "block_sparse_moe.gate.weight", torch.round(gate / gs).to(torch.int8)
                yield p + "block_sparse_moe.gate.weight_scale", gs
            else:
                yield p + "block_sparse_moe.gate.weight", rnd(cfg.num_layer_experts[i], hidden, scale=0.5)
            for e in range(cfg.num_layer_experts[i]):
                yield from mlp(p + f"block_sparse_moe.experts.{e}", inter)
            yield from mlp(p + "shared_mlp", cfg.shared_intermediate_size)


def build_model_70b(dev, quant, max_num_seqs, tp_group, layers, expert_parallel=False):
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

    cfg = config_70b(layers)
    q = {"bits": 4, "group_size": B1.GROUP, "zero_offset": 1} if quant == "gptq4" else None
    model = HymetaForCausalLM(cfg, max_num_seqs=max_num_seqs, quant=q, tp_group=tp_group, device=dev, expert_parallel=expert_parallel)
    model.load_weights(synthetic_checkpoint_70b(cfg, dev, quant))
    return model, cfg


class VocabParallelGreedy:
    """Greedy sampling with the lm_head cut along the vocabulary (vLLM's ParallelLMHead; the reference builds it at
    vllm_hymeta/models/modeling_hymeta.py:1412-1431): every rank multiplies the hidden states with its V / tp rows (fp32
    logits of bf16 operands), takes its own argmax, and the ranks exchange one (value, index) pair per sequence -- a
    [B, 2] fp32 all-gather -- instead of gathering [B, V] logits, which greedy sampling does not need.  The lowest rank
    wins among equal values, i.e. the lowest index, as a single argmax would."""

    def __init__(self, model, group, rank_in_group, tp, batch, dev, ranks_in_exchange=None):
        from vllm_hymeta import parallel

        n = ranks_in_exchange or tp           # differs from tp only when one process rehearses a larger group's shard

        V = model._lm_head.shape[0]
        assert V % (64 * tp) == 0, "vocabulary must cut into multiples of 64 rows"
        self.lo = rank_in_group * (V // tp)
        self.weight = model._lm_head[self.lo:self.lo + V // tp].contiguous()
        model._lm_head = self.weight                  # the rank keeps only its rows
        self.group, self.tp = group, tp
        self.rank_in_group = rank_in_group % n
        self.exchange = tp > 1 or parallel.SINGLE_RANK_COLLECTIVES
        self.pair = torch.zeros(batch, 2, dtype=torch.float32, device=dev)
        self.flat = torch.zeros(n * batch, 2, dtype=torch.float32, device=dev)     # the ranks' pairs, rank-major
        self.all = self.flat.view(n, batch, 2)
        self.rows = torch.arange(batch, device=dev)

    def __call__(self, hidden, out_ids):
        from vllm_hymeta import hip_ops
        from vllm_hymeta.lm_head import lm_head_logits

        logits = lm_head_logits(hidden, self.weight)
        idx = hip_ops.argmax_rows(logits) if logits.is_cuda else logits.argmax(-1)     # (CPU: the gloo test of the exchange)
        from vllm_hymeta import parallel

        self.pair[:, 0] = logits.gather(1, idx.unsqueeze(1)).squeeze(1)
        self.pair[:, 1] = (idx + self.lo).float()      # indices below 2^24: exact in fp32
        if self.exchange:      # through the group's one-shot exchange when it has one: a captured step then holds no backend call
            parallel.all_gather_rows(self.pair, self.all, self.group, self.rank_in_group)
        else:
            self.all[0].copy_(self.pair)
        best = self.all[:, :, 0].argmax(0)
        out_ids.copy_(self.all[best, self.rows, 1].long())


def experts_hit(model, runner):
    """Distinct experts the batch routes to, per mixture layer -- with expert parallelism: those among them that this
    rank owns (one eager step with a hook on each mixture: the host read stays outside the timed region)."""
    from vllm_hymeta.models.moe import HymetaMoE

    hit, hooks = [], []

    def pre(mod, inputs):
        x = inputs[0].reshape(-1, inputs[0].shape[-1])
        if mod.w8a8_gate is not None:
            logits = mod.w8a8_gate(x)[:, :mod.gate_weight.shape[0]]
        else:
            logits = torch.nn.functional.linear(x, mod.gate_weight)
        top = logits.float().topk(mod.top_k, dim=-1).indices.unique()
        if mod.local_experts is not None:
            top = top[(top >= mod.local_experts[0]) & (top < mod.local_experts[1])]
        hit.append(int(top.numel()))

    for mod in model.modules():
        if isinstance(mod, HymetaMoE):
            hooks.append(mod.register_forward_pre_hook(pre))
    runner.eager_step()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    return hit


def step_bytes_70b(cfg, quant, tp, batch, ctx, hit, expert_parallel=False):
    """Algorithmic HBM bytes of one step of ONE rank: every linear's weight share once, the routed experts' shares the
    batch touches (a 1 / tp slice of every expert that is hit; with expert parallelism the whole of those the rank owns:
    ``hit`` then counts the rank's own), the visible K/V rows (window + meta, or the whole sequence on the 4
    full-attention layers) of this rank's kv-heads, the GLA states of its heads read + written."""
    wb = B1.WBYTES[quant]
    L = cfg.num_hidden_layers
    HIDDEN, INTER, H, HKV, D = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    M_META = cfg.num_meta_tokens
    attn_w = L * (2 * (H + 2 * HKV) * D * HIDDEN + HIDDEN * H * D) * wb / tp
    moe_layers = [i for i in range(L) if cfg.num_layer_experts[i] > 1]
    dense_w = (L - len(moe_layers)) * 3 * INTER * HIDDEN * wb / tp
    shared_w = len(moe_layers) * 3 * cfg.shared_intermediate_size * HIDDEN * wb / tp
    expert_w = sum(hit) * 3 * INTER * HIDDEN * wb / (1 if expert_parallel else tp)
    kvh, heads = max(1, HKV // tp), H // tp
    full = [i for i in range(L) if i in FULL_LAYERS]
    kv = batch * 2 * kvh * D * 2 * ((L - len(full)) * (min(ctx + 1, cfg.sliding_window) + M_META) + len(full) * (ctx + 1 + M_META))
    gla = L * batch * 2 * heads * D * D * 2
    logits = cfg.vocab_size // tp * HIDDEN * 2 + 2 * batch * (cfg.vocab_size // tp) * 4     # bf16 lm_head rows, fp32 logits out and back
    parts = {"attention_linears": attn_w, "dense_mlp": dense_w, "shared_expert": shared_w, "routed_experts": expert_w,
             "kv_rows": kv, "gla_states": gla, "lm_head_and_logits": logits}
    return int(sum(parts.values())), {k: int(v) for k, v in parts.items()}


class MixedChunkStep:
    """One serving step of chunked prefill beside a decode batch (the engine's mixed step): a ``chunk``-token piece of a
    prompt that already has ``ctx`` tokens cached, and one token each of ``n_decode`` sequences at ``ctx`` tokens, through
    the whole rank (eager launches: the engine does not capture prefill steps)."""

    def __init__(self, model, chunk, ctx, n_decode, dev):
        from vllm_hymeta.attention.meta_attention import MetaAttentionMetadata

        self.model = model
        m = model.model
        per_seq = (ctx + chunk + B1.BLOCK - 1) // B1.BLOCK
        nseq = 1 + n_decode
        self.kv = [torch.randn(2, nseq * per_seq, B1.BLOCK, m.shard.num_kv_heads, B1.D, device=dev, dtype=torch.bfloat16)
                   for _ in m.attn_layers]
        bt = torch.arange(nseq * per_seq, dtype=torch.int32, device=dev).view(nseq, per_seq)
        g = torch.Generator(device="cpu")
        g.manual_seed(11)
        T = chunk + n_decode
        self.ids = torch.randint(0, m.config.vocab_size, (T,), generator=g).to(dev)
        self.pos = torch.cat([torch.arange(ctx, ctx + chunk), torch.full((n_decode,), ctx)]).to(dev)
        slots = torch.cat([bt[0, 0] * B1.BLOCK + torch.arange(ctx, ctx + chunk, device=dev),
                           bt[1:, 0] * B1.BLOCK + ctx]).long()
        qlens = [chunk] + [1] * n_decode
        seq_lens = [ctx + chunk] + [ctx + 1] * n_decode
        cu = lambda xs: torch.tensor([0] + list(torch.tensor(xs).cumsum(0)), dtype=torch.int32, device=dev)
        self.md = MetaAttentionMetadata(
            num_prefills=1, num_prefill_tokens=chunk, num_decode_tokens=n_decode, slot_mapping=slots,
            multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=False, seq_lens=seq_lens,
            seq_lens_tensor=torch.tensor(seq_lens, dtype=torch.int32, device=dev), max_prefill_seq_len=ctx + chunk,
            max_decode_seq_len=ctx + 1 if n_decode else 0,
            context_lens_tensor=torch.tensor([ctx] + [ctx] * n_decode, dtype=torch.int32, device=dev), block_tables=bt,
            use_cuda_graph=False, max_query_len=chunk, max_decode_query_len=1, query_start_loc=cu(qlens), seq_start_loc=cu(seq_lens))
        cache, idx = model.get_seqlen_agnostic_capture_inputs(nseq)
        idx.copy_(torch.arange(nseq, dtype=idx.dtype, device=dev))
        self.capture_inputs = (cache, idx)

    def step(self):
        return self.model(self.ids, self.pos, attn_metadata=self.md, kv_caches=self.kv,
                          seqlen_agnostic_capture_inputs=self.capture_inputs)


def time_mixed_chunks(model, ctx, dev, dist_=None):
    """ms per mixed step (max over ranks) for the chunk sizes of the serving configurations: 2048 and 8192 prompt tokens
    beside 32 decode rows at ``ctx`` cached tokens."""
    out = {}
    for chunk in (2048, 8192):
        stepper = MixedChunkStep(model, chunk, ctx - chunk, 32, dev)
        stepper.step()
        torch.cuda.synchronize()
        if dist_ is not None:
            dist_.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            stepper.step()
        torch.cuda.synchronize()
        ms = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], dtype=torch.float64, device=dev)
        if dist_ is not None:
            dist_.all_reduce(ms, op=dist_.ReduceOp.MAX)
        out[str(chunk)] = {"ms_per_step": round(float(ms[0]), 3), "prompt_tokens_per_s": round(chunk / float(ms[0]) * 1e3, 1)}
        del stepper
        torch.cuda.empty_cache()
    return out


def settle():
    """Called right before a stream capture.  What used to take the process down here: the process group's watchdog
    thread polls the end events of the eager collectives it still holds, those events sit on the group's communication
    stream, and a poll of one while that stream is being captured is refused by HIP (hipErrorCapturedEvent).  The cause is
    removed in ``tp_main``, not waited out: collectives recorded into a graph go over a process group of their own
    (``parallel.set_capture_group``) whose stream never holds an eager event, the capture runs in thread-local error mode
    (another thread's runtime calls are not the capture's business), the event cache is off so that eager work never
    holds an event a capture used, and with the one-shot exchange a captured step holds no backend call at all.  Left
    here: the device is idle when recording starts."""
    torch.cuda.synchronize()


def note(rank, msg):
    if rank == 0:
        print(f"[bench_tp] {msg}", file=sys.stderr, flush=True)


def collectives_alone(cfg, batch, group, dev, reps=5):
    """ms of the step's collectives replayed alone from one graph (per layer: the [B, 2] fp32 statistics, the out_proj
    output and the feed-forward output, [B, hidden] bf16 each; per step: the sampler's [B, 2] all-gather), back to back
    with no compute between them."""
    stats = torch.zeros(batch, 2, dtype=torch.float32, device=dev)
    y = torch.zeros(batch, cfg.hidden_size, dtype=torch.bfloat16, device=dev)
    gathered = torch.zeros(dist.get_world_size(group), batch, 2, dtype=torch.float32, device=dev)

    from vllm_hymeta import parallel

    def run():
        for _ in range(cfg.num_hidden_layers):
            parallel.all_reduce_variance(stats, group)      # the calls the layers make: one-shot exchange or RCCL
            parallel.all_reduce_sum(y, group)
            parallel.all_reduce_sum(y, group)
        parallel.all_gather_rows(stats, gathered, group, dist.get_rank(group))

    run()
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
        settle()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            run()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def tp_layout(world):
    """(tensor-parallel degree, replicas): 4 ranks per group when the job has a multiple of 4, else 2, else 1 (28 query
    heads / 4 kv-heads: the reference's sharding rules admit 1, 2 and 4)."""
    tp = 4 if world % 4 == 0 else 2 if world % 2 == 0 else 1
    return tp, world // tp


def tp_plan(rank, world):
    """Where ``rank`` of a job of ``world`` processes stands: the tensor-parallel degree and replica count
    (``tp_layout``), its replica and its rank inside that replica's group, the ranks of every group (every process
    creates every group, in this order), and the seed of its replica's batch (the ranks of a group decode the same
    sequences; replicas decode different ones)."""
    tp, replicas = tp_layout(world)
    return {"tp": tp, "replicas": replicas, "replica": rank // tp, "rank_in_group": rank % tp,
            "group_ranks": [list(range(r * tp, (r + 1) * tp)) for r in range(replicas)], "batch_seed": rank // tp}


class EagerRunner:
    """The same step without a graph (``--tp-eager``)."""

    def __init__(self, runner):
        self.r = runner

    def step(self):
        self.r.advance()
        self.r.eager_step()


def tp_main(args):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU per rank (the HIP kernels are the product path; there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")      # see settle()
    dist.init_process_group(backend=args.dist_backend, device_id=dev if args.dist_backend == "nccl" else None)
    from vllm_hymeta import hip_ops, parallel  # noqa: F401  (fails loudly if the HIP library is missing)

    plan = tp_plan(rank, world)
    tp, replicas = plan["tp"], plan["replicas"]
    groups = [dist.new_group(ranks) for ranks in plan["group_ranks"]]   # every rank creates every group
    tp_group = groups[plan["replica"]]
    # the collectives of captured steps go over a second group of the same ranks (see settle()); one eager call now, long
    # before the first capture, builds its communicator
    capture_groups = [dist.new_group(ranks) for ranks in plan["group_ranks"]]
    if args.dist_backend == "nccl":
        dist.all_reduce(torch.zeros(1, device=dev), group=capture_groups[plan["replica"]])
        parallel.set_capture_group(tp_group, capture_groups[plan["replica"]])
    if tp == 1:
        parallel.SINGLE_RANK_COLLECTIVES = True      # one-GPU rehearsal: keep the collectives in the step
    shard_tp = tp
    if world == 1 and args.emulate_tp:
        shard_tp = args.emulate_tp
        parallel.EMULATED_TP = (0, shard_tp)         # rank 0's shard of a larger group, collectives as one-rank exchanges

    # The layers' small all-reduces through the one-shot exchange over peer-mapped buffers, if it can be set up on every
    # rank and reproduces the rank-ordered sums bit for bit here and now; RCCL otherwise (and for anything larger).
    all_reduce_path, why = "rccl", "switched off (--rccl-only)"
    if not args.rccl_only:
        try:
            parallel.enable_one_shot_all_reduce(tp_group, dev)
            ok, why = parallel.one_shot_self_check(tp_group, dev)
        except RuntimeError as exc:
            ok, why = False, str(exc)
        if ok:
            all_reduce_path = "one-shot exchange over peer-mapped buffers (self-checked against rank-ordered sums); RCCL above 2 MiB"
        else:
            parallel.disable_one_shot_all_reduce(tp_group)
    note(rank, f"all-reduce path: {all_reduce_path}" + ("" if all_reduce_path != "rccl" else f" ({why})"))

    batch, ctx, total = args.batch, args.ctx if args.ctx_given else 32768, args.warmup + args.steps

    mixed = {}        # chunked-prefill steps per weight format, filled by measure()

    def measure(quant):
        """Build the rank's model from the synthetic checkpoint of one weight format, capture its step, time it: every
        rank makes the same calls (the step holds collectives).  The model is gone when this returns."""
        note(rank, f"{world} ranks: tp{tp} x {replicas}; building the {quant} model")
        model, cfg_ = build_model_70b(dev, quant, batch, tp_group, args.tp_layers, expert_parallel=args.ep)
        runner = B1.DecodeRunner(model, batch, ctx, total + 8, dev, seed=plan["batch_seed"])
        runner.sampler = VocabParallelGreedy(model, tp_group, plan["rank_in_group"], shard_tp, batch, dev, ranks_in_exchange=tp)
        hit_ = experts_hit(model, runner)
        note(rank, f"model built ({torch.cuda.memory_allocated(dev) / 2 ** 30:.1f} GiB allocated), experts hit {hit_}")
        if args.tp_eager:
            for _ in range(2):
                runner.eager_step()
            stepper = EagerRunner(runner)
        else:
            runner.capture(before_capture=settle, capture_error_mode="thread_local")
            stepper = runner
        torch.cuda.synchronize()
        note(rank, "step ready; timing")
        elapsed_, gpu_ms_ = B1.timed_steps(stepper, args.warmup, args.steps, dist=dist)
        t_ = torch.tensor([elapsed_, gpu_ms_], dtype=torch.float64, device=dev)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        del stepper, runner
        torch.cuda.empty_cache()
        if not args.no_side and not args.no_prefill:
            try:
                mixed[quant] = time_mixed_chunks(model, ctx, dev, dist)
            except Exception as exc:
                mixed[quant] = {"error": f"{type(exc).__name__}: {exc}"}
        del model
        from vllm_hymeta.models import attention
        for name in list(attention.registered_layers()):      # the registry of attention layers holds the model's KV caches
            attention.release_layers(name)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        note(rank, f"model released ({torch.cuda.memory_allocated(dev) / 2 ** 30:.1f} GiB still allocated)")
        return float(t_[0]), float(t_[1]), hit_, cfg_

    elapsed, gpu_ms, hit, cfg = measure(args.quant)
    note(rank, f"timed: {gpu_ms:.3f} ms/step; timing the collectives alone")
    coll_ms = collectives_alone(cfg, batch, tp_group, dev)
    c = torch.tensor([coll_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.MAX)
    coll_ms = float(c[0])

    comm = parallel._ONE_SHOT.get(tp_group)
    t = torch.tensor([float(comm.timeouts()) if comm is not None else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    one_shot_timeouts = int(t[0])

    nbytes, parts = step_bytes_70b(cfg, args.quant, shard_tp, batch, ctx, hit, expert_parallel=args.ep and shard_tp > 1)
    achieved = nbytes / (gpu_ms * 1e-3) / 1e9
    line = {
        "metric": "decode output tokens/s (whole model step)",
        "value": round(replicas * batch * args.steps / elapsed, 1),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if replicas == 1 else "strong inside a replica (TP), weak over replicas",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"Hymeta-70B {args.quant} TP={tp} decode, ctx {ctx}, block_size {B1.BLOCK}, {batch} sequences per replica; step = "
                        f"one token per sequence through {cfg.num_hidden_layers} layers of HymetaForCausalLM (embedding, norms, both "
                        "attention branches on the rank's heads, out_proj, dense MLP / 16-expert top-1 mixture + shared expert, final "
                        "norm) with the three all-reduces of every layer (config.all_reduce names their path), then greedy sampling with the lm_head cut along "
                        "the vocabulary (fp32 logits of the rank's rows, local argmax, one [B, 2] all-gather), fed back as the "
                        "next step's tokens.  "
                        "The N=1 line of this bench is the 7B model: a different workload, not the base of this curve",
            "batch_per_replica": batch,
            "global_batch": replicas * batch,
            "ctx": ctx,
            "linears": args.quant,
            "layers": cfg.num_hidden_layers,
            "parallelism": f"tp{tp}" + (" + expert parallelism (whole experts per rank, one all-reduce)" if args.ep and shard_tp > 1 else "") +
                           (f" x {replicas} replicas" if replicas > 1 else "") +
                           (f"; REHEARSAL of rank 0's shard of tp{shard_tp} on one GPU (one rank's kernels at the shard shapes, "
                            "collectives as one-rank exchanges): not a throughput of the job" if shard_tp != tp else ""),
            "rccl_ranks": world,
            "dist_backend": args.dist_backend,
            "all_reduce": all_reduce_path,
            "graph": not args.tp_eager,
        },
        "collectives": {
            "per_step": 3 * cfg.num_hidden_layers + 1,
            "messages": f"per layer: [{batch}, 2] fp32 statistics, 2 x [{batch}, {cfg.hidden_size}] bf16; per step: one [{batch}, 2] fp32 "
                        "all-gather of the ranks' best (logit, token) pairs",
            "ms_per_step_alone": round(coll_ms, 4),
            "share_of_step": round(coll_ms / gpu_ms, 4),
            "one_shot_timeouts": one_shot_timeouts,      # not 0 = a wait for a peer ran into its limit: the line is void
        },
        "roofline": {
            "kernel": "whole decode step of one rank (one graph launch)",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": B1.HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / B1.HBM_PEAK_GBS, 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": nbytes,
            "bytes": parts,
            "experts_hit_per_mixture_layer": hit,
            "avg_launch_ms": round(gpu_ms, 5),
        },
    }
    # the same step from the quantised checkpoints (experts and, for W8A8, the router quantised as the recipes do)
    if args.quant == "bf16" and not args.no_side:
        line["model_decode"] = {"bf16": {"ms_per_step": line["ms_per_step"], "tokens_per_s": line["value"], "hbm_frac": line["roofline"]["frac"]}}
        for q in ("gptq4", "w8a8"):
            try:
                el, ms, hit_q, _ = measure(q)
                nb, _ = step_bytes_70b(cfg, q, shard_tp, batch, ctx, hit_q, expert_parallel=args.ep and shard_tp > 1)
                line["model_decode"][q] = {"ms_per_step": round(el / args.steps * 1e3, 4),
                                           "tokens_per_s": round(replicas * batch * args.steps / el, 1),
                                           "hbm_frac": round(nb / (ms * 1e-3) / 1e9 / B1.HBM_PEAK_GBS, 4),
                                           "algorithmic_bytes_per_launch": nb, "experts_hit_per_mixture_layer": hit_q}
            except Exception as exc:      # the headline line stands on its own
                line["model_decode"][q] = {"error": f"{type(exc).__name__}: {exc}"}
                break
    if mixed:
        line["chunked_prefill"] = {
            "step": "one chunk of a prompt with the rest of the context cached + 32 decode rows at that context, whole rank, "
                    "eager launches, all-reduces included; ms = max over ranks",
            "ctx": ctx, "by_format": mixed}
    if rank == 0:
        print(json.dumps(line), flush=True)
    parallel.disable_one_shot_all_reduce(tp_group)
    parallel.set_capture_group(tp_group, None)
    dist.barrier()
    dist.destroy_process_group()
