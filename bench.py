#!/usr/bin/env python3
"""Benchmark of the Hymeta serving hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1]): Hymeta-7B bf16 TP=1, context 8192, block_size 256, greedy
decode of a batch of sequences.  One step = one decode token for every sequence through the WHOLE
28-layer stack of the model class the plugin registers (``GLAswaForCausalLM``, random-init weights of the
published geometry loaded by checkpoint name): block norms, qkv / gate / output projections, GLA decode
(14 layers) or paged sliding-window decode over 4097 keys with the fused KV write (14 layers), gated MLP,
final norm.  The step is replayed from one HIP graph, as the serving engine replays its decode step; the
per-step metadata (positions, sequence lengths, cache slots) lives in static buffers that a few tiny device
ops advance between replays.  Embedding lookup is in the step; LM head and sampler belong to the engine and
are not run.  Inputs are synthetic (seeded) and resident in HBM before timing.

``value`` = output tokens/s of that step; ``roofline`` prices the step against HBM: the bytes a step must
move (every linear's weights once, the visible K/V rows, the GLA states read + written) / step time.

N>1 (python -m torch.distributed.run ... bench.py --gpus N): tensor parallelism over heads with RCCL
all-reduces, see ``tp_main``.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# Hymeta-7B geometry (vllm_hymeta/model_for_7B/configuration_gla_swa.py:15-43 in the reference)
H, HKV, D = 28, 4, 128
HIDDEN, INTER = 3584, 18944
N_LAYERS, N_GLA, N_SWA = 28, 14, 14
WINDOW = 4096          # 7B attends to W+1 keys: flash window_left = W (hf_7B_model/window_attention.py:215)
BLOCK = 256
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
GROUP = 128            # GPTQ group size of the synthetic 4-bit checkpoint
WBYTES = {"bf16": 2.0, "w8a8": 1.0, "gptq4": 0.5, "gptq8": 1.0}
GPTQ_RECIPES = {"gptq4": (4, GROUP), "gptq8": (8, 64)}      # bits, group size (gptq8: hf_7B_model/quant_7B.py:32-36)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="decode sequences (global batch)")
    ap.add_argument("--ctx", type=int, default=None, help="context length (default: 8192 at N=1, 32768 in the tensor-parallel mode)")
    ap.add_argument("--quant", choices=("bf16", "gptq4", "gptq8", "w8a8"), default="bf16", help="linears of the timed model")
    ap.add_argument("--gla-state", choices=("kv_head", "query_head"), default="kv_head",
                    help="GLA state layout: one state per key/value head (the package's default; the 7 query heads "
                         "of a group hold identical states) or one per query head (the reference's cache shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the 32k TTFT / single-request latency side measurements")
    ap.add_argument("--prefill-ctx", type=int, default=32768)
    ap.add_argument("--no-layers-only", action="store_true",
                    help="skip the second timing of the step without lm_head and sampling (profiling runs: every step the same)")
    ap.add_argument("--no-side", action="store_true",
                    help="skip every side measurement (quantised models, kernel table, 70B shard): the timed region only")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs")
    ap.add_argument("--tp-layers", type=int, default=28, help="tensor-parallel mode: layers of the 70B stack (28 = the model)")
    ap.add_argument("--tp-eager", action="store_true", help="tensor-parallel mode: launch the step eagerly instead of from a HIP graph")
    ap.add_argument("--emulate-tp", type=int, default=0, choices=[0, 2, 4],
                    help="with --force-tp on one GPU: build and run rank 0's shard of a group of this size (weight cuts, heads, "
                         "expert shares of one rank; the collectives run as one-rank exchanges) -- timing of one rank's kernels "
                         "at the real shard shapes, not a throughput of the job")
    ap.add_argument("--rccl-only", action="store_true",
                    help="tensor-parallel mode: keep every all-reduce on RCCL (default: the small ones take the one-shot "
                         "exchange over peer-mapped buffers when its self-check passes on every rank)")
    ap.add_argument("--ep", action="store_true",
                    help="tensor-parallel mode: expert parallelism for the mixture layers (the reference's --enable-expert-parallel): "
                         "whole experts dealt to the ranks of a TP group instead of every expert cut along its intermediate size")
    ap.add_argument("--force-tp", action="store_true",
                    help="run the tensor-parallel mode with whatever WORLD_SIZE the launcher set, 1 included (a one-GPU rehearsal "
                         "of the multi-GPU step: the collectives run over a one-rank group)")
    args = ap.parse_args()
    args.ctx_given = args.ctx is not None
    if args.ctx is None:
        args.ctx = 8192
    return args


# --------------------------------------------------------------------------- synthetic checkpoint
VOCAB = 152064


def config_7b(vocab=VOCAB):
    """The published 7B geometry, vocabulary included (the step ends in the lm_head logits and their argmax)."""
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    return GLAswaConfig(vocab_size=vocab)


def synthetic_checkpoint_7b(cfg, dev, quant, seed=0):
    """(name, tensor) pairs of a random-init checkpoint under the published names, generated on the device one
    tensor at a time.  ``quant``: "bf16"; "gptq4" / "gptq8" (gptqmodel tensors: qweight / qzeros / scales / g_idx; 4 bit
    group 128, or the 8 bit group 64 of hf_7B_model/quant_7B.py:32-36; the rank-16 gate projection left in bf16 as
    hf_7B_model/gla_swa_gptq.py:18-24 leaves it); "w8a8"
    (compressed-tensors int8 weight + per-channel weight_scale, hf_model/quant_w8a8.py:121-124)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    hidden, inter = cfg.hidden_size, cfg.intermediate_size
    Hq, Hk, Dh = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device=dev, generator=g, dtype=torch.float32) * scale).to(torch.bfloat16)

    def linear(prefix, out_f, in_f, bias=False, quantise=True):
        if bias:
            yield prefix + ".bias", rnd(out_f, scale=0.02)
        if quant == "bf16" or not quantise:
            yield prefix + ".weight", rnd(out_f, in_f, scale=in_f ** -0.5)
        elif quant == "w8a8":
            yield prefix + ".weight", torch.randint(-127, 128, (out_f, in_f), dtype=torch.int8, device=dev, generator=g)
            yield prefix + ".weight_scale", (torch.rand(out_f, 1, device=dev, generator=g) * (2.0 / 127) * in_f ** -0.5).float()
        else:
            bits, group = GPTQ_RECIPES[quant]
            pack = 32 // bits
            yield prefix + ".qweight", torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // pack, out_f), dtype=torch.int32, device=dev, generator=g)
            yield prefix + ".qzeros", torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // group, out_f // pack), dtype=torch.int32, device=dev, generator=g)
            yield prefix + ".scales", (torch.rand(in_f // group, out_f, device=dev, generator=g) * (4.0 / 2 ** bits * in_f ** -0.5)).to(torch.float16)
            yield prefix + ".g_idx", (torch.arange(in_f, dtype=torch.int32, device=dev) // group)

    yield "model.embeddings.weight", rnd(cfg.vocab_size, hidden, scale=0.5)
    yield "model.norm.weight", torch.ones(hidden, dtype=torch.bfloat16, device=dev)
    yield "lm_head.weight", rnd(cfg.vocab_size, hidden, scale=0.02)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        yield from linear(p + "attn.q_proj", Hq * Dh, hidden, bias=True)
        yield from linear(p + "attn.k_proj", Hk * Dh, hidden, bias=True)
        yield from linear(p + "attn.v_proj", Hk * Dh, hidden, bias=True)
        yield from linear(p + "attn.o_proj", hidden, Hq * Dh)
        if i not in cfg.attn_layers:
            yield from linear(p + "attn.gk_proj.0", 16, hidden, quantise=False)
            yield from linear(p + "attn.gk_proj.1", Hk * Dh, 16, bias=True, quantise=False)
            yield p + "attn.g_norm.weight", torch.ones(Dh, dtype=torch.bfloat16, device=dev)
        yield p + "attn_norm.weight", torch.ones(hidden, dtype=torch.bfloat16, device=dev)
        yield p + "mlp_norm.weight", torch.ones(hidden, dtype=torch.bfloat16, device=dev)
        yield from linear(p + "mlp.gate_proj", inter, hidden)
        yield from linear(p + "mlp.up_proj", inter, hidden)
        yield from linear(p + "mlp.down_proj", hidden, inter)


def build_model_7b(dev, quant, max_num_seqs, gla_state="kv_head", tp_group=None):
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    cfg = config_7b()
    q = {"bits": GPTQ_RECIPES[quant][0], "group_size": GPTQ_RECIPES[quant][1], "zero_offset": 1} if quant in GPTQ_RECIPES else None
    model = GLAswaForCausalLM(cfg, max_num_seqs=max_num_seqs, quant=q, tp_group=tp_group, device=dev,
                              state_per_kv_head=gla_state == "kv_head")
    model.load_weights(synthetic_checkpoint_7b(cfg, dev, quant))
    return model


def linear_weight_bytes(quant, tp=1):
    """Bytes of linear weights one decode step of one rank must read (qkv, o, gate_up, down of every layer)."""
    per_layer = (H + 2 * HKV) * D * HIDDEN + HIDDEN * H * D + 3 * INTER * HIDDEN
    return int(N_LAYERS * per_layer * WBYTES[quant] / tp)


def logits_bytes(B):
    """lm_head (kept in bf16) read once, fp32 logits written and read back by the argmax."""
    return VOCAB * HIDDEN * 2 + 2 * B * VOCAB * 4


def attention_bytes(B, ctx, heads=H, kv_heads=HKV, state_heads=HKV):
    """Algorithmic bytes of the attention kernels of one step: GLA states read + written, visible K/V rows."""
    keys = min(ctx + 1, WINDOW + 1)
    gla = N_GLA * B * (2 * state_heads * D * D * 2 + heads * D * 4 + 3 * kv_heads * D * 2)
    swa = N_SWA * B * (2 * kv_heads * D * 2 * keys + heads * D * 4)
    return gla + swa


# --------------------------------------------------------------------------- decode runner
class DecodeRunner:
    """A batch of ``B`` sequences at context ``ctx`` decoding one token per step through ``model``: owns the paged KV
    caches (random content, scattered blocks), the static per-step metadata and the captured graph of the step."""

    def __init__(self, model, B, ctx, total_steps, dev, seed=0):
        from vllm_hymeta.attention.meta_attention import MetaAttentionMetadata

        self.model, self.B, self.dev = model, B, dev
        m = model.model
        kvh = m.shard.num_kv_heads
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        bps = (ctx + total_steps + 1 + BLOCK - 1) // BLOCK
        nblk = B * bps
        self.kv = [torch.randn(2, nblk, BLOCK, kvh, D, device=dev, dtype=torch.bfloat16) for _ in m.attn_layers]
        self.block_tables = torch.randperm(nblk, generator=g).to(torch.int32).view(B, bps).to(dev)
        self.positions = torch.full((B,), ctx, dtype=torch.int64, device=dev)       # position of the token being decoded
        self.seq_lens = torch.full((B,), ctx + 1, dtype=torch.int32, device=dev)
        self.slot_mapping = torch.zeros(B, dtype=torch.int64, device=dev)
        self.input_ids = torch.randint(0, m.config.vocab_size, (B,), generator=g).to(dev)
        self._refresh_slots()
        cache, idx = model.get_seqlen_agnostic_capture_inputs(B)
        idx.copy_(torch.arange(B, dtype=torch.int32, device=dev))
        cache[0].copy_((torch.randn(cache[0].shape, device=dev, dtype=torch.float32) * 0.1).to(torch.bfloat16))
        self.capture_inputs = (cache, idx)
        self.max_k = ctx + total_steps + 1
        self.md = MetaAttentionMetadata(
            num_prefills=0, num_prefill_tokens=0, num_decode_tokens=B, slot_mapping=self.slot_mapping,
            multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=False, seq_lens=None,
            seq_lens_tensor=self.seq_lens, max_prefill_seq_len=0, max_decode_seq_len=self.max_k,
            context_lens_tensor=None, block_tables=self.block_tables, use_cuda_graph=True, max_query_len=1,
            max_decode_query_len=1, query_start_loc=torch.arange(B + 1, dtype=torch.int32, device=dev),
            seq_start_loc=None)
        self.graph = None
        self.out = None
        self.with_logits = True
        self.sampler = None

    def _refresh_slots(self):
        blk = torch.gather(self.block_tables.long(), 1, (self.positions // BLOCK).unsqueeze(1)).squeeze(1)
        self.slot_mapping.copy_(blk * BLOCK + self.positions % BLOCK)

    def advance(self):
        """The next token's metadata, in place on the device (the engine's multi-step advance: one launch)."""
        from vllm_hymeta import hip_ops

        hip_ops.advance_step(self.B, self.B, BLOCK, self.seq_lens, self.slot_mapping, self.block_tables, input_positions=self.positions)

    def eager_step(self):
        """One greedy decode step: the 28 layers, the lm_head logits of every sequence, their argmax -- which becomes the
        next step's input token, as in the engine's decode loop."""
        from vllm_hymeta import hip_ops

        h = self.model(self.input_ids, self.positions, attn_metadata=self.md, kv_caches=self.kv,
                       seqlen_agnostic_capture_inputs=self.capture_inputs)
        if self.with_logits:
            if self.sampler is not None:
                self.sampler(h, self.input_ids)           # tensor-parallel mode: vocabulary-parallel logits
            else:
                hip_ops.argmax_rows(self.model.compute_logits(h), out=self.input_ids)
        return h

    def capture(self, before_capture=None, capture_error_mode="global"):
        """``before_capture``: called after the eager warm-up, right before recording starts; ``capture_error_mode``: as
        ``torch.cuda.graph`` takes it (the tensor-parallel mode records in "thread_local" mode: the process group's
        watchdog thread makes runtime calls of its own while a capture is open)."""
        for _ in range(2):                       # sizes every workspace, loads every kernel
            self.eager_step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            self.eager_step()
            if before_capture is not None:
                before_capture()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=capture_error_mode):
                self.out = self.eager_step()
        torch.cuda.synchronize()

    def step(self):
        self.advance()
        self.graph.replay()


def timed_steps(runner, warmup, steps, dist=None):
    for _ in range(warmup):
        runner.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        runner.step()
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return elapsed, e0.elapsed_time(e1) / steps


# --------------------------------------------------------------------------- side measurements (N = 1)
def graph_timed(fn, iters=10, reps=3):
    """ms per call of ``fn``, replayed from a HIP graph of ``iters`` calls, so that kernels shorter than a Python
    launch are timed on the GPU's clock."""
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * iters)


def kernel_table(runner, quant):
    """Per-kernel time of the step's launches at the timed shape, each over the 14 / 28 layers' own operands in turn
    (weights and caches cold, as inside the step), with the HBM rate of its algorithmic bytes."""
    from vllm_hymeta import _C, hip_ops

    m, B, dev = runner.model.model, runner.B, runner.dev
    x = (torch.randn(B, HIDDEN, device=dev) * 0.5).to(torch.bfloat16)
    xi = (torch.randn(B, INTER, device=dev) * 0.5).to(torch.bfloat16)
    xa = (torch.randn(B, H * D, device=dev) * 0.5).to(torch.bfloat16)
    layers = list(m.layers)
    rows = {}

    def add(name, fn, count, nbytes):
        ms = graph_timed(fn, iters=1, reps=5) / count
        rows[name] = {"us": round(ms * 1e3, 2), "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1), "bytes": int(nbytes)}

    wb = WBYTES[quant]
    # the projections as the step runs them: the K slices of o_proj / down_proj are summed by the fused add + norm that
    # follows (timed with it), gate_up carries silu * mul in its epilogue where the kernel has one (bf16, W8A8)
    res = torch.zeros(B, HIDDEN, device=dev, dtype=torch.bfloat16)

    def with_norm(proj, inp, norm):
        return lambda: [norm(proj(l)(inp, defer=True), res) for l in layers]

    add("qkv_proj", lambda: [l.attn.qkv_proj(x) for l in layers], len(layers), (H + 2 * HKV) * D * HIDDEN * wb)
    add("o_proj + add_rmsnorm", with_norm(lambda l: l.attn.o_proj, xa, layers[0].mlp_norm), len(layers), HIDDEN * H * D * wb)
    add("gate_up_proj + silu_mul", lambda: [l.mlp.gate_up_proj.forward_silu_mul(x) for l in layers], len(layers), 2 * INTER * HIDDEN * wb)
    add("down_proj + add_rmsnorm", with_norm(lambda l: l.mlp.down_proj, xi, layers[0].attn_norm), len(layers), HIDDEN * INTER * wb)
    qkv = (torch.randn(B, (H + 2 * HKV) * D, device=dev)).to(torch.bfloat16)
    q, k, v = qkv[:, : H * D].view(B, H, D), qkv[:, H * D:(H + HKV) * D].view(B, HKV, D), qkv[:, (H + HKV) * D:].view(B, HKV, D)
    out = torch.empty(B, H, D, device=dev, dtype=torch.bfloat16)
    keys = int(min(int(runner.seq_lens[0]), WINDOW + 1))
    add("paged_decode (+KV write)", lambda: [hip_ops.meta_swa_paged_decode(
        q, kv[0], kv[1], runner.seq_lens, runner.block_tables, window_left=WINDOW, max_seqlen_k=runner.max_k, out=out,
        new_key=k, new_value=v, slot_mapping=runner.slot_mapping) for kv in runner.kv], len(runner.kv),
        B * (2 * HKV * D * 2 * keys + H * D * 4))
    (state,), idx = runner.capture_inputs
    gk = torch.randn(B, HKV, D, device=dev).to(torch.bfloat16)
    g_norm = torch.ones(D, dtype=torch.bfloat16, device=dev)
    sh = state.shape[2]
    add("gla_decode", lambda: [hip_ops.gla_decode(q, k, v, gk, state[i], idx, out=out, gate_mode=_C.GATE_7B,
                                                  out_norm_weight=g_norm, out_norm_eps=1e-6) for i in range(state.shape[0])],
        state.shape[0], B * (2 * sh * D * D * 2 + H * D * 4 + 3 * HKV * D * 2))
    return rows


class PrefillRunner:
    """ONE prompt of ``L`` tokens through the whole model (time to first token), then single-sequence decode."""

    def __init__(self, model, L, new_tokens, dev):
        from vllm_hymeta.attention.meta_attention import MetaAttentionMetadata

        self.model, self.L, self.dev = model, L, dev
        m = model.model
        nblk = (L + new_tokens + 1 + BLOCK - 1) // BLOCK
        self.kv = [torch.zeros(2, nblk, BLOCK, m.shard.num_kv_heads, D, device=dev, dtype=torch.bfloat16) for _ in m.attn_layers]
        g = torch.Generator(device="cpu")
        g.manual_seed(3)
        self.ids = torch.randint(0, m.config.vocab_size, (L,), generator=g).to(dev)
        self.pos = torch.arange(L, dtype=torch.int64, device=dev)
        cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
        self.md = MetaAttentionMetadata(
            num_prefills=1, num_prefill_tokens=L, num_decode_tokens=0, slot_mapping=torch.arange(L, dtype=torch.int64, device=dev),
            multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=False, seq_lens=[L],
            seq_lens_tensor=torch.tensor([L], dtype=torch.int32, device=dev), max_prefill_seq_len=L, max_decode_seq_len=0,
            context_lens_tensor=torch.zeros(1, dtype=torch.int32, device=dev),
            block_tables=torch.zeros(1, 0, dtype=torch.int32, device=dev), use_cuda_graph=False, max_query_len=L,
            max_decode_query_len=1, query_start_loc=cu, seq_start_loc=cu)
        cache, idx = model.get_seqlen_agnostic_capture_inputs(1)
        idx.zero_()
        self.capture_inputs = (cache, idx)

    def prefill(self):
        """The prompt through the model, then the first generated token: the last position's logits and their argmax."""
        from vllm_hymeta import hip_ops

        h = self.model(self.ids, self.pos, attn_metadata=self.md, kv_caches=self.kv,
                       seqlen_agnostic_capture_inputs=self.capture_inputs)
        return hip_ops.argmax_rows(self.model.compute_logits(h[-1:]))


def ttft_only(model, dev, L, runs=3):
    """p50 of the time to first token of ONE ``L``-token prompt (see ``prefill_and_latency``), for the side table of the
    quantised checkpoints."""
    pre = PrefillRunner(model, L, 8, dev)
    pre.prefill()
    torch.cuda.synchronize()
    ttft = []
    for _ in range(runs):
        t0 = time.perf_counter()
        pre.prefill()
        torch.cuda.synchronize()
        ttft.append((time.perf_counter() - t0) * 1e3)
    ttft.sort()
    del pre
    torch.cuda.empty_cache()
    return round(ttft[len(ttft) // 2], 2)


def prefill_and_latency(model, dev, L, new_tokens=128, runs=5):
    """The reference's latency protocol (hf_7B_model/speed_7B.py:196-221, bench/speed_vllm.py:41-53): one request =
    an L-token prompt + 128 generated tokens, batch 1.  TTFT = the prompt through all 28 layers (projections and MLP
    included, library GEMMs at this row count) and the first token's logits + argmax, p50 of ``runs``; decode = 128
    replays of the batch-1 step graph (layers, logits, greedy token)."""
    pre = PrefillRunner(model, L, new_tokens, dev)
    pre.prefill()
    torch.cuda.synchronize()
    ttft = []
    for _ in range(runs):
        t0 = time.perf_counter()
        pre.prefill()
        torch.cuda.synchronize()
        ttft.append((time.perf_counter() - t0) * 1e3)
    ttft.sort()
    dec = DecodeRunner(model, 1, L, new_tokens + 8, dev, seed=5)
    dec.capture()
    total = []
    for _ in range(3):
        dec.positions.fill_(L)
        dec.seq_lens.fill_(L + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pre.prefill()
        for _ in range(new_tokens):
            dec.step()
        torch.cuda.synchronize()
        total.append((time.perf_counter() - t0) * 1e3)
    total.sort()
    t_dec = (total[len(total) // 2] - ttft[len(ttft) // 2]) / new_tokens
    return ({"prompt_tokens": L, "runs": runs, "ttft_ms_p50": round(ttft[len(ttft) // 2], 2), "ttft_ms_min": round(ttft[0], 2),
             "ttft_ms_max": round(ttft[-1], 2), "prefill_tokens_per_s": round(L / (ttft[len(ttft) // 2] * 1e-3), 1)},
            {"protocol": f"1 request: {L}-token prompt + {new_tokens} generated tokens, batch 1 (the reference's speed scripts)",
             "latency_ms_p50": round(total[len(total) // 2], 2), "decode_ms_per_token": round(t_dec, 4),
             "decode_tokens_per_s": round(1e3 / t_dec, 1),
             "reference_published_ms": "2332 +- 19 (7B bf16, HF eager, unstated GPU; hf_7B_model/speed_7B.py:205-208)"})


def shard70b_side_measurement(dev, B=64, ctx=32768, chunk=2048):
    """One tensor-parallel rank (TP=4: 7 q-heads, 1 kv-head, 128 meta tokens) of the 70B family's
    attention hot path at 32k context: a decode step of 28 layers (each: GLA decode with the
    key-derived gate + paged window / full attention over the sequence and the meta keys, KV write
    fused), and one 2048-token chunk of chunked prefill per layer kind.  Outside the timed region."""
    from vllm_hymeta import _C, hip_ops

    Hs, Ks, M, W = 7, 1, 128, 4096
    g = torch.Generator(device="cpu")
    g.manual_seed(2)
    bps = (ctx + 1 + BLOCK - 1) // BLOCK
    nblk = B * bps
    kvs = [torch.randn(2, nblk, BLOCK, Ks, D, device=dev, dtype=torch.bfloat16) for _ in range(2)]   # alternated: cold reads
    bt = torch.randperm(nblk, generator=g).to(torch.int32).view(B, bps).to(dev)
    seq_lens = torch.full((B,), ctx + 1, dtype=torch.int32, device=dev)
    slots = (bt[:, ctx // BLOCK].long() * BLOCK + ctx % BLOCK).contiguous()
    meta_k = torch.randn(M, Ks, D, device=dev).to(torch.bfloat16)
    meta_v = torch.randn(M, Ks, D, device=dev).to(torch.bfloat16)
    states = [(torch.randn(B, Hs, D, D, device=dev) * 0.1).to(torch.bfloat16) for _ in range(2)]
    slot_idx = torch.arange(B, dtype=torch.int32, device=dev)
    qkv = torch.randn(B, (Hs + 2 * Ks) * D, device=dev).to(torch.bfloat16)
    q, k, v = qkv[:, : Hs * D].view(B, Hs, D), qkv[:, Hs * D:(Hs + Ks) * D].view(B, Ks, D), qkv[:, (Hs + Ks) * D:].view(B, Ks, D)
    lb = torch.rand(Hs, D, device=dev) * 0.5
    out = torch.empty(B, Hs, D, device=dev, dtype=torch.bfloat16)
    full_layers = (3, 10, 17, 24)

    def attn(layer):
        wl = -1 if layer in full_layers else W - 1
        kv = kvs[layer & 1]
        hip_ops.meta_swa_paged_decode(q, kv[0], kv[1], seq_lens, bt, meta_key=meta_k, meta_value=meta_v, window_left=wl,
                                      max_seqlen_k=ctx + 1, out=out, new_key=k, new_value=v, slot_mapping=slots)

    def gla(layer):
        gate = dict(gate_mode=_C.GATE_70B_CLAMP, gate_clamp=0.95) if layer % 7 == 0 else \
            dict(gate_mode=_C.GATE_70B_LOWER_BOUND, lower_bound=lb)
        hip_ops.gla_decode(q, k, v, None, states[layer & 1], slot_idx, out=out, **gate)

    def step():
        for layer in range(28):
            gla(layer)
            attn(layer)

    t_step = graph_timed(step, 5)
    t_win, t_full, t_gla = graph_timed(lambda: attn(0)), graph_timed(lambda: attn(3)), graph_timed(lambda: gla(1))
    # algorithmic bytes of the step: states read + written, visible K/V rows (+ meta) per layer kind
    b_gla = B * (2 * Hs * D * D * 2)
    b_win = B * 2 * Ks * D * 2 * (W + M)
    b_full = B * 2 * Ks * D * 2 * (ctx + 1 + M)
    step_bytes = 28 * b_gla + 24 * b_win + 4 * b_full

    # chunked prefill: one chunk of one sequence that already holds ctx - chunk keys
    cq = torch.randn(chunk, Hs, D, device=dev).to(torch.bfloat16)
    cuq = torch.tensor([0, chunk], dtype=torch.int32, device=dev)
    sl1 = torch.tensor([ctx], dtype=torch.int32, device=dev)
    cout = torch.empty(chunk, Hs, D, device=dev, dtype=torch.bfloat16)

    def chunk_attn(wl):
        hip_ops.meta_swa_prefill(cq, cuq, chunk, key_cache=kvs[0][0], value_cache=kvs[0][1], block_tables=bt[:1], seq_lens_k=sl1,
                                 meta_key=meta_k, meta_value=meta_v, window_left=wl, out=cout, max_seqlen_k=ctx)

    ck = torch.randn(chunk, Ks, D, device=dev).to(torch.bfloat16)
    cstate = torch.zeros(1, Hs, D, D, dtype=torch.bfloat16, device=dev)
    mstate = torch.zeros(Hs, D, D, dtype=torch.bfloat16, device=dev)
    ctx_len = torch.tensor([ctx - chunk], dtype=torch.int32, device=dev)

    def chunk_gla():
        hip_ops.gla_prefill_varlen(cq, ck, ck, None, cstate, slot_idx[:1], cuq, chunk, context_lens=ctx_len, meta_state=mstate,
                                   out=cout, gate_mode=_C.GATE_70B_LOWER_BOUND, lower_bound=lb)

    t_cw, t_cf, t_cg = graph_timed(lambda: chunk_attn(W - 1)), graph_timed(lambda: chunk_attn(-1)), graph_timed(chunk_gla)
    return {
        "shape": f"TP=4 rank: {Hs} q-heads, {Ks} kv-head, {M} meta tokens, ctx {ctx}, {B} sequences, block_size {BLOCK}",
        "decode_ms_per_step": round(t_step, 4),
        "decode_tokens_per_s": round(B / (t_step * 1e-3), 1),
        "decode_hbm_gbs": round(step_bytes / (t_step * 1e-3) / 1e9, 1),
        "window_layer_decode_us": round(t_win * 1e3, 1),
        "full_layer_decode_us": round(t_full * 1e3, 1),
        "gla_decode_us": round(t_gla * 1e3, 1),
        f"chunk{chunk}_window_layer_us": round(t_cw * 1e3, 1),
        f"chunk{chunk}_full_layer_us": round(t_cf * 1e3, 1),
        f"chunk{chunk}_gla_us": round(t_cg * 1e3, 1),
        f"chunk{chunk}_attention_ms_28_layers": round(24 * t_cw + 4 * t_cf + 28 * t_cg, 3),
    }


def cpu_baseline(args):
    """The CPU restatement (oracle/) of the same decode step on the host cores, on a bounded sample: 2 sequences at
    the timed context through ONE GLA layer and ONE sliding-window layer of the published geometry (block norms,
    projections, attention, gated MLP; 2 of the 28 layers), scaled to 28 layers."""
    from oracle import gla_decode_step_ref, meta_swa_paged_decode_ref, reshape_and_cache_ref
    from oracle.layers import _linear, add_rms_norm_ref, gate_7b_ref, rms_norm_ref, rope_cos_sin_cache_ref, rope_neox_ref

    Bc, ctx = 2, args.ctx
    torch.manual_seed(0)
    nb = (ctx + 8 + BLOCK - 1) // BLOCK
    kc = torch.randn(Bc * nb, BLOCK, HKV, D).to(torch.bfloat16)
    vc = torch.randn(Bc * nb, BLOCK, HKV, D).to(torch.bfloat16)
    bt = torch.arange(Bc * nb, dtype=torch.int32).view(Bc, nb)
    state = (torch.randn(Bc, H, D, D) * 0.1).to(torch.bfloat16)
    slot = torch.arange(Bc, dtype=torch.int32)
    rb = lambda *s, sc=1.0: (torch.randn(*s) * sc).to(torch.bfloat16)
    qkv_n = (H + 2 * HKV) * D
    lay = [dict(qkv=rb(qkv_n, HIDDEN, sc=HIDDEN ** -0.5), qkv_b=rb(qkv_n, sc=0.02), o=rb(HIDDEN, H * D, sc=HIDDEN ** -0.5),
                gate_up=rb(2 * INTER, HIDDEN, sc=HIDDEN ** -0.5), down=rb(HIDDEN, INTER, sc=INTER ** -0.5)) for _ in range(2)]
    gk0, gk1, gk1_b = rb(16, HIDDEN, sc=HIDDEN ** -0.5), rb(HKV * D, 16, sc=0.25), rb(HKV * D, sc=0.02)
    ones_h, ones_d = torch.ones(HIDDEN, dtype=torch.bfloat16), torch.ones(D, dtype=torch.bfloat16)
    lm_head = rb(VOCAB, HIDDEN, sc=HIDDEN ** -0.5).float()      # the reference widens lm_head to fp32 (modeling_gla_swa.py:548)
    table = rope_cos_sin_cache_ref(D, ctx + 64, 1e6)
    x0 = rb(Bc, HIDDEN, sc=0.5)

    def mlp(p, n):
        gu = _linear(n, p["gate_up"])
        return _linear((torch.nn.functional.silu(gu[:, :INTER].float()) * gu[:, INTER:].float()).to(torch.bfloat16), p["down"])

    def step(s):
        x, pos = x0, ctx + s
        # GLA layer
        h, _ = add_rms_norm_ref(x, None, ones_h)
        qkv = _linear(h, lay[0]["qkv"], lay[0]["qkv_b"])
        q, k, v = qkv[:, : H * D].view(Bc, H, D), qkv[:, H * D:(H + HKV) * D].view(Bc, HKV, D), qkv[:, (H + HKV) * D:].view(Bc, HKV, D)
        gk = _linear(_linear(h, gk0), gk1, gk1_b).view(Bc, HKV, D)
        qg, kg, gg = gate_7b_ref(q, k, gk)
        att = rms_norm_ref(gla_decode_step_ref(qg, kg, v, gg, state, slot), ones_d)
        n, r = add_rms_norm_ref(_linear(att.reshape(Bc, H * D).to(torch.bfloat16), lay[0]["o"]), x, ones_h)
        x = (r.float() + mlp(lay[0], n).float()).to(torch.bfloat16)
        # sliding-window layer
        h, _ = add_rms_norm_ref(x, None, ones_h)
        qkv = _linear(h, lay[1]["qkv"], lay[1]["qkv_b"])
        q, k, v = qkv[:, : H * D].view(Bc, H, D), qkv[:, H * D:(H + HKV) * D].view(Bc, HKV, D), qkv[:, (H + HKV) * D:].view(Bc, HKV, D)
        q, k = rope_neox_ref(torch.full((Bc,), pos), q, k, table)
        slots = bt[:, pos // BLOCK].long() * BLOCK + pos % BLOCK
        reshape_and_cache_ref(k, v, kc, vc, slots)
        att = meta_swa_paged_decode_ref(q, kc, vc, [pos + 1] * Bc, bt, None, None, WINDOW)
        n, r = add_rms_norm_ref(_linear(att.reshape(Bc, H * D).to(torch.bfloat16), lay[1]["o"]), x, ones_h)
        return (r.float() + mlp(lay[1], n).float()).to(torch.bfloat16)

    def head(x):
        """final norm, fp32 logits, greedy token: once per step whatever the layer count"""
        h, _ = add_rms_norm_ref(x, None, ones_h)
        return (h.float() @ lm_head.t()).argmax(dim=-1)

    head(step(0))  # warm-up
    n, t_layers, t_head = 0, 0.0, 0.0
    while True:
        t0 = time.perf_counter()
        x = step(1 + n)
        t1 = time.perf_counter()
        head(x)
        t_layers, t_head = t_layers + (t1 - t0), t_head + (time.perf_counter() - t1)
        n += 1
        if t_layers + t_head > 10.0 or n >= 8:
            break
    per_step_28 = t_layers / n * (N_LAYERS / 2) + t_head / n
    return {"value": round(Bc / per_step_28, 3), "unit": "tokens/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle/ (torch CPU, {torch.get_num_threads()} threads): {Bc} sequences x {n} decode steps at ctx {ctx} through "
                      "one GLA layer + one sliding-window layer of the published geometry (norms, projections, attention, MLP), "
                      "layer time scaled from 2 to 28 layers, plus the final norm, the fp32 lm_head logits and the argmax once per step"}


# --------------------------------------------------------------------------- main (N = 1)
def read_traffic(B, ctx, quant):
    tpath = os.path.join(ROOT, "profiles", "traffic_model_decode.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            if rec.get("batch") == B and rec.get("ctx") == ctx and rec.get("linears", "bf16") == quant:
                return rec.get("hbm_bytes_per_step")
        except Exception:
            pass
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or args.force_tp:
        from bench_tp import tp_main      # tensor-parallel mode: its own file, imported before any GPU call
        return tp_main(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP kernels are the product path; there is no CPU fallback)")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from vllm_hymeta import hip_ops  # noqa: F401  (fails loudly if the HIP library is missing)

    B, total = args.batch, args.warmup + args.steps
    model = build_model_7b(dev, args.quant, max_num_seqs=B, gla_state=args.gla_state)
    runner = DecodeRunner(model, B, args.ctx, 2 * total + 16, dev)
    runner.capture()
    elapsed, gpu_ms = timed_steps(runner, args.warmup, args.steps)

    state_heads = HKV if args.gla_state == "kv_head" else H
    step_bytes = linear_weight_bytes(args.quant) + attention_bytes(B, args.ctx, state_heads=state_heads) + logits_bytes(B)
    achieved = step_bytes / (gpu_ms * 1e-3) / 1e9
    keys = min(args.ctx + 1, WINDOW + 1)
    line = {
        "metric": "decode output tokens/s (whole model step)",
        "value": round(B * args.steps / elapsed, 1),
        "unit": "tokens/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"Hymeta-7B {args.quant} TP=1 decode, ctx {args.ctx}, block_size {BLOCK}, {B} sequences; step = one token per "
                        "sequence through all 28 layers of GLAswaForCausalLM (embedding, norms, qkv/o/gate_up/down projections, 14 x GLA "
                        f"decode, 14 x paged sliding-window decode over {keys} keys with fused KV write, final norm), then the fp32 "
                        "lm_head logits of every sequence and their argmax (greedy sampling), which is the next step's input token; "
                        "one HIP graph per step",
            "batch_per_gpu": B,
            "ctx": args.ctx,
            "linears": args.quant,
            "parallelism": "single GPU",
            "gla_state": ("one state per kv-head, shared by its 7 query heads (identical in the reference)"
                          if args.gla_state == "kv_head" else "one state per query head (the reference's cache shape)"),
        },
        "roofline": {
            "kernel": "whole decode step (one graph launch): linear weights + visible K/V rows + GLA states + lm_head and logits",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": read_traffic(B, args.ctx, args.quant),
            "traffic_source": "profiles/traffic_model_decode.json: FETCH_SIZE / WRITE_SIZE counter passes (rocprofv3 --pmc, separate runs, "
                              "tools/profile_model_step.sh) over this command on an earlier run of this tree; NOT collected by this run "
                              "(null when no recorded pass matches this batch, context and weight format)",
            "algorithmic_bytes_per_launch": int(step_bytes),
            "linear_weight_bytes": linear_weight_bytes(args.quant),
            "attention_bytes": int(attention_bytes(B, args.ctx, state_heads=state_heads)),
            "logits_bytes": logits_bytes(B),
            "avg_launch_ms": round(gpu_ms, 5),
        },
    }
    # the same step without lm_head and sampling (what a captured model forward of the engine holds)
    if not args.no_layers_only:
        runner.with_logits = False
        runner.capture()
        _, layers_ms = timed_steps(runner, args.warmup, args.steps)
        line["layers_only_ms_per_step"] = round(layers_ms, 4)
        runner.with_logits = True
    if not args.no_side:
        line["kernels"] = kernel_table(runner, args.quant)
        if not args.no_prefill:
            line["prefill"], line["latency_b1"] = prefill_and_latency(model, dev, args.prefill_ctx)
            line["ttft_ms_p50"] = line["prefill"]["ttft_ms_p50"]
        del runner
        torch.cuda.empty_cache()
        b1_main = None if args.no_prefill else DecodeRunner(model, 1, args.prefill_ctx, total + 8, dev, seed=7)
        if b1_main is not None:
            b1_main.capture()
            b1_main = round(timed_steps(b1_main, args.warmup, args.steps)[1], 4)
        del model
        torch.cuda.empty_cache()
        def decode_b1(m):
            """ms per generated token of ONE sequence at the prefill context (logits and sampling included)."""
            r1 = DecodeRunner(m, 1, args.prefill_ctx, total + 8, dev, seed=7)
            r1.capture()
            return round(timed_steps(r1, args.warmup, args.steps)[1], 4)

        line["model_decode"] = {args.quant: {"ms_per_step": line["ms_per_step"], "tokens_per_s": line["value"], "batch1_ms_per_token": b1_main}}
        # the same step at other batch sizes (up to 256 rows the linears stream their weights once; beyond, library GEMMs)
        line["decode_by_batch"] = {}
        mb = build_model_7b(dev, args.quant, max_num_seqs=256, gla_state=args.gla_state)
        for bb in (8, 128, 256):
            rb = DecodeRunner(mb, bb, args.ctx, total + 8, dev, seed=bb)
            rb.capture()
            _, msb = timed_steps(rb, args.warmup, args.steps)
            line["decode_by_batch"][str(bb)] = {"ms_per_step": round(msb, 4), "tokens_per_s": round(bb / (msb * 1e-3), 1)}
            del rb
            torch.cuda.empty_cache()
        del mb
        torch.cuda.empty_cache()
        for q in ("bf16", "gptq4", "gptq8", "w8a8"):
            if q == args.quant:
                continue
            mq = build_model_7b(dev, q, max_num_seqs=max(B, 256), gla_state=args.gla_state)
            rq = DecodeRunner(mq, B, args.ctx, total + 8, dev)
            rq.capture()
            el, ms = timed_steps(rq, args.warmup, args.steps)
            sb = linear_weight_bytes(q) + attention_bytes(B, args.ctx, state_heads=state_heads) + logits_bytes(B)
            line["model_decode"][q] = {"ms_per_step": round(ms, 4), "tokens_per_s": round(B / (ms * 1e-3), 1),
                                       "hbm_frac": round(sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernels": kernel_table(rq, q)}
            del rq
            torch.cuda.empty_cache()
            # 256 sequences: the 129..256-row streaming kernels of this weight format
            r256 = DecodeRunner(mq, 256, args.ctx, total + 8, dev, seed=256)
            r256.capture()
            _, ms256 = timed_steps(r256, args.warmup, args.steps)
            line["model_decode"][q]["batch256"] = {"ms_per_step": round(ms256, 4), "tokens_per_s": round(256 / (ms256 * 1e-3), 1)}
            del r256
            torch.cuda.empty_cache()
            line["model_decode"][q]["batch1_ms_per_token"] = decode_b1(mq)
            if not args.no_prefill:
                line["model_decode"][q]["ttft_ms_p50"] = ttft_only(mq, dev, args.prefill_ctx)
            del mq
            torch.cuda.empty_cache()
        line["shard_70b_tp4"] = shard70b_side_measurement(dev)
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
