"""BASELINE configs 4 and 5 at the geometry of ONE tensor-parallel rank of the 70B model (TP = 4: 7 query heads,
1 kv-head, 128 meta tokens, hidden 3584, block size 256): one MIXED serving step -- a 2048-token chunk of chunked
prefill of a sequence that already holds cached context, together with a decode batch -- through the intra-layer
hybrid attention block with bf16 linears (config 4), W8A8 linears (config 5: llmcompressor int8 weights, dynamic int8
activations) and GPTQ 4-bit linears, against the eager restatement run over each whole sequence.  The window is cut
to 1024 keys in the windowed case so that it truncates inside a context the fp64 oracle can afford; the 4096-key
window and 32k contexts are covered by the size-independent checks of tests/test_gpu_fullsize.py.  The rank is run as
a whole 7-head model (its two group norms see its own 896 channels); the all-reduced statistics of real tensor
parallelism are checked by the two-shard test of tests/test_gpu_layers.py and the multi-process tests."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16
from test_gpu_layers import _md

pytestmark = pytest.mark.gpu
D, H, HKV, HIDDEN, M_META, BS = 128, 7, 1, 3584, 128, 256
QKV = (H + 2 * HKV) * D


def _linears(kind, g_, dev):
    """(device HymetaLinear, oracle callable) for the two fused qkv projections and out_proj of one rank."""
    from oracle.layers import _linear
    from oracle.quant import gptq_gemm_ref, gptq_pack_ref, w8a8_gemm_ref
    from vllm_hymeta.linear import HymetaLinear

    def make(out_f, in_f, bias):
        b = randn_bf16(g_, out_f, scale=0.1) if bias else None
        bd = None if b is None else b.to(dev)
        if kind == "bf16":
            w = randn_bf16(g_, out_f, in_f, scale=in_f ** -0.5)
            return HymetaLinear.from_weight(w.to(dev), bd), lambda x: _linear(x, w, b)
        if kind == "w8a8":
            wq = torch.randint(-127, 128, (out_f, in_f), generator=g_, dtype=torch.int8)
            ws = (torch.rand(out_f, generator=g_) * 0.5 + 0.75) * (2.0 / 127) * in_f ** -0.5
            return (HymetaLinear.from_w8a8(wq.to(dev), ws.to(dev), bd),
                    lambda x: w8a8_gemm_ref(x, wq, ws, b).to(torch.bfloat16))
        group = 128
        wi = torch.randint(0, 16, (in_f, out_f), generator=g_)
        z = torch.randint(1, 16, (in_f // group, out_f), generator=g_)
        sc = ((torch.rand(in_f // group, out_f, generator=g_) * 0.5 + 0.75) * 0.25 * in_f ** -0.5).to(torch.float16)
        qw, qz = gptq_pack_ref(wi, z, 4)
        return (HymetaLinear.from_gptq(qw.to(dev), qz.to(dev), sc.to(dev), 4, group, bias=bd),
                lambda x: gptq_gemm_ref(x, qw, qz, sc, 4, group, b).to(torch.bfloat16))

    dev_l, ref_l = {}, {}
    for name, out_f, in_f, bias in (("vanilla_attn.qkv", QKV, HIDDEN, True), ("linear_attn.qkv", QKV, HIDDEN, True),
                                    ("out_proj", HIDDEN, H * D, False)):
        dev_l[name], ref_l[name] = make(out_f, in_f, bias)
    return dev_l, ref_l


@pytest.mark.parametrize("kind", ["bf16", "w8a8", "gptq4"])
@pytest.mark.parametrize("layer_idx,window", [(1, 1024), (3, None)])
def test_tp4_rank_mixed_chunked_prefill_and_decode_step(kind, layer_idx, window):
    from oracle.layers import hybrid_layer_70b_ref, lower_bounds_ref
    from vllm_hymeta.model_for_7B.window_attention import rope_cos_sin_cache
    from vllm_hymeta.models.hybrid_attention import HLinearAttention, IntraHybridAttention, MetaFlashAttention
    from vllm_hymeta.parallel import shard_heads

    dev = torch.device("cuda")
    g_ = gen(40 + layer_idx)
    dev_l, ref_l = _linears(kind, g_, dev)
    norms = {"norm1.weight": 1 + 0.1 * torch.randn(H * D, generator=g_), "norm2.weight": 1 + 0.1 * torch.randn(H * D, generator=g_)}
    lb_full = lower_bounds_ref(torch.randn(7, HIDDEN, generator=g_))[layer_idx % 7]      # all 28 heads; rank 0 owns the first 7
    lb = lb_full.view(-1, D)[:H].contiguous()
    meta_x = randn_bf16(g_, M_META, HIDDEN)
    # sequence A: 1024 cached tokens, then a 2048-token chunk; B and C: prompts, then one decode token each
    ctx_a, chunk = 1024, 2048
    lens = {"a": ctx_a + chunk, "b": 700 + 1, "c": 1500 + 1}
    xs = {k: randn_bf16(g_, L, HIDDEN) for k, L in lens.items()}
    max_pos = 4096
    refs = {k: hybrid_layer_70b_ref(xs[k], meta_x, torch.arange(lens[k]), norms, H, HKV, window, lb,
                                    first_layer=(layer_idx % 7 == 0), max_position=max_pos + M_META, linears=ref_l)
            for k in lens}

    table = rope_cos_sin_cache(D, max_pos + M_META, 1e6, device=dev)
    shard = shard_heads(28, 4, 4, 0)
    assert (shard.num_heads, shard.num_kv_heads) == (H, HKV)
    attn = MetaFlashAttention(HIDDEN, 28, 4, dev_l["vanilla_attn.qkv"], table, window, num_meta_tokens=M_META, shard=shard)
    lin = HLinearAttention(HIDDEN, 28, 4, dev_l["linear_attn.qkv"], layer_idx, num_meta_tokens=M_META, shard=shard)
    block = IntraHybridAttention(attn, lin, dev_l["out_proj"], norms["norm1.weight"].to(dev), norms["norm2.weight"].to(dev))
    meta_kv = torch.zeros(2, M_META, HKV, D, dtype=torch.bfloat16, device=dev)
    meta_lin = torch.zeros(H, D, D, dtype=torch.bfloat16, device=dev)
    attn.prepare_meta_state(meta_x.to(dev), meta_kv)
    lin.prepare_meta_state(meta_x.to(dev), meta_lin, lb_full.to(dev))

    order = ["a", "b", "c"]
    nblk = {k: (lens[k] + BS - 1) // BS for k in order}
    total_blocks = sum(nblk.values()) + 2
    perm = torch.randperm(total_blocks, generator=g_).to(torch.int32)
    bt = torch.zeros(3, max(nblk.values()), dtype=torch.int32)
    off = 0
    for i, k in enumerate(order):
        bt[i, : nblk[k]] = perm[off: off + nblk[k]]
        off += nblk[k]
    bt = bt.to(dev)
    kv = torch.zeros(2, total_blocks, BS, HKV, D, dtype=torch.bfloat16, device=dev)
    slot = lambda i, t: int(bt[i, t // BS]) * BS + t % BS
    state = torch.full((4, H, D, D), 3.0, dtype=torch.bfloat16, device=dev)        # stale contents: a new sequence must not see them
    sidx = torch.tensor([2, 0, 3], dtype=torch.int32, device=dev)
    lbd = lb_full.to(dev)

    # step 1: whole-prompt prefill of A's first 1024 tokens and of B's and C's prompts
    pre = [ctx_a, lens["b"] - 1, lens["c"] - 1]
    x1 = torch.cat([xs[k][:n] for k, n in zip(order, pre)]).to(dev)
    pos1 = torch.cat([torch.arange(n) for n in pre]).to(dev)
    slots1 = torch.tensor([slot(i, t) for i, n in enumerate(pre) for t in range(n)], device=dev)
    md1 = _md(dev, prefill_lens=pre, context_lens=[0, 0, 0], slot_mapping=slots1,
              block_tables=torch.zeros(3, 0, dtype=torch.int32, device=dev))
    y1 = block(x1, pos1, kv, state, meta_lin, meta_kv, sidx, md1, lbd)

    # step 2, the mixed step: A's 2048-token chunk through the block table over its cached context + one decode token of B and C
    x2 = torch.cat([xs["a"][ctx_a:], xs["b"][-1:], xs["c"][-1:]]).to(dev)
    pos2 = torch.cat([torch.arange(ctx_a, lens["a"]), torch.tensor([lens["b"] - 1, lens["c"] - 1])]).to(dev)
    slots2 = torch.tensor([slot(0, t) for t in range(ctx_a, lens["a"])] + [slot(1, lens["b"] - 1), slot(2, lens["c"] - 1)], device=dev)
    md2 = _md(dev, prefill_lens=[chunk], context_lens=[ctx_a], decode_seq_lens=[lens["b"], lens["c"]], slot_mapping=slots2, block_tables=bt)
    y2 = block(x2, pos2, kv, state, meta_lin, meta_kv, sidx, md2, lbd)
    torch.cuda.synchronize()

    tol = dict(rel=1.2e-2) if kind == "bf16" else dict(rel=2e-2, frac=2.0 ** -5)
    ref1 = torch.cat([refs[k][:n] for k, n in zip(order, pre)])
    assert_bf16_close(y1.cpu(), ref1, f"TP4 rank, {kind}, layer {layer_idx}: whole-prompt step", **tol)
    assert_bf16_close(y2[:chunk].cpu(), refs["a"][ctx_a:], f"TP4 rank, {kind}, layer {layer_idx}: 2048-token chunk after 1024 cached", **tol)
    assert_bf16_close(y2[chunk:].cpu(), torch.stack([refs["b"][-1], refs["c"][-1]]), f"TP4 rank, {kind}, layer {layer_idx}: decode rows", **tol)
