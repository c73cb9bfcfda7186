"""The ``MetaAttention`` layer object and the two opaque ops
(reference ``vllm_hymeta/models/attention.py:25-409``): registration and shapes on CPU, the
whole path (KV write, prefill with the meta tokens' own rows, paged decode) on the GPU."""
import pytest
import torch

from _util import assert_bf16_close, gen, randn_bf16

D = 128


def test_ops_are_registered_with_the_reference_schemas():
    from vllm_hymeta.models import attention as A

    s1 = str(torch.ops.vllm.unified_meta_attention.default._schema)
    s2 = str(torch.ops.vllm.unified_meta_attention_with_output.default._schema)
    assert "Tensor query, Tensor key, Tensor value, str layer_name, Tensor? query2=None" in s1 and s1.endswith("-> Tensor")
    assert "Tensor(a!) output" in s2 and s2.endswith("-> ()")          # mutates_args=["output"]
    A.release_layer("cpu.attn")
    layer = A.MetaAttention(4, D, D ** -0.5, 2, per_layer_sliding_window=48, prefix="cpu.attn", num_meta_tokens=32)
    assert layer.use_output and layer.impl.num_meta_tokens == 32 and layer.layer_name == "cpu.attn"
    assert len(layer.kv_cache) == 1 and layer.kv_cache[0].numel() == 0
    with pytest.raises(ValueError, match="Duplicate layer name"):
        A.MetaAttention(4, D, 1.0, 2, prefix="cpu.attn")
    with pytest.raises(NotImplementedError):
        A.MetaAttention(4, D, 1.0, 2, prefix="cpu.attn2", kv_sharing_target_layer_name="x")
    # fake (shape-only) implementation: the meta tokens' rows come first in the output
    q, kv, q2 = torch.empty(5, 4 * D, device="meta"), torch.empty(5, 2 * D, device="meta"), torch.empty(32, 4 * D, device="meta")
    assert torch.ops.vllm.unified_meta_attention(q, kv, kv, "cpu.attn", q2, None, None).shape == (37, 4 * D)
    assert torch.ops.vllm.unified_meta_attention(q, kv, kv, "cpu.attn").shape == (5, 4 * D)
    A.release_layer("cpu.attn")


@pytest.mark.gpu
def test_layer_prefill_with_meta_rows_then_paged_decode():
    from oracle import meta_swa_varlen_ref
    from test_gpu_layers import _md
    from vllm_hymeta.models import attention as A

    dev = torch.device("cuda")
    g_ = gen(21)
    H, Hkv, M, W, T, bs = 4, 2, 32, 48, 150, 16
    A.release_layer("gpu.attn")
    layer = A.MetaAttention(H, D, D ** -0.5, Hkv, per_layer_sliding_window=W, prefix="gpu.attn", num_meta_tokens=M)
    q, k, v = randn_bf16(g_, T + 1, H * D), randn_bf16(g_, T + 1, Hkv * D), randn_bf16(g_, T + 1, Hkv * D)
    q2, k2, v2 = randn_bf16(g_, M, H * D), randn_bf16(g_, M, Hkv * D), randn_bf16(g_, M, Hkv * D)
    cu = torch.tensor([0, T + 1], dtype=torch.int32)
    r = lambda t, h: t.view(-1, h, D)
    ref = meta_swa_varlen_ref(r(q, H), r(k, Hkv), r(v, Hkv), cu, cu, r(k2, Hkv), r(v2, Hkv), r(q2, H), W - 1)   # [M + T + 1, H, D]
    ref = ref[0] if isinstance(ref, tuple) else ref

    nblk = (T + 1 + bs - 1) // bs
    layer.kv_cache = [torch.zeros(2, nblk + 1, bs, Hkv, D, dtype=torch.bfloat16, device=dev)]
    bt = torch.randperm(nblk + 1, generator=g_)[:nblk].to(torch.int32).view(1, -1).to(dev)
    slots = torch.stack([bt[0, t // bs].long() * bs + t % bs for t in range(T + 1)])
    c = lambda t: t.to(dev)
    with A.forward_context(_md(dev, prefill_lens=[T], context_lens=[0], slot_mapping=slots[:T])):
        out_pre = layer(c(q[:T]), c(k[:T]), c(v[:T]), query2=c(q2), key2=c(k2), value2=c(v2))
    with A.forward_context(_md(dev, decode_seq_lens=[T + 1], block_tables=bt, slot_mapping=slots[T:])):
        out_dec = layer(c(q[T:]), c(k[T:]), c(v[T:]), key2=c(k2), value2=c(v2))
        out_dec2 = torch.ops.vllm.unified_meta_attention(c(q[T:]), c(k[T:]), c(v[T:]), "gpu.attn", None, c(k2), c(v2))
    torch.cuda.synchronize()
    assert out_pre.shape == (M + T, H * D) and out_dec.shape == (1, H * D)
    assert_bf16_close(out_pre.cpu().view(-1, H, D), ref[: M + T], "layer prefill, meta rows first")
    assert_bf16_close(out_dec.cpu().view(-1, H, D), ref[M + T:], "layer paged decode")
    assert torch.equal(out_dec, out_dec2), "the op without an output buffer is the same computation"
    A.release_layer("gpu.attn")
