"""The model classes as vLLM constructs and drives them, without vLLM: built from a duck-typed ``VllmConfig``
(``HymetaForCausalLM(vllm_config=..., prefix=...)``, reference ``vllm_hymeta/models/modeling_hymeta.py:1386-1411``),
attention layers found by the reference's names in the layer registry, paged KV tensors bound to them the way
vLLM's ``bind_kv_cache`` does, and a call through the opaque op that reaches the backend with that tensor and the
forward context's metadata."""
from types import SimpleNamespace

import pytest
import torch

from _util import gen
from test_gpu_model_70b import _checkpoint as ckpt_70b
from test_gpu_model_7b import _checkpoint as ckpt_7b

D = 128


def vllm_config_like(hf_config, max_num_seqs=5, block_size=256, quant_config=None, tp=1, pp=1, max_model_len=None, device="cpu"):
    return SimpleNamespace(
        model_config=SimpleNamespace(hf_config=hf_config, max_model_len=max_model_len, enforce_eager=True),
        cache_config=SimpleNamespace(block_size=block_size, cache_dtype="auto", sliding_window=None, calculate_kv_scales=False),
        scheduler_config=SimpleNamespace(max_num_seqs=max_num_seqs),
        parallel_config=SimpleNamespace(tensor_parallel_size=tp, pipeline_parallel_size=pp),
        quant_config=quant_config, lora_config=None, device_config=SimpleNamespace(device=device))


@pytest.fixture
def fresh_registry():
    from vllm_hymeta.models import attention
    attention.release_layers("model.")
    attention.release_layers("lm.")
    yield attention
    attention.release_layers("model.")
    attention.release_layers("lm.")


def _cfg70():
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    return HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                        intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                        sliding_window=48, max_position_embeddings=1024)


def _cfg7():
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    return GLAswaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                        intermediate_size=1024, sliding_window=48, max_position_embeddings=1024)


def test_model_classes_carry_the_vllm_interfaces():
    from vllm_hymeta import _compat
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

    for cls in (HymetaForCausalLM, GLAswaForCausalLM):
        # both classes have pipeline stages and say so (vLLM checks ``supports_pp`` and ``make_empty_intermediate_tensors``)
        for mixin in (_compat.HasInnerState, _compat.IsHybrid, _compat.SupportsV0Only, _compat.SupportsPP):
            assert issubclass(cls, mixin)
        assert callable(getattr(cls, "make_empty_intermediate_tensors"))


def test_70b_built_from_vllm_config_registers_and_binds(fresh_registry):
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

    attention = fresh_registry
    cfg = _cfg70()
    model = HymetaForCausalLM(vllm_config=vllm_config_like(cfg, max_num_seqs=7, max_model_len=640), prefix="")
    assert model.config is cfg and cfg.max_model_len == 640                # reference :1401-1402
    m = model.model
    assert m.cache_shape == (4, 7, 4, D, D), "state slots = scheduler_config.max_num_seqs"
    assert m.cos_sin_cache.shape[0] == 640 + 32
    names = sorted(n for n in attention.registered_layers() if n.startswith("model."))
    assert names == [f"model.layers.{i}.attn.vanilla_attn.attn" for i in range(4)]          # reference :742,843,955,1133
    layers = attention.registered_layers()
    windows = [layers[n].impl.sliding_window for n in names]
    assert windows == [(47, 0), (47, 0), (47, 0), (-1, -1)], "three window layers, one full-attention layer"
    assert all(layers[n].num_meta_tokens == 32 and layers[n].kv_cache[0].numel() == 0 for n in names)
    with pytest.raises(ValueError, match="Duplicate layer name"):
        HymetaForCausalLM(vllm_config=vllm_config_like(cfg), prefix="")
    # weights arrive after construction; the layers built then use the registered attention objects
    model.load_weights(ckpt_70b(gen(1), cfg).items())
    for i, blk in enumerate(m.layers):
        assert blk.attn.vanilla_attn.attn is layers[names[i]]

    # bind_kv_cache: one tensor per layer in the backend's shape, by layer name
    from vllm_hymeta.attention.meta_attention import MetaAttentionBackend
    shape = MetaAttentionBackend.get_kv_cache_shape(3, 256, 2, D)
    kv = [torch.full(shape, float(i), dtype=torch.bfloat16) for i in range(4)]
    for i, n in enumerate(names):
        attention.bind_kv_cache(layers[n], kv[i])

    # a call through the opaque op reaches the backend with the bound tensor and the context's metadata
    seen = {}

    def fake_forward(layer, q, k, v, q2, k2, v2, kv_cache, attn_metadata, output, output_scale=None):
        seen.update(layer=layer, kv=kv_cache, md=attn_metadata, q=q.shape, k=k.shape, k2=k2.shape)
        output.fill_(3.0)
        return output

    a = layers[names[2]]
    a.impl.forward = fake_forward
    md = object()
    T = 6
    q, k, v = torch.zeros(T, 4 * D, dtype=torch.bfloat16), torch.zeros(T, 2 * D, dtype=torch.bfloat16), torch.zeros(T, 2 * D, dtype=torch.bfloat16)
    k2 = torch.zeros(32, 2, D, dtype=torch.bfloat16)
    with attention.forward_context(md):
        out = a(q, k, v, query2=None, key2=k2, value2=k2)
    assert out.shape == (T, 4 * D) and bool((out == 3).all())
    assert seen["layer"] is a and seen["md"] is md and seen["kv"] is kv[2]
    assert seen["q"] == (T, 4, D) and seen["k"] == (T, 2, D) and seen["k2"] == (32, 2, D)
    # without metadata in the context (vLLM's memory-profiling pass before any cache exists) the model returns None
    assert model(torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long)) is None


def test_7b_built_from_vllm_config_registers_sliding_window_layers(fresh_registry):
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    attention = fresh_registry
    cfg = _cfg7()
    quant = SimpleNamespace(weight_bits=8, group_size=64, desc_act=False)          # hf_7B_model/quant_7B.py:32-36
    model = GLAswaForCausalLM(vllm_config=vllm_config_like(cfg, max_num_seqs=3, quant_config=quant), prefix="lm")
    assert model.model.quant == dict(bits=8, group_size=64, zero_offset=1)
    names = sorted(n for n in attention.registered_layers() if n.startswith("lm."))
    assert names == ["lm.model.layers.1.attn.attn", "lm.model.layers.3.attn.attn"]            # reference :201,257,346
    for n in names:
        layer = attention.registered_layers()[n]
        assert layer.num_meta_tokens == 0 and layer.impl.sliding_window == (48, 0)            # W keys before the query + itself
    model.load_weights(ckpt_7b(gen(0), cfg).items())
    assert model.model.layers[1].attn.attn is attention.registered_layers()[names[0]]
    logits = model.compute_logits(torch.ones(4, 512, dtype=torch.bfloat16),
                                  SimpleNamespace(selected_token_indices=torch.tensor([1, 3])))
    assert logits.shape == (2, 512) and logits.dtype == torch.float32


def test_unsupported_engine_settings_are_refused(fresh_registry):
    from vllm_hymeta.models.modeling_hymeta import HymetaForCausalLM

    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    cfg = _cfg70()
    with pytest.raises(RuntimeError, match="pipeline_parallel_rank"):
        GLAswaForCausalLM(vllm_config=vllm_config_like(_cfg7(), pp=2))  # no vLLM pipeline group and no rank given
    with pytest.raises(RuntimeError, match="pipeline_parallel_rank"):
        HymetaForCausalLM(vllm_config=vllm_config_like(cfg, pp=2))      # no vLLM pipeline group and no rank given
    with pytest.raises(NotImplementedError, match="desc_act"):
        HymetaForCausalLM(vllm_config=vllm_config_like(cfg, quant_config=SimpleNamespace(weight_bits=4, group_size=128, desc_act=True)))
    with pytest.raises(RuntimeError, match="tensor_parallel_size"):
        HymetaForCausalLM(vllm_config=vllm_config_like(cfg, tp=2))      # no process group of that size in this process


def test_directly_built_models_get_private_layer_names(fresh_registry):
    from vllm_hymeta.model_for_7B.modeling_gla_swa import GLAswaForCausalLM

    cfg = _cfg7()
    a = GLAswaForCausalLM(cfg, max_num_seqs=2, device="cpu")
    b = GLAswaForCausalLM(cfg, max_num_seqs=2, device="cpu")
    na = [l.layer_name for l in a.model.attn_layers.values()]
    nb = [l.layer_name for l in b.model.attn_layers.values()]
    assert len(set(na + nb)) == 4 and all(n.startswith("standalone") for n in na + nb)
