"""Model classes without a GPU: the registry's targets exist, checkpoints load by name into the
right modules and shapes for every tensor-parallel rank, and the whole-model restatements run."""
import importlib

import pytest
import torch

from _util import gen
from test_gpu_model_70b import _checkpoint as ckpt_70b
from test_gpu_model_7b import _checkpoint as ckpt_7b

D = 128


def test_registered_model_classes_exist():
    for target in ("vllm_hymeta.model_for_7B.modeling_gla_swa:GLAswaForCausalLM",
                   "vllm_hymeta.models.modeling_hymeta:HymetaForCausalLM"):
        module, cls = target.split(":")
        c = getattr(importlib.import_module(module), cls)
        assert c.packed_modules_mapping == {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}
        for method in ("forward", "compute_logits", "load_weights", "copy_inputs_before_cuda_graphs",
                       "get_seqlen_agnostic_capture_inputs", "get_input_embeddings"):
            assert callable(getattr(c, method))
    import vllm_hymeta.model_for_7B as m7
    import vllm_hymeta.models as m70
    src = open(m7.__file__).read() + open(m70.__file__).read()
    assert "modeling_gla_swa:GLAswaForCausalLM" in src and "modeling_hymeta:HymetaForCausalLM" in src


@pytest.mark.parametrize("tp", [1, 2])
def test_7b_checkpoint_loads_per_rank(tp, monkeypatch):
    from vllm_hymeta.model_for_7B import modeling_gla_swa as mod
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig

    cfg = GLAswaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=1024, sliding_window=48, max_position_embeddings=1024)
    ckpt = ckpt_7b(gen(0), cfg)
    for rank in range(tp):
        monkeypatch.setattr(mod, "tp_rank_and_size", lambda group, rank=rank: (rank, tp))
        model = mod.GLAswaForCausalLM(cfg, max_num_seqs=3, device="cpu")
        loaded = model.load_weights(ckpt.items())
        assert len(loaded) == 39
        m = model.model
        # one GLA state per key/value head of the rank (the reference's per-query-head shape on request)
        assert m.cache_shape == (2, 3, 2 // tp, D, D) and m.gla_cache.cache[0].shape == m.cache_shape
        per_head = mod.GLAswaModel(cfg, max_num_seqs=3, device="cpu", state_per_kv_head=False)
        assert per_head.cache_shape == (2, 3, 4 // tp, D, D)
        assert len(m.layers) == 4 and len(model.kv_cache) == 2
        H, Hkv = 4 // tp, 2 // tp
        for i, layer in enumerate(m.layers):
            # a GLA layer's qkv projection carries the gate's first linear: 16 more rows, padded to 64
            gate_rows = 64 if i % 2 == 0 else 0
            assert layer.attn.qkv_proj.weight.shape == ((H + 2 * Hkv) * D + gate_rows, 512)
            if gate_rows:
                assert layer.attn.gate_in_qkv
                assert torch.equal(layer.attn.qkv_proj.weight[(H + 2 * Hkv) * D:(H + 2 * Hkv) * D + 16], layer.attn.gk_proj0.weight)
                assert not layer.attn.qkv_proj.weight[(H + 2 * Hkv) * D + 16:].any() and not layer.attn.qkv_proj.bias[(H + 2 * Hkv) * D:].any()
            assert layer.attn.o_proj.weight.shape == (512, H * D)
            assert layer.mlp.gate_up_proj.weight.shape == (2 * 1024 // tp, 512)
            assert layer.mlp.down_proj.weight.shape == (512, 1024 // tp)
            assert (type(layer.attn).__name__ == "SlidingWindowAttention") == (i % 2 == 1)
            if i % 2 == 0:
                assert layer.attn.gk_proj1.weight.shape == (Hkv * D, 16) and layer.attn.gk_proj0.weight.shape == (16, 512)
        # rank r holds q-heads [r*H, (r+1)*H) of the checkpoint
        q = ckpt["model.layers.0.attn.q_proj.weight"]
        assert torch.equal(m.layers[0].attn.qkv_proj.weight[: H * D], q[rank * H * D:(rank + 1) * H * D])
        assert model.lm_head.dtype == torch.float32 and m.embeddings.dtype == torch.bfloat16


@pytest.mark.parametrize("tp", [1, 2])
def test_70b_checkpoint_loads_per_rank(tp, monkeypatch):
    from vllm_hymeta.models import modeling_hymeta as mod
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig
    from oracle.layers import lower_bounds_ref

    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    ckpt = ckpt_70b(gen(1), cfg)
    for rank in range(tp):
        monkeypatch.setattr(mod, "tp_rank_and_size", lambda group, rank=rank: (rank, tp))
        model = mod.HymetaForCausalLM(cfg, max_num_seqs=3, device="cpu")
        model.load_weights(ckpt.items())
        m = model.model
        H, Hkv = 4 // tp, 2 // tp
        assert m.cache_shape == (4, 3, H, D, D)
        assert m.meta_linear_cache_shape == (4, H, D, D) and m.meta_fattn_cache_shape == (4, 2, 32, Hkv, D)
        assert torch.equal(m.lower_bounds, lower_bounds_ref(ckpt["model.lower_bounds"]))
        assert not m.has_meta_cache, "the meta pass needs the GPU kernels; without it forward() must refuse"
        for i, layer in enumerate(m.layers):
            a = layer.attn
            assert a.vanilla_attn.qkv_proj.weight.shape == ((H + 2 * Hkv) * D, 512)
            assert a.linear_attn.qkv_proj.weight.shape == ((H + 2 * Hkv) * D, 512)
            assert a.out_proj.weight.shape == (512, H * D)
            assert torch.equal(a.norm1_weight, ckpt[f"model.layers.{i}.attn.norm1.weight"][rank * H * D:(rank + 1) * H * D])
            if i < 3:
                assert layer.mlp.gate_up_proj.weight.shape == (2 * 512 // tp, 512)
            else:
                moe = layer.mlp
                assert moe.w13.shape == (4, 2 * 512 // tp, 512) and moe.w2.shape == (4, 512, 512 // tp)
                assert moe.gate_weight.shape == (4, 512) and moe.top_k == 1 and not moe.renormalize
                assert moe.shared_mlp.gate_up_proj.weight.shape == (2 * 256 // tp, 512) and not moe.shared_mlp.reduce_results


def test_pipeline_stage_layer_ranges():
    """vLLM's partition rule: equal shares, the remainder one each to the stages counted back from the second to last."""
    from vllm_hymeta.models.modeling_hymeta import pp_layer_range

    assert [pp_layer_range(28, r, 2) for r in range(2)] == [(0, 14), (14, 28)]
    assert [pp_layer_range(28, r, 4) for r in range(4)] == [(0, 7), (7, 14), (14, 21), (21, 28)]
    assert [pp_layer_range(30, r, 4) for r in range(4)] == [(0, 7), (7, 15), (15, 23), (23, 30)]
    assert [pp_layer_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert pp_layer_range(28, 0, 1) == (0, 28)
    for L, n in ((28, 3), (31, 8), (7, 7)):
        spans = [pp_layer_range(L, r, n) for r in range(n)]
        assert spans[0][0] == 0 and spans[-1][1] == L and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(ValueError):
        pp_layer_range(28, 2, 2)


def test_70b_checkpoint_loads_per_pipeline_stage():
    """A stage keeps its own layers (caches and attention layers sized to them, registered under their global names), the
    first one the embedding and the meta tokens, the last one the final norm and the lm_head."""
    from vllm_hymeta.models import modeling_hymeta as mod
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig

    cfg = HymetaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=512, shared_intermediate_size=256, num_local_experts=4, num_meta_tokens=32,
                       sliding_window=48, max_position_embeddings=1024)
    ckpt = ckpt_70b(gen(1), cfg)
    whole = mod.HymetaForCausalLM(cfg, max_num_seqs=3, device="cpu")
    whole.load_weights(ckpt.items())
    loaded_all = set()
    for rank in range(2):
        model = mod.HymetaForCausalLM(cfg, max_num_seqs=3, device="cpu", pp_rank=rank, pp_size=2)
        loaded = model.load_weights(ckpt.items())
        loaded_all |= loaded
        m = model.model
        assert (m.start_layer, m.end_layer) == ((0, 2), (2, 4))[rank] and len(m.layers) == 2 and len(m.attn_layers) == 2
        assert m.cache_shape[0] == 2 and m.meta_fattn_cache_shape[0] == 2
        assert [a.layer_name.split("model.")[-1] for a in m.attn_layers] == [f"layers.{i}.attn.vanilla_attn.attn" for i in
                                                                             range(m.start_layer, m.end_layer)]
        assert (m.embeddings is not None) == (rank == 0) and (m.meta_tokens is not None) == (rank == 0)
        assert (m.norm is not None) == (rank == 1) and (model.lm_head is not None) == (rank == 1)
        assert all(not (f"layers.{i}." in name) for name in loaded for i in range(4) if not m.start_layer <= i < m.end_layer)
        for j, layer in enumerate(m.layers):      # the stage's layer j is the whole model's layer start + j
            ref = whole.model.layers[m.start_layer + j]
            assert torch.equal(layer.attn.out_proj.weight, ref.attn.out_proj.weight)
            assert type(layer.mlp) is type(ref.mlp)
        assert torch.equal(m.lower_bounds, whole.model.lower_bounds)
        assert callable(model.make_empty_intermediate_tensors)
        empty = model.make_empty_intermediate_tensors(5, torch.bfloat16, "cpu")
        assert empty["hidden_states"].shape == (5, 512) and empty["residual"].shape == (5, 512)
    assert "lm_head.weight" in loaded_all and "model.embeddings.weight" in loaded_all


def test_7b_checkpoint_loads_per_pipeline_stage():
    """A stage of the 7B class keeps its own layers (state cache and sliding-window attention layers sized to them,
    registered under their global names), the first one the embedding, the last one the final norm and the lm_head."""
    from vllm_hymeta.model_for_7B import modeling_gla_swa as mod
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig

    cfg = GLAswaConfig(vocab_size=512, hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                       intermediate_size=1024, sliding_window=48, max_position_embeddings=1024)
    ckpt = ckpt_7b(gen(0), cfg)
    whole = mod.GLAswaForCausalLM(cfg, max_num_seqs=3, device="cpu")
    whole.load_weights(ckpt.items())
    loaded_all = set()
    for rank, (lo, hi) in enumerate(((0, 1), (1, 3), (3, 4))):                  # vLLM's partition of 4 layers over 3 stages
        model = mod.GLAswaForCausalLM(cfg, max_num_seqs=3, device="cpu", pp_rank=rank, pp_size=3)
        loaded = model.load_weights(ckpt.items())
        loaded_all |= loaded
        m = model.model
        assert (m.start_layer, m.end_layer) == (lo, hi) and len(m.layers) == hi - lo
        assert list(m.attn_layers.keys()) == [str(i) for i in range(lo, hi) if i in cfg.attn_layers]
        assert all(a.layer_name.endswith(f"model.layers.{i}.attn.attn") for i, a in m.attn_layers.items())
        assert m.cache_shape[0] == sum(1 for i in range(lo, hi) if i not in cfg.attn_layers)
        assert len(model.kv_cache) == len(m.attn_layers)
        assert (m.embeddings is not None) == (rank == 0)
        assert (m.norm is not None) == (rank == 2) and (model.lm_head is not None) == (rank == 2)
        assert all(not (f"layers.{i}." in name) for name in loaded for i in range(4) if not lo <= i < hi)
        for j, layer in enumerate(m.layers):
            assert torch.equal(layer.attn.o_proj.weight, whole.model.layers[lo + j].attn.o_proj.weight)
        empty = model.make_empty_intermediate_tensors(5, torch.bfloat16, "cpu")
        assert empty["hidden_states"].shape == (5, 512) and empty["residual"].shape == (5, 512)
    assert loaded_all == whole.load_weights(ckpt.items())


def test_whole_model_restatements_run():
    from oracle.model import glaswa_model_ref, hymeta_model_ref
    from vllm_hymeta.model_for_7B.configuration_gla_swa import GLAswaConfig
    from vllm_hymeta.models.configuration_hymeta import HymetaConfig

    g_ = gen(2)
    cfg7 = GLAswaConfig(vocab_size=64, hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                        intermediate_size=256, sliding_window=8, max_position_embeddings=64)
    ids = torch.randint(0, 64, (20,), generator=g_)
    ck = ckpt_7b(g_, cfg7)
    h, logits = glaswa_model_ref(ids, ck, cfg7)
    assert h.shape == (20, 256) and logits.shape == (20, 64) and torch.isfinite(logits).all()
    # causal: a prefix of the sequence gives the prefix of the result
    h2, _ = glaswa_model_ref(ids[:11], ck, cfg7)
    assert torch.equal(h2, h[:11])

    cfg70 = HymetaConfig(vocab_size=64, hidden_size=256, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
                         intermediate_size=256, shared_intermediate_size=128, num_local_experts=2, num_meta_tokens=8,
                         sliding_window=8, max_position_embeddings=64)
    ck = ckpt_70b(g_, cfg70)
    h, logits = hymeta_model_ref(ids, ck, cfg70)
    assert h.shape == (20, 256) and logits.shape == (20, 64) and torch.isfinite(logits).all()
    h2, _ = hymeta_model_ref(ids[:11], ck, cfg70)
    assert torch.equal(h2, h[:11])
