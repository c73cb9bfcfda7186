"""HF config of the 70B "hybrid" family (``model_type = "hybrid"``).

Field names and defaults follow the reference's vLLM-side config
(``vllm_hymeta/models/configuration_hymeta.py:15-92``), because checkpoints and
vLLM's hybrid-model bookkeeping read them by name: ``attn_type_list`` (all 1: every
layer owns a softmax KV cache), ``interleaved_sliding_window`` (0 marks the
full-attention layers; the backend turns ``0 - 1`` into "unbounded"),
``num_layer_experts`` (1 on the dense-MLP layers).
"""

from __future__ import annotations

from transformers.configuration_utils import PretrainedConfig

_FULL_ATTN_LAYERS = (3, 10, 17, 24)
_DENSE_MLP_LAYERS = (0, 1, 2, 4, 6, 8, 10)

_DEFAULTS = dict(
    vocab_size=152064, hidden_size=3584, num_hidden_layers=28, attn_mode="chunk",
    num_attention_heads=28, num_key_value_heads=4, use_short_conv=False, conv_size=4,
    use_lower_bound=True, num_meta_tokens=128, intermediate_size=18944, hidden_act="swish",
    max_position_embeddings=131072, sliding_window=4096, elementwise_affine=True, norm_eps=1e-6,
    rope_theta=1000000.0, attention_dropout=0.0, use_cache=True, initializer_range=0.02,
    fuse_cross_entropy=True, num_local_experts=16, num_experts_per_topk=1,
    router_jitter_noise=0.0, shared_intermediate_size=18944,
)


class HymetaConfig(PretrainedConfig):
    model_type = "hybrid"
    keys_to_ignore_at_inference = ["past_key_values"]

    def __init__(self, pad_token_id=None, bos_token_id=151643, eos_token_id=151643,
                 tie_word_embeddings=False, **kwargs):
        for name, default in _DEFAULTS.items():
            setattr(self, name, kwargs.pop(name, default))
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        self.head_dim = self.hidden_size // self.num_attention_heads
        n = self.num_hidden_layers
        self.full_attn_layers = list(_FULL_ATTN_LAYERS)
        self.dense_mlp_layers = list(_DENSE_MLP_LAYERS)
        self.interleaved_sliding_window = [
            0 if i in _FULL_ATTN_LAYERS else self.sliding_window for i in range(n)]
        self.attn_type_list = [1] * n
        self.num_layer_experts = [
            1 if i in _DENSE_MLP_LAYERS else self.num_local_experts for i in range(n)]
        if "quantization_config" in kwargs:
            self.quantization_config = dict(kwargs["quantization_config"])
        super().__init__(pad_token_id=pad_token_id, bos_token_id=bos_token_id,
                         eos_token_id=eos_token_id, tie_word_embeddings=tie_word_embeddings, **kwargs)
