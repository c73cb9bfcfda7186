"""HF config of the 7B "gla_swa" family (``model_type = "gla_swa"``).

Field names and defaults follow the reference
(``vllm_hymeta/model_for_7B/configuration_gla_swa.py:15-68``): odd layers are
sliding-window softmax attention (``attn_layers``), even layers GLA; ``attn_type_list``
marks which layers own a paged KV cache.
"""

from __future__ import annotations

from transformers.configuration_utils import PretrainedConfig

_DEFAULTS = dict(
    vocab_size=152064, hidden_size=3584, num_hidden_layers=28, attn_mode="chunk",
    num_attention_heads=28, num_key_value_heads=4, use_short_conv=False, conv_size=4,
    gate_logit_normalizer=16, gate_low_rank_dim=16, intermediate_size=18944, hidden_act="swish",
    max_position_embeddings=131072, sliding_window=4096, elementwise_affine=True, norm_eps=1e-6,
    rope_theta=1000000.0, attention_dropout=0.0, use_cache=True, initializer_range=0.02,
    fuse_cross_entropy=True,
)


class GLAswaConfig(PretrainedConfig):
    model_type = "gla_swa"
    keys_to_ignore_at_inference = ["past_key_values"]

    def __init__(self, pad_token_id=None, bos_token_id=151643, eos_token_id=151643,
                 tie_word_embeddings=False, **kwargs):
        for name, default in _DEFAULTS.items():
            setattr(self, name, kwargs.pop(name, default))
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        self.head_dim = self.hidden_size // self.num_attention_heads
        self.attn_layers = list(range(1, self.num_hidden_layers, 2))
        self.attn_type_list = [int(i in self.attn_layers) for i in range(self.num_hidden_layers)]
        super().__init__(pad_token_id=pad_token_id, bos_token_id=bos_token_id,
                         eos_token_id=eos_token_id, tie_word_embeddings=tie_word_embeddings, **kwargs)
