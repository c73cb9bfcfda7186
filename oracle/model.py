"""CPU restatement of the 7B "gla_swa" and 70B "hybrid" models' forward pass on ONE sequence.  Test infrastructure
only: nothing in the product path imports this.  Parity unpinned (the reference holds no golden
outputs for the whole model); it composes the per-layer restatements of ``oracle/layers.py``.

Follows ``GLAswaModel.forward`` / ``GLAswaForCausalLM.compute_logits``
(``vllm_hymeta/model_for_7B/modeling_gla_swa.py:416-470, 520-528``): embedding lookup, decoder
blocks (attention kind by ``config.attn_layers``), final RMS norm, fp32 LM head; and
``HymetaModel.forward`` (``vllm_hymeta/models/modeling_hymeta.py:1233-1375``) in the form it takes
on a fresh server: the meta tokens in front of the sequence, their rows dropped after the final norm.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .layers import (_linear, decoder_block_ref, gla_layer_7b_ref, hybrid_layer_70b_ref, lower_bounds_ref, rms_norm_ref,
                     swa_layer_7b_ref)
from .moe import mlp_ref, moe_ref


def glaswa_model_ref(input_ids: torch.Tensor, weights: Dict[str, torch.Tensor], cfg) -> Tuple[torch.Tensor, torch.Tensor]:
    """``weights``: bf16 tensors under their checkpoint names (``model.embeddings.weight``,
    ``model.layers.<i>.attn.q_proj.weight`` ... ``lm_head.weight``).  Returns
    ``(hidden [T, hidden] bf16 after the final norm, logits [T, vocab] fp32)``."""
    T = input_ids.shape[0]
    positions = torch.arange(T)
    h = weights["model.embeddings.weight"][input_ids]
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        p = {k[len(pre) + 5:]: v for k, v in weights.items() if k.startswith(pre + "attn.")}
        if i in cfg.attn_layers:
            attn = lambda x, p=p: swa_layer_7b_ref(x, positions, p, cfg.num_attention_heads, cfg.num_key_value_heads,
                                                   cfg.sliding_window, cfg.rope_theta, cfg.max_position_embeddings)
        else:
            attn = lambda x, p=p: gla_layer_7b_ref(x, p, cfg.num_attention_heads, cfg.num_key_value_heads,
                                                   norm_eps=cfg.norm_eps)[0]
        w13 = torch.cat([weights[pre + "mlp.gate_proj.weight"], weights[pre + "mlp.up_proj.weight"]])
        mlp = lambda x, w13=w13, w2=weights[pre + "mlp.down_proj.weight"]: mlp_ref(x, w13, w2)
        h = decoder_block_ref(h, attn, mlp, weights[pre + "attn_norm.weight"], weights[pre + "mlp_norm.weight"], cfg.norm_eps)
    h = rms_norm_ref(h, weights["model.norm.weight"], cfg.norm_eps)
    return h, h.float() @ weights["lm_head.weight"].float().t()


def hymeta_model_ref(input_ids: torch.Tensor, weights: Dict[str, torch.Tensor], cfg) -> Tuple[torch.Tensor, torch.Tensor]:
    """70B family.  ``weights`` under checkpoint names (``model.layers.<i>.attn.vanilla_attn.q_proj.weight``,
    ``...block_sparse_moe.experts.<e>.gate_proj.weight``, ``model.meta_tokens``, ``model.lower_bounds`` ...).
    The hidden states carry the M meta rows in front through every block; window, experts and shared
    expert per layer come from the config lists.  Returns ``(hidden [T, hidden], logits [T, vocab])``
    of the real tokens."""
    T, M = input_ids.shape[0], cfg.num_meta_tokens
    positions = torch.arange(T)
    lbs = lower_bounds_ref(weights["model.lower_bounds"]) if cfg.use_lower_bound else None
    h = torch.cat([weights["model.meta_tokens"].to(torch.bfloat16), weights["model.embeddings.weight"][input_ids]])
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    for i in range(cfg.num_hidden_layers):
        pre = f"model.layers.{i}."
        p = {k[len(pre) + 5:]: v for k, v in weights.items() if k.startswith(pre + "attn.")}
        window = cfg.interleaved_sliding_window[i % len(cfg.interleaved_sliding_window)]
        lb = lbs[i % 7].view(H, -1) if lbs is not None else None

        def attn(x, p=p, window=window, lb=lb, i=i):
            return hybrid_layer_70b_ref(x[M:], x[:M], positions, p, H, Hkv, window or None, lb, first_layer=(i % 7 == 0),
                                        rope_theta=cfg.rope_theta, max_position=cfg.max_position_embeddings,
                                        norm_eps=cfg.norm_eps, return_meta=True)

        def dense(x, name):
            w13 = torch.cat([weights[pre + name + ".gate_proj.weight"], weights[pre + name + ".up_proj.weight"]])
            return mlp_ref(x, w13, weights[pre + name + ".down_proj.weight"])

        n_exp = cfg.num_layer_experts[i] if isinstance(cfg.num_layer_experts, list) else cfg.num_layer_experts
        if n_exp == 1:
            ffn = lambda x, dense=dense: dense(x, "mlp")
        else:
            def ffn(x, pre=pre, n_exp=n_exp, dense=dense):
                ex = pre + "block_sparse_moe.experts."
                w13 = torch.stack([torch.cat([weights[f"{ex}{e}.gate_proj.weight"], weights[f"{ex}{e}.up_proj.weight"]])
                                   for e in range(n_exp)])
                w2 = torch.stack([weights[f"{ex}{e}.down_proj.weight"] for e in range(n_exp)])
                logits = _linear(x, weights[pre + "block_sparse_moe.gate.weight"])
                shared = dense(x, "shared_mlp") if (pre + "shared_mlp.gate_proj.weight") in weights else None
                return moe_ref(x, logits, w13, w2, cfg.num_experts_per_topk, False, shared)
        h = decoder_block_ref(h, attn, ffn, weights[pre + "attn_norm.weight"], weights[pre + "mlp_norm.weight"], cfg.norm_eps)
    final_norm = weights["model.norm.weight"] if "model.norm.weight" in weights else weights["mode.norm.weight"]   # checkpoint typo
    h = rms_norm_ref(h, final_norm, cfg.norm_eps)[M:]
    return h, h.float() @ weights["lm_head.weight"].float().t()
