"""7B flavour of the GLA decode step (reference:
``vllm_hymeta/model_for_7B/my_fused_recurrent.py:123-153``): same kernel, bf16 gate accepted,
result shaped ``[B, H, D]`` (``:117-119``)."""

from __future__ import annotations

from typing import Optional

import torch

from .. import hip_ops
from ..models.my_fused_recurrent import _as_bhd


def my_fused_recurrent(q, k, v, g=None, kv_caches=None, slot_idx=None, scale=None,
                       initial_state=None, output_final_state=False, reverse=False) -> torch.Tensor:
    return hip_ops.gla_decode(_as_bhd(q), _as_bhd(k), _as_bhd(v), _as_bhd(g), kv_caches,
                              slot_idx.to(torch.int32), scale=scale)


def my_fused_recurrent_gla(q, k, v, g=None, kv_caches=None, slot_idx=None, scale: Optional[float] = None,
                           initial_state=None, output_final_state=False, reverse=False) -> torch.Tensor:
    if scale is None:
        scale = q.shape[-1] ** -0.5
    return my_fused_recurrent(q, k, v, g, kv_caches, slot_idx, scale, initial_state, output_final_state, reverse)
