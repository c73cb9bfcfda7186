"""``my_fused_recurrent_gla`` (reference: ``vllm_hymeta/models/my_fused_recurrent_gla.py:8-23``)."""

from __future__ import annotations

from typing import Optional

import torch

from .my_fused_recurrent import my_fused_recurrent


def my_fused_recurrent_gla(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    g: Optional[torch.Tensor] = None,
    kv_caches: Optional[torch.Tensor] = None,
    slot_idx: Optional[torch.Tensor] = None,
    scale: Optional[float] = None,
    initial_state=None,
    output_final_state: bool = False,
    reverse: bool = False,
) -> torch.Tensor:
    if scale is None:
        scale = q.shape[-1] ** -0.5
    return my_fused_recurrent(q, k, v, g, kv_caches, slot_idx, scale, initial_state, output_final_state, reverse)
