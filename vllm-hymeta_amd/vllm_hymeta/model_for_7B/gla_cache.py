"""Recurrent-state cache of the 7B model (reference: ``vllm_hymeta/model_for_7B/gla_cache.py:9-33``):
one tensor ``[L_gla, slots, H/tp, D, D]`` in the model dtype."""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from ..state_cache import ConstantSizeStateCache


@dataclass
class GLACacheParams:
    gla_cache: torch.Tensor = field(default_factory=torch.Tensor)
    state_indices_tensor: torch.Tensor = field(default_factory=torch.Tensor)

    def at_layer_idx(self, layer_idx: int) -> "GLACacheParams":
        return GLACacheParams(self.gla_cache[layer_idx], self.state_indices_tensor)


class GLACacheManager(ConstantSizeStateCache):
    def __init__(self, dtype, cache_shape, device="cuda"):
        super().__init__(cache_shape[1], device)
        self._gla_cache = torch.zeros(cache_shape, dtype=dtype, device=self.device)

    @property
    def cache(self):
        return (self._gla_cache,)

    def current_run_tensors(self, **kwargs):
        cache_tensors, state_indices_tensor = super().current_run_tensors(**kwargs)
        return cache_tensors[0], state_indices_tensor
