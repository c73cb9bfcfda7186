"""Recurrent-state cache of the 70B model (reference: ``vllm_hymeta/models/hymeta_cache.py:9-62``).

Three tensors, all in the model dtype (bf16):
  * ``hymeta_cache``      ``[L, slots, H/tp, D, D]``  per-sequence GLA state;
  * ``meta_linear_cache`` ``[L, H/tp, D, D]``          GLA state after the 128 meta tokens (shared);
  * ``meta_fattn_cache``  ``[L, 2, M, Hkv/tp, D]``     roped K/V of the meta tokens (shared).
Shapes are built by the model (reference: ``modeling_hymeta.py:1137-1154``).
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from ..state_cache import ConstantSizeStateCache


@dataclass
class HymetaCacheParams:
    hymeta_cache: torch.Tensor = field(default_factory=torch.Tensor)
    meta_linear_cache: torch.Tensor = field(default_factory=torch.Tensor)
    meta_fattn_cache: torch.Tensor = field(default_factory=torch.Tensor)
    # slot of every sequence of the current step, batch order (prefills first, then decodes)
    state_indices_tensor: torch.Tensor = field(default_factory=torch.Tensor)

    def at_layer_idx(self, layer_idx: int) -> "HymetaCacheParams":
        return HymetaCacheParams(self.hymeta_cache[layer_idx], self.meta_linear_cache[layer_idx],
                                 self.meta_fattn_cache[layer_idx], self.state_indices_tensor)


class HymetaCacheManager(ConstantSizeStateCache):
    def __init__(self, dtype, cache_shape, meta_linear_cache_shape, meta_fattn_cache_shape, device="cuda"):
        super().__init__(cache_shape[1], device)
        # The per-sequence state is zeroed (or seeded from the meta state) by the prefill kernel on
        # a sequence's first chunk, so it needs no initial value; zeros keep NaNs out of debug dumps.
        self._hymeta_cache = (
            torch.zeros(cache_shape, dtype=dtype, device=self.device),
            torch.zeros(meta_linear_cache_shape, dtype=dtype, device=self.device),
            torch.zeros(meta_fattn_cache_shape, dtype=dtype, device=self.device),
        )

    @property
    def cache(self):
        return self._hymeta_cache

    def _slot_tensors(self):
        return (self._hymeta_cache[0],)   # only the per-sequence state has a slot axis

    def current_run_tensors(self, **kwargs):
        cache_tensors, state_indices_tensor = super().current_run_tensors(**kwargs)
        return cache_tensors[0], cache_tensors[1], cache_tensors[2], state_indices_tensor
