"""GLA single-token decode step, reference name and argument order
(``vllm_hymeta/models/my_fused_recurrent.py:123-135``), on the HIP kernel."""

from __future__ import annotations

from typing import Optional

import torch

from .. import hip_ops


def _as_bhd(x: torch.Tensor) -> torch.Tensor:
    """[B, H, 1, D] (the reference layout) or [B, H, D] -> a [B, H, D] view, no copy."""
    if x.dim() == 4:
        assert x.shape[2] == 1, "decode step: sequence length must be 1"
        return x[:, :, 0, :]
    return x


def my_fused_recurrent(
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
    """q, k, v, g ``[B, H, 1, D]``; ``kv_caches [slots, H, D, D]`` updated in place at
    ``slot_idx`` (-1 skips).  Returns ``[B, H*D]`` (reference: ``my_fused_recurrent.py:118-119``).
    ``initial_state`` / ``output_final_state`` / ``reverse`` are accepted and ignored, as the
    reference ignores them."""
    B, H, D = q.shape[0], q.shape[1], q.shape[-1]
    assert k.shape[:2] == (B, k.shape[1]) and v.shape[-1] == D
    out = hip_ops.gla_decode(_as_bhd(q), _as_bhd(k), _as_bhd(v), _as_bhd(g), kv_caches,
                             slot_idx.to(torch.int32), scale=scale)
    return out.view(B, H * D)
