"""The output projection to the vocabulary.

Reference: ``lm_head.float()`` applied to ``hidden_states.float()`` in ``compute_logits``
(``vllm_hymeta/model_for_7B/modeling_gla_swa.py:548,579-583``, ``vllm_hymeta/models/modeling_hymeta.py:1429,1462-1466``).
A bf16 checkpoint tensor widened to fp32 holds the same values, so it is kept in bf16 (half the bytes a decode step
has to read) and multiplied with fp32 accumulation and an fp32 result; any other dtype is widened once, as the
reference does.
"""

from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops


class LmHeadMixin:
    """``lm_head`` reads as the reference's fp32 ``[vocab, hidden]`` tensor; ``_lm_head`` is what is stored."""

    _lm_head: Optional[torch.Tensor] = None

    @property
    def lm_head(self) -> Optional[torch.Tensor]:
        return None if self._lm_head is None else self._lm_head.float()

    @lm_head.setter
    def lm_head(self, weight: Optional[torch.Tensor]) -> None:
        self._lm_head = None if weight is None else (weight if weight.dtype == torch.bfloat16 else weight.float())


def lm_head_logits(hidden_states: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """fp32 logits ``[rows, vocab]`` of bf16 (or any) hidden states against the stored lm_head."""
    if weight.dtype == torch.bfloat16 and hip_ops.logits_supported(hidden_states, weight):
        return hip_ops.bf16_logits(hidden_states, weight)
    return torch.nn.functional.linear(hidden_states.float(), weight.float())
