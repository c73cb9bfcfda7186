"""Decoder block around the attention layers: pre-norm, attention, norm with the residual add
folded in, feed-forward, residual.

Reference: ``HybridBlock.forward`` (``vllm_hymeta/models/modeling_hymeta.py:1009-1075``) and the 7B
``GLAswaBlock.forward`` (``vllm_hymeta/model_for_7B/modeling_gla_swa.py:283-306``) -- the same
sequence in both models; vLLM's ``RMSNorm`` (``:949,957`` / ``:241,271``) is ``HymetaRMSNorm`` here.
"""

from __future__ import annotations

from typing import Optional, Tuple, Union

import torch

from .. import hip_ops


class HymetaRMSNorm(torch.nn.Module):
    """vLLM ``RMSNorm`` call convention: ``norm(x)`` returns the normed tensor, ``norm(x, residual)``
    returns ``(normed(x + residual), x + residual)`` and updates ``residual`` in place."""

    def __init__(self, hidden_size: int, eps: float = 1e-6, weight: Optional[torch.Tensor] = None,
                 device=None):
        super().__init__()
        if weight is None:
            weight = torch.ones(hidden_size, dtype=torch.bfloat16, device=device)
        if weight.numel() != hidden_size:
            raise ValueError(f"weight has {weight.numel()} elements, hidden_size is {hidden_size}")
        self.register_buffer("weight", weight.to(torch.bfloat16).contiguous())
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, int8: bool = False
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        """``int8``: the consumer is a W8A8 linear: the launch also quantises its output (``hip_ops.add_rmsnorm``)."""
        if isinstance(x, hip_ops.PartialSums):      # a K-sliced linear's result, summed while it is read
            if residual is None:
                x = x.tensor()
            else:
                r2 = residual.reshape(-1, residual.shape[-1])
                if r2.data_ptr() != residual.data_ptr():
                    raise ValueError("residual must be viewable as [tokens, hidden] without a copy (it is updated in place)")
                return hip_ops.add_rmsnorm(x, self.weight, self.variance_epsilon, residual=r2, int8=int8), residual
        x2 = x.reshape(-1, x.shape[-1])
        keep = (lambda t: t) if int8 and x.dim() == 2 else (lambda t: t.view_as(x))      # a view would drop the int8 rows
        if residual is None:
            return keep(hip_ops.add_rmsnorm(x2, self.weight, self.variance_epsilon, int8=int8))
        r2 = residual.reshape(-1, residual.shape[-1])
        if r2.data_ptr() != residual.data_ptr():
            raise ValueError("residual must be viewable as [tokens, hidden] without a copy (it is updated in place)")
        out = hip_ops.add_rmsnorm(x2, self.weight, self.variance_epsilon, residual=r2, int8=int8)
        return keep(out), residual


def _first_linears_are_w8a8(module: torch.nn.Module, names) -> bool:
    """Whether every linear called ``names`` directly in ``module`` or in one of its children (the two branches of the 70B
    attention) is W8A8, and there is at least one: they all read the normed input."""
    found = [getattr(m, n) for m in [module] + list(module.children()) for n in names if hasattr(m, n)]
    return bool(found) and all(getattr(lin, "kind", None) == "w8a8" for lin in found)


class DecoderBlock(torch.nn.Module):
    """``attn`` is any of the attention layers of this package (its keyword arguments pass through),
    ``mlp`` a ``HymetaMLP`` or ``HymetaMoE``.  Returns ``(hidden_states, None)`` as the reference's
    70B block does; the 7B block's second value (its residual) is never read by its caller
    (``model_for_7B/modeling_gla_swa.py:479-485``)."""

    def __init__(self, attn_norm: HymetaRMSNorm, attn: torch.nn.Module, mlp_norm: HymetaRMSNorm, mlp: torch.nn.Module):
        super().__init__()
        self.attn_norm, self.attn, self.mlp_norm, self.mlp = attn_norm, attn, mlp_norm, mlp
        # a W8A8 projection right behind a norm: the norm quantises its own output (one launch less per linear)
        self.attn_int8 = _first_linears_are_w8a8(attn, ("qkv_proj",))
        self.mlp_int8 = _first_linears_are_w8a8(mlp, ("gate_up_proj",))

    def forward(self, hidden_states: torch.Tensor, attn_fn=None, carry_residual: bool = False,
                residual: Optional[torch.Tensor] = None, **attn_kwargs) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """``attn_fn``: call this on the normed input instead of ``self.attn(normed, **attn_kwargs)``
        (the 70B model's one-off pass of the meta tokens uses the layer's ``prepare_meta_state``).

        ``carry_residual``: return ``(mlp output, residual stream)`` instead of their sum, and take the pair
        of the block before (``residual`` None for the first block).  The sum then happens inside the next
        block's fused add + norm (or the model's final norm) instead of in two launches of its own per
        block; the values are the same (the sum is rounded to bf16 either way)."""
        if carry_residual and residual is not None:
            normed, residual = self.attn_norm(hidden_states, residual, int8=self.attn_int8)    # residual += hidden_states, in place
        else:
            # the block's input becomes the residual stream; it is updated in place by mlp_norm, so
            # take a private copy when the caller keeps using its tensor
            residual = hidden_states.clone()
            normed = self.attn_norm(hidden_states, int8=self.attn_int8)
        # the projections that end the attention and the MLP feed a fused add + norm: they may leave their K slices'
        # sum to it (modules that take the flag; a tensor-parallel one ignores it, its all-reduce needs the sum)
        attn_out = attn_fn(normed) if attn_fn is not None else (
            self.attn(normed, defer_out=True, **attn_kwargs) if getattr(self.attn, "can_defer_out", False)
            else self.attn(normed, **attn_kwargs))
        normed, residual = self.mlp_norm(attn_out, residual, int8=self.mlp_int8)
        if carry_residual:
            if getattr(self.mlp, "can_defer_out", False):
                return self.mlp(normed, defer_out=True), residual
            return self.mlp(normed), residual
        return residual + self.mlp(normed), None


HybridBlock = DecoderBlock      # vllm_hymeta/models/modeling_hymeta.py:935
GLAswaBlock = DecoderBlock      # vllm_hymeta/model_for_7B/modeling_gla_swa.py:221
