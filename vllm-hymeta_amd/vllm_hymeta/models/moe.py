"""Feed-forward blocks of the 70B model on the HIP kernels: the dense gated MLP and the sparse
mixture of experts with its shared expert.

Reference: ``HymetaMLP`` (``vllm_hymeta/models/modeling_hymeta.py:236-271``), ``HymetaMoE``
(``:273-364``, vLLM ``FusedMoE(top_k, renormalize=False, reduce_results=True)``) and the
shared-expert sum in ``HybridBlock.forward`` (``:1044-1058``); eager: ``hf_model/moe.py:34-101``.

What changed in mechanism: routing, the counting sort by expert, both expert GEMMs, the
activation and the combine (with the shared expert's output folded in) run as six stream-ordered
launches with no host read of the expert counts; the ``copy.deepcopy`` of the activations
(``:1048``) is gone; under expert parallelism a rank computes only its experts' pairs and the one
all-reduce of the layer sums the ranks (what vLLM's EP does at DP = 1).
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import hip_ops
from ..linear import HymetaLinear
from .. import parallel
from ..parallel import all_reduce_sum


class HymetaMLP(torch.nn.Module):
    """``down(silu(gate(x)) * up(x))`` with the gate and up projections fused (``gate_up_proj`` rows =
    gate then up)."""

    def __init__(self, gate_up_proj: HymetaLinear, down_proj: HymetaLinear, tp_group=None, reduce_results: bool = True):
        super().__init__()
        self.gate_up_proj, self.down_proj = gate_up_proj, down_proj
        self.tp_group, self.reduce_results = tp_group, reduce_results

    can_defer_out = True

    def forward(self, x: torch.Tensor, defer_out: bool = False):
        """``defer_out``: the result goes straight into a fused add + norm, which may sum ``down_proj``'s K slices itself
        (see ``HymetaLinear.forward``); when an all-reduce follows, the group's one-shot exchange does."""
        reduce = self.reduce_results and parallel._active(self.tp_group)
        # the K slices' sum is left to whoever reads the result next: the fused add + norm (no all-reduce), or the
        # group's one-shot exchange
        defer = parallel.defers_into_all_reduce(self.tp_group) if reduce else defer_out
        y = self.down_proj(self.gate_up_proj.forward_silu_mul(x), defer=defer)
        return all_reduce_sum(y, self.tp_group) if reduce else y


def expert_range(num_experts: int, ep_size: int, ep_rank: int) -> Tuple[int, int]:
    """Experts owned by one expert-parallel rank: contiguous, equal shares (``num_experts % ep_size == 0``)."""
    if num_experts % ep_size != 0:
        raise ValueError(f"num_experts {num_experts} must be divisible by the expert-parallel size {ep_size}")
    per = num_experts // ep_size
    return ep_rank * per, (ep_rank + 1) * per


class HymetaMoE(torch.nn.Module):
    """Top-k routed experts (+ optional shared expert).  ``w13 [E, 2I, H]`` / ``w2 [E, H, I]`` hold every expert, or,
    for a rank that owns experts ``local_experts = [lo, hi)`` under expert parallelism, those experts only
    (``[hi - lo, 2I, H]`` / ``[hi - lo, H, I]``).  Every rank routes every token; a rank computes the pairs that went
    to its experts (rows of the others stay zero, the shared expert's partial sums are added on every rank) and the
    all-reduce over ``group`` sums the ranks: expert parallelism without data parallelism, as vLLM runs it for the
    reference's serving commands (``readme.md:115,167``).  An all-to-all dispatch would only pay with data parallelism
    (ranks holding different tokens), which the reference never runs."""

    def __init__(self, gate_weight: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, top_k: int = 1,
                 renormalize: bool = False, shared_mlp: Optional[HymetaMLP] = None,
                 local_experts: Optional[Tuple[int, int]] = None, group=None,
                 w13_scale: Optional[torch.Tensor] = None, w2_scale: Optional[torch.Tensor] = None,
                 gate_scale: Optional[torch.Tensor] = None):
        """W8A8 experts (a compressed-tensors W8A8 checkpoint, ``hf_model/quant_w8a8.py:121-132``, quantises every Linear
        but the lm_head: experts and router included): ``w13`` / ``w2`` int8 with ``w13_scale [E, 2I]`` / ``w2_scale [E, H]``;
        an int8 ``gate_weight`` comes with ``gate_scale [E]`` and the router logits are then that W8A8 linear's output.
        GPTQ experts (``hf_model/hymeta_gptq.py:30-32``): ``w13`` / ``w2`` as ``hip_ops.GptqExperts`` stacks."""
        super().__init__()
        self.w8a8_gate = None
        if gate_weight.dtype == torch.int8:
            # the W8A8 linear kernels take whole 64-column blocks: the E <= 64 router rows are padded with zero rows
            E, H = gate_weight.shape
            pad = torch.zeros(64, H, dtype=torch.int8, device=gate_weight.device)
            pad[:E] = gate_weight
            sc = torch.ones(64, dtype=torch.float32, device=gate_weight.device)
            sc[:E] = gate_scale.reshape(-1).float()
            self.w8a8_gate = HymetaLinear.from_w8a8(pad, sc)
        self.register_buffer("gate_weight", gate_weight)   # [E, H], replicated
        self._gptq = None
        if isinstance(w13, hip_ops.GptqExperts):
            # the stacks' tensors are the buffers; ``w13`` / ``w2`` (properties below) put them together again
            self._gptq = {"w13": w13[3:], "w2": w2[3:]}
            for name, w in (("w13", w13), ("w2", w2)):
                for leaf, t in zip(("qweight", "qzeros", "scales"), w[:3]):
                    self.register_buffer(f"{name}_{leaf}", t)
        else:
            self.register_buffer("_w13", w13)
            self.register_buffer("_w2", w2)
        self.register_buffer("w13_scale", w13_scale)
        self.register_buffer("w2_scale", w2_scale)
        self.register_buffer("gate_scale", None if gate_scale is None else gate_scale.reshape(-1).float().contiguous())
        self.top_k, self.renormalize = top_k, renormalize
        self.shared_mlp = shared_mlp
        self.local_experts = local_experts
        self.group = group

    def _stack(self, name: str):
        if self._gptq is None:
            return getattr(self, "_" + name)
        return hip_ops.GptqExperts(getattr(self, name + "_qweight"), getattr(self, name + "_qzeros"), getattr(self, name + "_scales"),
                                   *self._gptq[name])

    @property
    def w13(self):
        """bf16 | int8 ``[E, 2I, H]``, or the ``GptqExperts`` stack standing for it."""
        return self._stack("w13")

    @property
    def w2(self):
        return self._stack("w2")

    def router_logits(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """The router's logits where the block cannot compute them itself (an int8 router above the batch size of the
        one-launch router: the padded W8A8 linear); ``None``: run inside ``moe_forward``."""
        if self.w8a8_gate is None or hip_ops.moe_router_fused(x, self.gate_weight):
            return None
        return self.w8a8_gate(x)[:, :self.gate_weight.shape[0]]

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        # (a 2-D input is taken as it is: a norm that quantised its output for the W8A8 consumers below attached the int8
        # rows to that very tensor object)
        x = hidden_states if hidden_states.dim() == 2 else hidden_states.reshape(-1, hidden_states.shape[-1])
        shared = None
        if self.shared_mlp is not None:
            # the shared expert is tensor-parallel like a dense MLP; its partial sums ride in the
            # same all-reduce as the experts' when it is built with reduce_results=False
            shared = self.shared_mlp(x)
        # the router (a replicated linear layer, reference modeling_hymeta.py:312-324) runs inside the block: in the
        # routing launch itself for decode-sized batches
        logits = self.router_logits(x)
        out = hip_ops.moe_forward(x, logits, self.w13, self.w2, top_k=self.top_k, renormalize=self.renormalize,
                                  shared=shared, local_experts=self.local_experts,
                                  gate_weight=self.gate_weight if logits is None else None, gate_scale=self.gate_scale,
                                  w13_scale=self.w13_scale, w2_scale=self.w2_scale)
        return all_reduce_sum(out, self.group).view_as(hidden_states)
