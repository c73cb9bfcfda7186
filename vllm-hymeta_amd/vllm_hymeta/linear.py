"""Linear layers of the attention blocks: bf16 or one of the fused dequant-GEMMs (GPTQ 4/8-bit, W8A8).
Decode-sized batches (up to 256 rows; GPTQ: up to 64 and 129 to 256) run on this package's weight-streaming kernels; prefill-sized
batches of bf16 / expanded GPTQ / int8 weights go to the library GEMM.  Stands where the reference puts vLLM's
``QKVParallelLinear`` / ``RowParallelLinear`` / ``ReplicatedLinear`` with a ``quant_config``
(``vllm_hymeta/model_for_7B/gla_attention.py:147-176``): a tensor-parallel shard is simply a
``HymetaLinear`` built from the shard's rows (column parallel) or columns (row parallel)."""

from __future__ import annotations

from typing import Optional

import torch

from . import hip_ops


# Rows above which a GPTQ linear expands its weights and calls the library GEMM (measured on the 7B
# projection shapes: the fused kernel wins up to a few hundred rows, the library from there on).
GPTQ_EXPAND_ROWS = 256
# Rows above which a W8A8 linear may put the library int8 GEMM between its own quantisation and
# scaling kernels: the streaming int8 kernels cover up to 256 rows (one pass over the weights); both
# paths give the same bits (exact int32 sums).
W8A8_LIBRARY_ROWS = 256
W8A8_STREAM_ROWS = 128       # the decode-sized kernel: any K of whole 128-byte lines, N of whole 64-column groups


def w8a8_use_library(rows: int, in_features: int, out_features: int) -> bool:
    """Above the streaming kernels' rows a W8A8 linear is either ONE tiled int8-MFMA launch with the scales in its
    epilogue (``hymeta_w8a8_gemm``: K in whole 128-deep steps, N in whole 128-column tiles) or quantise -> library int8
    GEMM -> scale (three launches, the int32 sums through memory); both give the same bits.  The tiled kernel takes the
    shapes where it was measured ahead (7B shapes, us, tiled / library): gate_up 70 / 72 at 129 rows, 68 / 101 at 256,
    331 / 412 at 2048, 4400 / 5568 at 32768 (up to 256 rows the streaming kernel has since taken over: 59 at 256, the
    down projection 48 against 118); qkv 67 / 71 at 2048, 601 / 736 at 32768; o 51 / 57 and 481 / 594; the deep
    narrow down projection (K = 5.3 N: few tiles, long K loops) 587 / 683 at 8192 and level at 32768 -- and leaves the
    rest to the library (qkv / o below 1024 rows, down below 4096: 210 / 189 at 2048)."""
    if rows <= W8A8_STREAM_ROWS:
        return False
    if in_features % 128 != 0 or out_features % 128 != 0:
        return True
    if rows <= W8A8_LIBRARY_ROWS:
        return False
    if out_features >= 16384:
        return False
    if in_features >= 4 * out_features:
        return rows < 4096
    return rows < 1024


class HymetaLinear(torch.nn.Module):
    """y = x W^T + b with W in one of three storage formats."""

    def __init__(self, kind: str = "bf16", **tensors):
        super().__init__()
        assert kind in ("bf16", "gptq", "w8a8")
        self.kind = kind
        self.meta = {k: tensors.pop(k) for k in ("bits", "group_size", "zero_offset") if k in tensors}
        for name, t in tensors.items():
            if t is not None:
                self.register_buffer(name, t, persistent=True)
        self.bias = tensors.get("bias")

    @classmethod
    def from_weight(cls, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> "HymetaLinear":
        """weight bf16 [out, in]."""
        return cls("bf16", weight=weight, bias=bias)

    @classmethod
    def from_gptq(cls, qweight, qzeros, scales, bits: int, group_size: int, bias=None,
                  zero_offset: int = 1) -> "HymetaLinear":
        return cls("gptq", qweight=qweight, qzeros=qzeros, scales=scales, bias=bias, bits=bits,
                   group_size=group_size, zero_offset=zero_offset)

    @classmethod
    def from_w8a8(cls, qweight, w_scale, bias=None) -> "HymetaLinear":
        return cls("w8a8", qweight=qweight, w_scale=w_scale, bias=bias)

    def forward_silu_mul(self, x: torch.Tensor) -> torch.Tensor:
        """``silu(y[:, :I]) * y[:, I:]`` of ``y = self(x)`` for a fused gate/up projection (output rows = gate, then up):
        decode-sized batches of a bf16 or W8A8 projection run it in the GEMM's epilogue (same bits, one launch, no
        ``[M, 2 I]`` intermediate)."""
        if self.bias is None and self.kind in ("bf16", "w8a8"):
            w = self.weight if self.kind == "bf16" else self.qweight
            if hip_ops.gemm_silu_mul_supported(x, w):
                return hip_ops.gemm_silu_mul(x, w, None if self.kind == "bf16" else self.w_scale)
        return hip_ops.silu_and_mul(self(x))

    def forward(self, x: torch.Tensor, defer: bool = False):
        """``defer``: the caller passes the result straight to a fused add + RMS norm (``HymetaRMSNorm`` with a
        residual); a decode-sized bf16 / W8A8 linear whose K is cut into slices then returns their
        ``hip_ops.PartialSums`` and the norm adds them as it reads its input.  Everything else returns the tensor."""
        if self.kind == "bf16":
            # decode-sized batches: the weight-streaming kernel; prefill-sized ones: a plain library GEMM
            if hip_ops.bf16_gemm_supported(x, self.weight):
                return hip_ops.bf16_gemm(x, self.weight, self.bias, defer=defer)
            M, (N, K) = x.shape[0], self.weight.shape
            if (x.dim() == 2 and hip_ops.BF16_GEMM_MAX_ROWS < M <= 2 * hip_ops.BF16_GEMM_MAX_ROWS and K >= 4 * N
                    and hip_ops.bf16_gemm_supported(x[:hip_ops.BF16_GEMM_MAX_ROWS], self.weight)):
                # deep, narrow matrices (the MLP's down projection) at 129 .. 256 rows: two passes of the streaming kernel
                # over the two halves of the rows (the second finds most of the weights in the last-level cache) against
                # a library GEMM that has too few tiles to fill the chip: measured 90 against 126 us at 256 rows
                out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
                half = hip_ops.BF16_GEMM_MAX_ROWS
                hip_ops.bf16_gemm(x[:half], self.weight, self.bias, out=out[:half])
                hip_ops.bf16_gemm(x[half:], self.weight, self.bias, out=out[half:])
                return out
            return torch.nn.functional.linear(x, self.weight, self.bias)
        if self.kind == "gptq":
            if x.shape[0] > GPTQ_EXPAND_ROWS:
                # prefill regime: expand the weights once, plain library bf16 GEMM on them -- as W^T [N, K] where the
                # expansion kernel's tiles cover the matrix, the layout the library runs 12 % faster at 32768 rows
                nk = hip_ops.gptq_dequant_nk_supported(self.qweight, self.meta["bits"])
                w = hip_ops.gptq_dequant(self.qweight, self.qzeros, self.scales, self.meta["bits"],
                                         self.meta["group_size"], zero_offset=self.meta.get("zero_offset", 1), transposed=nk)
                if nk:
                    return torch.nn.functional.linear(x, w, self.bias)
                y = torch.matmul(x, w)
                return y if self.bias is None else y.add_(self.bias)
            return hip_ops.gptq_gemm(x, self.qweight, self.qzeros, self.scales, self.meta["bits"],
                                     self.meta["group_size"], bias=self.bias,
                                     zero_offset=self.meta.get("zero_offset", 1), defer=defer)
        if w8a8_use_library(x.shape[0], self.qweight.shape[1], self.qweight.shape[0]):
            # the same per-token quantisation and scaling, exact int32 sums from the library GEMM in between
            xq, xs = hip_ops.w8a8_quantize(x)
            acc = torch._int_mm(xq, self.qweight.t())
            return hip_ops.w8a8_scale_out(acc, xs, self.w_scale, bias=self.bias)
        return hip_ops.w8a8_gemm(x, self.qweight, self.w_scale, bias=self.bias, defer=defer)


class LinearColumns(torch.nn.Module):
    """Output columns ``[lo, hi)`` of a linear layer that several consumers share: the qkv projections of the two
    branches of the 70B attention read the same input, so the layer holds them as ONE linear (one weight stream, one
    launch) and hands each branch its columns; a branch called on its own (the meta tokens' one-off pass, a test of one
    branch) runs the shared linear through this view.  ``kind`` / ``weight`` / ``qweight`` / ``bias`` read as the
    branch's own projection would."""

    def __init__(self, parent: HymetaLinear, lo: int, hi: int):
        super().__init__()
        self._shared = (parent,)      # in a tuple: the owner registers the module, this view does not register it again
        self.lo, self.hi = lo, hi

    @property
    def parent(self) -> HymetaLinear:
        return self._shared[0]

    @property
    def kind(self) -> str:
        return self.parent.kind

    @property
    def weight(self) -> torch.Tensor:
        return self.parent.weight[self.lo:self.hi]

    @property
    def qweight(self) -> torch.Tensor:
        q = self.parent.qweight
        return q[:, self.lo:self.hi] if self.kind == "gptq" else q[self.lo:self.hi]

    @property
    def bias(self) -> Optional[torch.Tensor]:
        b = self.parent.bias
        return None if b is None else b[self.lo:self.hi]

    def forward(self, x: torch.Tensor, defer: bool = False) -> torch.Tensor:
        return self.parent(x)[:, self.lo:self.hi]

