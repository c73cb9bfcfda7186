"""Torch-tensor front end of the C ABI: checks tensors on the host, fills the
argument structs and enqueues the kernels on torch's current HIP stream.

torch is used for device memory and the stream handle only; every op below
runs in libhymeta_hip.so.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch

from . import _C

_DT = {torch.bfloat16: _C.HYMETA_BF16, torch.float32: _C.HYMETA_F32}


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require(cond: bool, msg: str) -> None:
    if not cond:
        raise ValueError(msg)


def _dev(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        _require(t.is_cuda, "hymeta ops need HIP device tensors")
        dev = dev or t.device
        _require(t.device == dev, "all tensors must live on one device")
    return dev


def _inner_contig(t: torch.Tensor, name: str) -> None:
    _require(t.stride(-1) == 1, f"{name}: innermost dimension must be contiguous")


# ---------------------------------------------------------------------------
# GLA decode
# ---------------------------------------------------------------------------
def gla_decode(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    g: Optional[torch.Tensor],
    state: torch.Tensor,
    slot_idx: torch.Tensor,
    scale: Optional[float] = None,
    out: Optional[torch.Tensor] = None,
    gate_mode: int = _C.GATE_NONE,
    lower_bound: Optional[torch.Tensor] = None,
    gate_normalizer: float = 16.0,
    gate_clamp: float = 0.95,
    out_norm_weight: Optional[torch.Tensor] = None,
    out_norm_eps: float = 1e-6,
    g_w1: Optional[torch.Tensor] = None,
    g_bias: Optional[torch.Tensor] = None,
    qkv_partials: Optional["PartialSums"] = None,
    qkv_cols: Optional[tuple] = None,
    g_w0: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """q [B,H,D]; k, v, g [B,Hkv,D]; state [slots,H,D,D] (one state per query head, the reference's
    layout) or [slots,Hkv,D,D] (one per key/value head; any gate but the 70B lower-bound one), updated
    in place; slot_idx [B] int32.

    Low-rank gate (7B): with ``g_w1`` bf16 [Hkv*D, 16] (and ``g_bias`` bf16 [Hkv*D]) ``g`` is the rank-16 input
    bf16 [B, 16] and the kernel applies the second linear of the gate projection itself.  With ``g_w0`` bf16
    [16, hidden] as well, ``g`` is the layer input bf16 [B, hidden] and the kernel applies both linears (state per
    key/value head only).

    ``qkv_partials`` (``PartialSums`` of the fused qkv projection) with ``qkv_cols = (q_col, k_col, v_col, g_col)``:
    q, k, v and the rank-16 gate input are read from the partial sums (the projection's reduce launch is skipped); then
    ``q`` / ``k`` / ``v`` / ``g`` are given as shapes only: ``(B, H, D)``, ``(B, Hkv, D)``, ``(B, Hkv, D)``, None.
    With ``g_w0`` as well (a quantised projection carries no rank columns; its partial sums may be the int32 ones of a
    W8A8 linear) ``g`` is the layer input bf16 [B, hidden] and ``g_col`` is unused; state per key/value head only.

    Returns out [B,H,D] bf16 (rows of skipped sequences, slot -1, are left as they were).
    """
    if qkv_partials is not None:
        return _gla_decode_from_partials(q, k, qkv_partials, qkv_cols, state, slot_idx, scale, out, gate_mode, gate_normalizer,
                                         gate_clamp, out_norm_weight, out_norm_eps, g_w1, g_bias, g, g_w0)
    _dev(q, k, v, g, state, slot_idx, lower_bound, out_norm_weight)
    B, H, D = q.shape
    Hkv = k.shape[1]
    _require(q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16,
             "q, k, v must be bfloat16")
    _require(k.shape == (B, Hkv, D) and v.shape == (B, Hkv, D), "k, v must be [B, Hkv, D]")
    if g_w1 is not None:
        _dev(g_w1, g_bias, g_w0)
        width = 16 if g_w0 is None else g_w0.shape[1]
        _require(g_w0 is None or (g_w0.dtype == torch.bfloat16 and g_w0.is_contiguous() and g_w0.dim() == 2 and g_w0.shape[0] == 16
                                  and width % 8 == 0 and state.shape[1] == Hkv != H),
                 "g_w0 must be contiguous bf16 [16, hidden], with a state per key/value head")
        _require(g is not None and g.shape == (B, width) and g.dtype == torch.bfloat16 and g.stride(1) == 1 and g.stride(0) % 8 == 0,
                 "low-rank gate: g must be bf16 [B, 16] (or [B, hidden] with g_w0) with rows a multiple of 8 elements apart")
        _require(g_w1.dtype == torch.bfloat16 and g_w1.is_contiguous() and g_w1.shape == (Hkv * D, 16), "g_w1 must be contiguous bf16 [Hkv*D, 16]")
        _require(g_bias is None or (g_bias.dtype == torch.bfloat16 and g_bias.is_contiguous() and g_bias.numel() == Hkv * D),
                 "g_bias must be contiguous bf16 [Hkv*D]")
    else:
        _require(g_w0 is None, "g_w0 needs g_w1")
        _require(g is None or g.shape == (B, Hkv, D), "g must be [B, Hkv, D]")
    _require(state.dim() == 4 and state.shape[2:] == (D, D) and state.shape[1] in (H, Hkv),
             "state must be [slots, H, D, D] or [slots, Hkv, D, D]")
    _require(state.shape[1] == H or gate_mode != _C.GATE_70B_LOWER_BOUND,
             "the lower-bound gate depends on the query head: the state must be per query head")
    _require(state.stride(3) == 1 and state.stride(2) == D, "state (Dk, Dv) blocks must be contiguous")
    _require(state.dtype in _DT, "state must be bfloat16 or float32")
    _require(slot_idx.dtype == torch.int32 and slot_idx.shape == (B,) and slot_idx.is_contiguous(),
             "slot_idx must be contiguous int32 [B]")
    for name, t in (("q", q), ("k", k), ("v", v), ("g", g)):
        if t is not None:
            _inner_contig(t, name)
    if out is None:
        out = torch.empty(B, H, D, dtype=torch.bfloat16, device=q.device)
    _require(out.shape == (B, H, D) and out.dtype == torch.bfloat16 and out.stride(-1) == 1,
             "out must be bf16 [B, H, D]")
    if lower_bound is not None:
        _require(lower_bound.dtype == torch.float32 and lower_bound.is_contiguous()
                 and lower_bound.numel() == H * D, "lower_bound must be contiguous fp32 [H*D]")
    if out_norm_weight is not None:
        _require(out_norm_weight.dtype == torch.bfloat16 and out_norm_weight.is_contiguous()
                 and out_norm_weight.numel() == D, "out_norm_weight must be contiguous bf16 [D]")
    a = _C.GlaDecodeArgs()
    a.q, a.k, a.v, a.g, a.out, a.state = _p(q), _p(k), _p(v), _p(g), _p(out), _p(state)
    a.slot_idx, a.lower_bound, a.out_norm_weight = _p(slot_idx), _p(lower_bound), _p(out_norm_weight)
    a.batch, a.num_heads, a.num_kv_heads, a.head_dim = B, H, Hkv, D
    a.q_stride_b, a.q_stride_h = q.stride(0), q.stride(1)
    a.k_stride_b, a.k_stride_h = k.stride(0), k.stride(1)
    a.v_stride_b, a.v_stride_h = v.stride(0), v.stride(1)
    if g is not None:
        a.g_stride_b, a.g_stride_h = g.stride(0), g.stride(1)
        _require(g.dtype in _DT, "g must be bfloat16 or float32")
        a.g_dtype = _DT[g.dtype]
    a.g_w1, a.g_bias = _p(g_w1), _p(g_bias)
    if g_w0 is not None:
        a.g_w0, a.g_hidden = _p(g_w0), g_w0.shape[1]
    a.out_stride_b, a.out_stride_h = out.stride(0), out.stride(1)
    a.state_stride_slot, a.state_stride_h = state.stride(0), state.stride(1)
    a.state_dtype = _DT[state.dtype]
    a.state_heads = state.shape[1]
    a.gate_mode = gate_mode
    a.scale = float(scale) if scale is not None else 0.0
    a.gate_normalizer, a.gate_clamp, a.out_norm_eps = gate_normalizer, gate_clamp, out_norm_eps
    _C.check(_C.load().hymeta_gla_decode(C.byref(a), _stream(q)), "hymeta_gla_decode")
    return out


def _gla_decode_from_partials(q_shape, k_shape, partials, cols, state, slot_idx, scale, out, gate_mode, gate_normalizer,
                              gate_clamp, out_norm_weight, out_norm_eps, g_w1, g_bias, g=None, g_w0=None) -> torch.Tensor:
    partials.check_live()
    B, H, D = q_shape
    Hkv = k_shape[1]
    _dev(state, slot_idx, out_norm_weight, g_w1, g_bias, out, g, g_w0)
    _require(partials.shape[0] == B, "partial sums must be slabs of B rows")
    _require(not partials.desc.is_int32 or state.shape[1] == Hkv != H, "int32 partial sums: state per key/value head only")
    if g_w0 is not None:
        _require(g_w0.dtype == torch.bfloat16 and g_w0.is_contiguous() and g_w0.dim() == 2 and g_w0.shape[0] == 16
                 and g_w0.shape[1] % 8 == 0 and state.shape[1] == Hkv != H,
                 "g_w0 must be contiguous bf16 [16, hidden], with a state per key/value head")
        _require(g is not None and g.shape == (B, g_w0.shape[1]) and g.dtype == torch.bfloat16 and g.stride(1) == 1 and g.stride(0) % 8 == 0,
                 "with g_w0, g must be the layer input bf16 [B, hidden] with rows a multiple of 8 elements apart")
    else:
        _require(g is None, "partial-sum inputs carry the rank-16 gate input themselves (or give g_w0 and the layer input)")
    _require(gate_mode == _C.GATE_7B and g_w1 is not None, "partial-sum inputs are wired for the 7B low-rank gate")
    _require(g_w1.dtype == torch.bfloat16 and g_w1.is_contiguous() and g_w1.shape == (Hkv * D, 16), "g_w1 must be contiguous bf16 [Hkv*D, 16]")
    _require(g_bias is None or (g_bias.dtype == torch.bfloat16 and g_bias.is_contiguous() and g_bias.numel() == Hkv * D),
             "g_bias must be contiguous bf16 [Hkv*D]")
    _require(state.dim() == 4 and state.shape[2:] == (D, D) and state.shape[1] in (H, Hkv) and state.stride(3) == 1
             and state.stride(2) == D and state.dtype in _DT, "state must be [slots, H | Hkv, D, D] with contiguous blocks")
    _require(slot_idx.dtype == torch.int32 and slot_idx.shape == (B,) and slot_idx.is_contiguous(), "slot_idx must be contiguous int32 [B]")
    if out is None:
        out = torch.empty(B, H, D, dtype=torch.bfloat16, device=partials.device)
    _require(out.shape == (B, H, D) and out.dtype == torch.bfloat16 and out.stride(-1) == 1, "out must be bf16 [B, H, D]")
    if out_norm_weight is not None:
        _require(out_norm_weight.dtype == torch.bfloat16 and out_norm_weight.is_contiguous() and out_norm_weight.numel() == D,
                 "out_norm_weight must be contiguous bf16 [D]")
    a = _C.GlaDecodeArgs()
    a.out, a.state, a.slot_idx, a.out_norm_weight = _p(out), _p(state), _p(slot_idx), _p(out_norm_weight)
    a.batch, a.num_heads, a.num_kv_heads, a.head_dim = B, H, Hkv, D
    a.out_stride_b, a.out_stride_h = out.stride(0), out.stride(1)
    a.state_stride_slot, a.state_stride_h = state.stride(0), state.stride(1)
    a.state_dtype, a.g_dtype = _DT[state.dtype], _C.HYMETA_BF16
    a.state_heads = state.shape[1]
    a.gate_mode = gate_mode
    a.scale = float(scale) if scale is not None else 0.0
    a.gate_normalizer, a.gate_clamp, a.out_norm_eps = gate_normalizer, gate_clamp, out_norm_eps
    a.g_w1, a.g_bias = _p(g_w1), _p(g_bias)
    d = partials.desc
    a.qkv_slabs, a.qkv_bias, a.qkv_splits, a.qkv_n = d.slabs, d.bias, d.splits, d.N
    if d.is_int32:
        a.qkv_x_scale, a.qkv_w_scale = d.x_scale, d.w_scale
    a.q_col, a.k_col, a.v_col, a.g_col = (int(c) for c in cols)
    if g_w0 is not None:
        a.g, a.g_stride_b, a.g_w0, a.g_hidden = _p(g), g.stride(0), _p(g_w0), g_w0.shape[1]
    _C.check(_C.load().hymeta_gla_decode(C.byref(a), _stream(out)), "hymeta_gla_decode")
    return out


# ---------------------------------------------------------------------------
# GLA varlen prefill
# ---------------------------------------------------------------------------
def gla_prefill_varlen(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    g: Optional[torch.Tensor],
    state: torch.Tensor,
    slot_idx: torch.Tensor,
    cu_seqlens: torch.Tensor,
    max_seqlen: int,
    context_lens: Optional[torch.Tensor] = None,
    meta_state: Optional[torch.Tensor] = None,
    scale: Optional[float] = None,
    out: Optional[torch.Tensor] = None,
    gate_mode: int = _C.GATE_NONE,
    lower_bound: Optional[torch.Tensor] = None,
    gate_normalizer: float = 16.0,
    gate_clamp: float = 0.95,
    out_norm_weight: Optional[torch.Tensor] = None,
    out_norm_eps: float = 1e-6,
    num_segments: int = 0,
) -> torch.Tensor:
    """q [T,H,D]; k, v, g [T,Hkv,D]; cu_seqlens int32 [nseq+1]; state [slots,H,D,D] or [slots,Hkv,D,D]
    (see gla_decode) updated in place.
    ``num_segments``: 0 lets the library decide whether to cut long sequences into parallel runs,
    1 forces a single run per (sequence, head), >1 forces that many."""
    _dev(q, k, v, g, state, slot_idx, cu_seqlens, context_lens, meta_state, lower_bound, out_norm_weight)
    T, H, D = q.shape
    Hkv = k.shape[1]
    nseq = cu_seqlens.numel() - 1
    _require(q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16,
             "q, k, v must be bfloat16")
    _require(k.shape == (T, Hkv, D) and v.shape == (T, Hkv, D), "k, v must be [T, Hkv, D]")
    _require(g is None or g.shape == (T, Hkv, D), "g must be [T, Hkv, D]")
    _require(state.dim() == 4 and state.shape[2:] == (D, D) and state.shape[1] in (H, Hkv),
             "state must be [slots, H, D, D] or [slots, Hkv, D, D]")
    SH = state.shape[1]
    _require(SH == H or gate_mode != _C.GATE_70B_LOWER_BOUND,
             "the lower-bound gate depends on the query head: the state must be per query head")
    _require(state.stride(3) == 1 and state.stride(2) == D, "state (Dk, Dv) blocks must be contiguous")
    _require(state.dtype in _DT, "state must be bfloat16 or float32")
    for name, t, n in (("slot_idx", slot_idx, nseq), ("cu_seqlens", cu_seqlens, nseq + 1),
                       ("context_lens", context_lens, nseq)):
        if t is not None:
            _require(t.dtype == torch.int32 and t.is_contiguous() and t.numel() == n,
                     f"{name} must be contiguous int32 [{n}]")
    for name, t in (("q", q), ("k", k), ("v", v), ("g", g)):
        if t is not None:
            _inner_contig(t, name)
    if meta_state is not None:
        _require(meta_state.shape == (SH, D, D) and meta_state.dtype == state.dtype
                 and meta_state.stride(2) == 1 and meta_state.stride(1) == D,
                 "meta_state must be [state heads, D, D] in the state dtype with contiguous blocks")
    if out is None:
        out = torch.empty(T, H, D, dtype=torch.bfloat16, device=q.device)
    _require(out.shape == (T, H, D) and out.dtype == torch.bfloat16 and out.stride(-1) == 1,
             "out must be bf16 [T, H, D]")
    a = _C.GlaPrefillArgs()
    a.q, a.k, a.v, a.g, a.out, a.state = _p(q), _p(k), _p(v), _p(g), _p(out), _p(state)
    a.slot_idx, a.cu_seqlens, a.context_lens = _p(slot_idx), _p(cu_seqlens), _p(context_lens)
    a.meta_state, a.lower_bound, a.out_norm_weight = _p(meta_state), _p(lower_bound), _p(out_norm_weight)
    a.num_seqs, a.total_tokens, a.max_seqlen = nseq, T, int(max_seqlen)
    a.num_heads, a.num_kv_heads, a.head_dim = H, Hkv, D
    a.q_stride_t, a.q_stride_h = q.stride(0), q.stride(1)
    a.k_stride_t, a.k_stride_h = k.stride(0), k.stride(1)
    a.v_stride_t, a.v_stride_h = v.stride(0), v.stride(1)
    if g is not None:
        _require(g.dtype in _DT, "g must be bfloat16 or float32")
        a.g_stride_t, a.g_stride_h = g.stride(0), g.stride(1)
        a.g_dtype = _DT[g.dtype]
    a.out_stride_t, a.out_stride_h = out.stride(0), out.stride(1)
    a.state_stride_slot, a.state_stride_h = state.stride(0), state.stride(1)
    if meta_state is not None:
        a.meta_state_stride_h = meta_state.stride(0)
    a.state_dtype = _DT[state.dtype]
    a.state_heads = SH
    a.gate_mode = gate_mode
    a.scale = float(scale) if scale is not None else 0.0
    a.gate_normalizer, a.gate_clamp, a.out_norm_eps = gate_normalizer, gate_clamp, out_norm_eps
    lib = _C.load()
    nseg = int(num_segments)
    if nseg == 0:
        nseg = int(lib.hymeta_gla_prefill_segments(nseq, SH, int(max_seqlen)))   # SH: heads that run as workgroups
    nseg = max(1, min(nseg, (int(max_seqlen) + 31) // 32))
    a.num_segments = nseg
    nbytes = int(lib.hymeta_gla_prefill_workspace_bytes(nseq, H, SH, nseg))
    if nbytes:
        ws = _workspace(q.device, nbytes, "gla_prefill")
        a.workspace, a.workspace_bytes = _p(ws), nbytes
    _C.check(lib.hymeta_gla_prefill_varlen(C.byref(a), _stream(q)), "hymeta_gla_prefill_varlen")
    return out


# ---------------------------------------------------------------------------
# Paged KV write
# ---------------------------------------------------------------------------
def reshape_and_cache(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
) -> None:
    """key/value [T,Hkv,D] -> caches [nblk,bs,Hkv,D] at slot_mapping [<=T] int64 (negative: skip)."""
    _dev(key, value, key_cache, value_cache, slot_mapping)
    _require(key.dtype == value.dtype == key_cache.dtype == value_cache.dtype,
             "key, value and both caches must share a dtype (kv_cache_dtype 'auto')")
    _require(key_cache.shape == value_cache.shape and key_cache.stride() == value_cache.stride(),
             "key and value caches must have identical layout")
    nblk, bs, Hkv, D = key_cache.shape
    _require(key.shape[1:] == (Hkv, D) and value.shape == key.shape, "key/value must be [T, Hkv, D]")
    _require(slot_mapping.dtype == torch.int64 and slot_mapping.is_contiguous(), "slot_mapping must be int64")
    n = slot_mapping.numel()
    _require(n <= key.shape[0], "slot_mapping longer than key")
    for name, t in (("key", key), ("value", value), ("key_cache", key_cache)):
        _inner_contig(t, name)
    a = _C.ReshapeAndCacheArgs()
    a.key, a.value, a.key_cache, a.value_cache = _p(key), _p(value), _p(key_cache), _p(value_cache)
    a.slot_mapping = _p(slot_mapping)
    a.num_tokens, a.num_kv_heads, a.head_dim, a.block_size = n, Hkv, D, bs
    a.elem_bytes = key.element_size()
    a.num_blocks = nblk
    a.key_stride_t, a.key_stride_h = key.stride(0), key.stride(1)
    a.value_stride_t, a.value_stride_h = value.stride(0), value.stride(1)
    a.cache_stride_block, a.cache_stride_pos, a.cache_stride_h = (
        key_cache.stride(0), key_cache.stride(1), key_cache.stride(2))
    _C.check(_C.load().hymeta_reshape_and_cache(C.byref(a), _stream(key)), "hymeta_reshape_and_cache")


# ---------------------------------------------------------------------------
# Paged decode attention (sliding window + meta keys)
# ---------------------------------------------------------------------------
_workspaces = {}
_retired = []
_TICKETED = ("gptq_gemm", "bf16_gemm", "w8a8_gemm", "moe_router")     # workspaces that start with split-K ticket counters


class _Workspace:
    __slots__ = ("buf", "eager_clean", "captures", "uses")

    def __init__(self, buf, eager_clean):
        self.buf, self.eager_clean, self.captures, self.uses = buf, eager_clean, set(), 0


def _capture_id(stream) -> int:
    cid = C.c_uint64(0)
    _C.check(_C.load().hymeta_stream_capture_id(C.c_void_p(stream.cuda_stream), C.byref(cid)), "hymeta_stream_capture_id")
    return int(cid.value)


def _workspace(device: torch.device, nbytes: int, tag: str = "paged_decode") -> torch.Tensor:
    """Grow-only scratch per (device, stream, op): split-KV partials, prefill segment states, split-K
    slabs.  Calls on one stream are ordered, so they share; calls of one op on two streams (a shared
    expert beside the routed ones) each get their own, ticket counters included.

    The split-K GEMMs keep ticket counters at the front of theirs, which must be zero when a call starts
    and which every completed call leaves zero.  Eagerly that is one zero fill when the buffer is made.
    Under graph capture a fill recorded once would only run when that graph is replayed, so every
    capture (told apart by its capture id) records its own clearing kernel in front of its first use of a
    ticketed workspace: each replay then starts from zero tickets whatever ran before it, and a buffer
    made during a capture is cleared eagerly before its first eager use."""
    stream = torch.cuda.current_stream(device)
    key = (device, stream.cuda_stream, tag)
    cid = _capture_id(stream) if torch.cuda.is_current_stream_capturing() else 0
    ticketed = tag in _TICKETED
    ws = _workspaces.get(key)
    if ws is None or ws.buf.numel() < nbytes:
        if ws is not None:
            _retired.append(ws.buf)     # a captured HIP graph may still hold its address: never hand it back
        size = max(nbytes, 1 << 20)
        if cid == 0:
            ws = _Workspace(torch.zeros(size, dtype=torch.uint8, device=device) if ticketed
                            else torch.empty(size, dtype=torch.uint8, device=device), True)
        else:
            ws = _Workspace(torch.empty(size, dtype=torch.uint8, device=device), False)
        _workspaces[key] = ws
    if ticketed:
        if cid != 0 and cid not in ws.captures:
            ws.buf[: _C.GEMM_TICKET_BYTES].zero_()      # a fill kernel node of this graph (no memset node)
            ws.captures.add(cid)
        elif cid == 0 and not ws.eager_clean:
            ws.buf[: _C.GEMM_TICKET_BYTES].zero_()
            ws.eager_clean = True
    ws.uses += 1
    return ws.buf


class PartialSums:
    """Result of a K-sliced linear whose slice sum is left to its consumer (``bf16_gemm`` / ``w8a8_gemm`` with
    ``defer=True``): ``add_rmsnorm`` takes it in place of ``x`` and adds the slices while it reads them, ``tensor()``
    runs the linear's own reduce launch.  The partial sums live in the linear's workspace: they must be consumed
    before the next linear of that kind on the stream (checked)."""

    def __init__(self, desc, shape, device, key, keep):
        self.desc, self.shape, self.device = desc, shape, device
        self._key, self._keep = key, keep            # keep: tensors the descriptor points into
        self._uses = _workspaces[key].uses

    def check_live(self) -> None:
        ws = _workspaces.get(self._key)
        _require(ws is not None and ws.uses == self._uses and ws.buf is self._keep[0],
                 "partial sums were overwritten: another linear used the workspace before they were consumed")

    def tensor(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.check_live()
        M, N = self.shape
        if out is None:
            out = torch.empty(M, N, dtype=torch.bfloat16, device=self.device)
        _require(out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be bf16 [M, N]")
        _C.check(_C.load().hymeta_gemm_partials_reduce(C.byref(self.desc), _p(out), out.stride(0),
                                                       C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                 "hymeta_gemm_partials_reduce")
        return out


def _partials(desc, M, N, device, tag, *keep):
    stream = torch.cuda.current_stream(device)
    key = (device, stream.cuda_stream, tag)
    return PartialSums(desc, (M, N), device, key, (_workspaces[key].buf,) + keep)


def reset_workspace_tickets(device: Optional[torch.device] = None) -> None:
    """Zero the ticket counters of every split-K workspace (of ``device``): the recovery step after a launch
    that failed or was aborted midway, which may leave tickets drawn."""
    for (dev, _, tag), ws in _workspaces.items():
        if tag in _TICKETED and (device is None or dev == torch.device(device)):
            ws.buf[: _C.GEMM_TICKET_BYTES].zero_()


def paged_decode_splits(batch: int, num_kv_heads: int, max_seqlen_k: int, window_left: int,
                        num_meta: int) -> int:
    return int(_C.load().hymeta_meta_swa_paged_decode_splits(batch, num_kv_heads, max_seqlen_k,
                                                             window_left, num_meta))


def meta_swa_paged_decode(
    q: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    cache_seqlens: torch.Tensor,
    block_tables: torch.Tensor,
    meta_key: Optional[torch.Tensor] = None,
    meta_value: Optional[torch.Tensor] = None,
    window_left: int = -1,
    scale: Optional[float] = None,
    max_seqlen_k: int = 0,
    num_splits: int = 0,
    out: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
    new_key: Optional[torch.Tensor] = None,
    new_value: Optional[torch.Tensor] = None,
    slot_mapping: Optional[torch.Tensor] = None,
    static_scratch: bool = False,
) -> torch.Tensor:
    """q [B,H,D]; caches [nblk,bs,Hkv,D]; cache_seqlens int32 [B]; block_tables int32 [B,maxblk];
    meta_key/meta_value [M,Hkv,D] or None.  Returns out [B,H,D] bf16.
    ``static_scratch``: run ``hymeta_meta_swa_paged_decode_static`` (split-KV partials in the library's
    own fixed device array; no workspace is allocated or passed; pass ``out`` to allocate nothing at all).
    With ``new_key`` / ``new_value`` [B,Hkv,D] and ``slot_mapping`` int64 [B] the call first writes the
    decoded tokens' K/V rows into the cache (the fused form of ``reshape_and_cache`` + decode)."""
    _dev(q, key_cache, value_cache, cache_seqlens, block_tables, meta_key, meta_value, out, workspace,
         new_key, new_value, slot_mapping)
    B, H, D = q.shape
    nblk, bs, Hkv, Dc = key_cache.shape
    _require(q.dtype == torch.bfloat16 and key_cache.dtype == torch.bfloat16
             and value_cache.dtype == torch.bfloat16, "q and the KV cache must be bfloat16")
    _require(Dc == D and value_cache.shape == key_cache.shape
             and value_cache.stride() == key_cache.stride(), "key/value cache layout mismatch")
    _require(cache_seqlens.dtype == torch.int32 and cache_seqlens.is_contiguous()
             and cache_seqlens.numel() == B, "cache_seqlens must be contiguous int32 [B]")
    _require(block_tables.dtype == torch.int32 and block_tables.dim() == 2
             and block_tables.shape[0] == B and block_tables.stride(1) == 1,
             "block_tables must be int32 [B, max_blocks]")
    _inner_contig(q, "q")
    _inner_contig(key_cache, "key_cache")
    M = 0
    if meta_key is not None:
        _require(meta_value is not None and meta_key.shape == meta_value.shape
                 and meta_key.stride() == meta_value.stride(), "meta key/value layout mismatch")
        _require(meta_key.dtype == torch.bfloat16 and meta_key.shape[1:] == (Hkv, D)
                 and meta_key.stride(-1) == 1, "meta_key must be bf16 [M, Hkv, D]")
        M = meta_key.shape[0]
    if out is None:
        out = torch.empty(B, H, D, dtype=torch.bfloat16, device=q.device)
    _require(out.shape == (B, H, D) and out.dtype == torch.bfloat16 and out.stride(-1) == 1,
             "out must be bf16 [B, H, D]")
    max_blocks = block_tables.shape[1]
    if max_seqlen_k <= 0:
        max_seqlen_k = max_blocks * bs
    lib = _C.load()
    if num_splits <= 0:
        num_splits = lib.hymeta_meta_swa_paged_decode_splits(B, Hkv, max_seqlen_k, window_left, M)
    need = 0 if static_scratch else lib.hymeta_meta_swa_paged_decode_workspace_bytes(B, H, D, num_splits)
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = _workspace(q.device, need)
    a = _C.PagedDecodeArgs()
    a.q, a.out, a.key_cache, a.value_cache = _p(q), _p(out), _p(key_cache), _p(value_cache)
    a.block_tables, a.cache_seqlens = _p(block_tables), _p(cache_seqlens)
    a.meta_key, a.meta_value = _p(meta_key), _p(meta_value)
    a.workspace = _p(workspace) if need else None
    a.workspace_bytes = workspace.numel() * workspace.element_size() if need else 0
    a.batch, a.num_heads, a.num_kv_heads, a.head_dim = B, H, Hkv, D
    a.block_size, a.max_blocks_per_seq, a.num_meta = bs, max_blocks, M
    a.window_left, a.max_seqlen_k, a.num_splits = int(window_left), int(max_seqlen_k), int(num_splits)
    a.q_stride_b, a.q_stride_h = q.stride(0), q.stride(1)
    a.out_stride_b, a.out_stride_h = out.stride(0), out.stride(1)
    if new_key is not None:
        _require(new_value is not None and slot_mapping is not None, "new_key needs new_value and slot_mapping")
        _require(new_key.shape == (B, Hkv, D) and new_value.shape == (B, Hkv, D)
                 and new_key.dtype == torch.bfloat16 and new_value.dtype == torch.bfloat16,
                 "new_key / new_value must be bf16 [B, Hkv, D]")
        _require(slot_mapping.dtype == torch.int64 and slot_mapping.is_contiguous() and slot_mapping.numel() == B,
                 "slot_mapping must be contiguous int64 [B]")
        _inner_contig(new_key, "new_key")
        _inner_contig(new_value, "new_value")
        a.new_key, a.new_value, a.slot_mapping = _p(new_key), _p(new_value), _p(slot_mapping)
        a.new_key_stride_b, a.new_key_stride_h = new_key.stride(0), new_key.stride(1)
        a.new_value_stride_b, a.new_value_stride_h = new_value.stride(0), new_value.stride(1)
    a.cache_stride_block, a.cache_stride_pos, a.cache_stride_h = (
        key_cache.stride(0), key_cache.stride(1), key_cache.stride(2))
    a.block_table_stride = block_tables.stride(0)
    if meta_key is not None:
        a.meta_stride_t, a.meta_stride_h = meta_key.stride(0), meta_key.stride(1)
    a.scale = float(scale) if scale is not None else 0.0
    if static_scratch:
        _C.check(lib.hymeta_meta_swa_paged_decode_static(C.byref(a), _stream(q)), "hymeta_meta_swa_paged_decode_static")
    else:
        _C.check(lib.hymeta_meta_swa_paged_decode(C.byref(a), _stream(q)), "hymeta_meta_swa_paged_decode")
    return out


# ---------------------------------------------------------------------------
# Varlen prefill attention (sliding window + meta keys)
# ---------------------------------------------------------------------------
def meta_swa_prefill(
    q: torch.Tensor,
    cu_seqlens_q: torch.Tensor,
    max_seqlen_q: int,
    key: Optional[torch.Tensor] = None,
    value: Optional[torch.Tensor] = None,
    cu_seqlens_k: Optional[torch.Tensor] = None,
    key_cache: Optional[torch.Tensor] = None,
    value_cache: Optional[torch.Tensor] = None,
    block_tables: Optional[torch.Tensor] = None,
    seq_lens_k: Optional[torch.Tensor] = None,
    meta_query: Optional[torch.Tensor] = None,
    meta_key: Optional[torch.Tensor] = None,
    meta_value: Optional[torch.Tensor] = None,
    window_left: int = -1,
    scale: Optional[float] = None,
    out: Optional[torch.Tensor] = None,
    max_seqlen_k: int = 0,
    num_splits: int = 0,
) -> torch.Tensor:
    """q [T,H,D].  Un-paged keys: key/value [Tk,Hkv,D] + cu_seqlens_k; paged keys: caches +
    block_tables + seq_lens_k.  Returns [(M if meta_query)+T, H, D] bf16, meta rows first.
    ``max_seqlen_k`` (0: unknown) and ``num_splits`` (0: library's choice) steer the key slicing of
    short chunks of long sequences."""
    _dev(q, cu_seqlens_q, key, value, cu_seqlens_k, key_cache, value_cache, block_tables,
         seq_lens_k, meta_query, meta_key, meta_value, out)
    T, H, D = q.shape
    nseq = cu_seqlens_q.numel() - 1
    paged = block_tables is not None
    _require(q.dtype == torch.bfloat16, "q must be bfloat16")
    _inner_contig(q, "q")
    _require(cu_seqlens_q.dtype == torch.int32 and cu_seqlens_q.is_contiguous(), "cu_seqlens_q must be int32")
    a = _C.SwaPrefillArgs()
    if paged:
        _require(key_cache is not None and value_cache is not None and seq_lens_k is not None,
                 "paged prefill needs key_cache, value_cache and seq_lens_k")
        nblk, bs, Hkv, Dc = key_cache.shape
        _require(Dc == D and key_cache.dtype == torch.bfloat16 and value_cache.shape == key_cache.shape
                 and value_cache.stride() == key_cache.stride() and key_cache.stride(-1) == 1,
                 "key/value cache layout mismatch")
        _require(block_tables.dtype == torch.int32 and block_tables.shape[0] == nseq
                 and block_tables.stride(1) == 1, "block_tables must be int32 [nseq, max_blocks]")
        _require(seq_lens_k.dtype == torch.int32 and seq_lens_k.is_contiguous()
                 and seq_lens_k.numel() == nseq, "seq_lens_k must be contiguous int32 [nseq]")
        a.key_cache, a.value_cache = _p(key_cache), _p(value_cache)
        a.block_tables, a.seq_lens_k = _p(block_tables), _p(seq_lens_k)
        a.block_size, a.max_blocks_per_seq = bs, block_tables.shape[1]
        a.cache_stride_block, a.cache_stride_pos, a.cache_stride_h = (
            key_cache.stride(0), key_cache.stride(1), key_cache.stride(2))
        a.block_table_stride = block_tables.stride(0)
    else:
        _require(key is not None and value is not None and cu_seqlens_k is not None,
                 "un-paged prefill needs key, value and cu_seqlens_k")
        Hkv = key.shape[1]
        _require(key.dtype == torch.bfloat16 and value.dtype == torch.bfloat16
                 and key.shape[1:] == (Hkv, D) and value.shape == key.shape, "key/value must be bf16 [Tk, Hkv, D]")
        _inner_contig(key, "key")
        _inner_contig(value, "value")
        _require(cu_seqlens_k.dtype == torch.int32 and cu_seqlens_k.is_contiguous()
                 and cu_seqlens_k.numel() == nseq + 1, "cu_seqlens_k must be contiguous int32 [nseq+1]")
        a.key, a.value, a.cu_seqlens_k = _p(key), _p(value), _p(cu_seqlens_k)
        a.k_stride_t, a.k_stride_h = key.stride(0), key.stride(1)
        a.v_stride_t, a.v_stride_h = value.stride(0), value.stride(1)
    M = 0
    if meta_key is not None:
        _require(meta_value is not None and meta_key.shape == meta_value.shape
                 and meta_key.stride() == meta_value.stride(), "meta key/value layout mismatch")
        _require(meta_key.dtype == torch.bfloat16 and meta_key.shape[1:] == (Hkv, D)
                 and meta_key.stride(-1) == 1, "meta_key must be bf16 [M, Hkv, D]")
        M = meta_key.shape[0]
        a.meta_key, a.meta_value = _p(meta_key), _p(meta_value)
        a.meta_stride_t, a.meta_stride_h = meta_key.stride(0), meta_key.stride(1)
    rows = T
    if meta_query is not None:
        _require(M > 0 and meta_query.shape == (M, H, D) and meta_query.dtype == torch.bfloat16
                 and meta_query.stride(-1) == 1, "meta_query must be bf16 [M, H, D]")
        a.meta_query = _p(meta_query)
        a.meta_q_stride_t, a.meta_q_stride_h = meta_query.stride(0), meta_query.stride(1)
        rows += M
    if out is None:
        out = torch.empty(rows, H, D, dtype=torch.bfloat16, device=q.device)
    _require(out.shape == (rows, H, D) and out.dtype == torch.bfloat16 and out.stride(-1) == 1,
             f"out must be bf16 [{rows}, H, D]")
    a.q, a.cu_seqlens_q, a.out = _p(q), _p(cu_seqlens_q), _p(out)
    a.num_seqs, a.total_q, a.max_seqlen_q = nseq, T, int(max_seqlen_q)
    a.num_heads, a.num_kv_heads, a.head_dim = H, Hkv, D
    a.num_meta, a.window_left = M, int(window_left)
    a.q_stride_t, a.q_stride_h = q.stride(0), q.stride(1)
    a.out_stride_t, a.out_stride_h = out.stride(0), out.stride(1)
    a.scale = float(scale) if scale is not None else 0.0
    lib = _C.load()
    a.max_seqlen_k = max(int(max_seqlen_k), 0)
    splits = int(num_splits) if num_splits > 0 else int(lib.hymeta_meta_swa_prefill_splits(
        T, H, M, M if meta_query is not None else 0, int(window_left), a.max_seqlen_k))
    a.num_splits = splits
    if splits > 1:
        ws = _workspace(q.device, int(lib.hymeta_meta_swa_prefill_workspace_bytes(rows, H, splits)), "swa_prefill")
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    _C.check(lib.hymeta_meta_swa_prefill(C.byref(a), _stream(q)), "hymeta_meta_swa_prefill")
    return out


# ---------------------------------------------------------------------------
# Layer glue: rotary embedding and the 70B two-branch norm combine
# ---------------------------------------------------------------------------
def rope_neox_(positions: torch.Tensor, q: torch.Tensor, k: Optional[torch.Tensor],
               cos_sin_cache: torch.Tensor, position_offset: int = 0) -> None:
    """In-place NeoX rotary embedding.  positions int64 [T]; q [T,H,D]; k [T,Hkv,D] or None;
    cos_sin_cache bf16 [max_pos, D] (cos | sin)."""
    _dev(positions, q, k, cos_sin_cache)
    T, H, D = q.shape
    _require(q.dtype == torch.bfloat16 and (k is None or k.dtype == torch.bfloat16), "q, k must be bfloat16")
    _require(positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == T,
             "positions must be contiguous int64 [T]")
    _require(cos_sin_cache.dtype == torch.bfloat16 and cos_sin_cache.is_contiguous()
             and cos_sin_cache.shape[1] == D, "cos_sin_cache must be contiguous bf16 [max_pos, D]")
    _inner_contig(q, "q")
    a = _C.RopeArgs()
    a.q, a.positions, a.cos_sin_cache = _p(q), _p(positions), _p(cos_sin_cache)
    a.num_tokens, a.num_heads, a.head_dim, a.rot_dim = T, H, D, D
    a.max_position = cos_sin_cache.shape[0]
    a.position_offset = int(position_offset)
    a.q_stride_t, a.q_stride_h = q.stride(0), q.stride(1)
    if k is not None:
        _require(k.shape[0] == T and k.shape[2] == D, "k must be [T, Hkv, D]")
        _inner_contig(k, "k")
        a.k, a.num_kv_heads = _p(k), k.shape[1]
        a.k_stride_t, a.k_stride_h = k.stride(0), k.stride(1)
    _C.check(_C.load().hymeta_rope_neox(C.byref(a), _stream(q)), "hymeta_rope_neox")


def rope_from_partials(partials: "PartialSums", positions: torch.Tensor, cos_sin_cache: torch.Tensor, num_heads: int,
                       num_kv_heads: int, position_offset: int = 0) -> torch.Tensor:
    """The fused qkv projection finished from its K slices' partial sums with NeoX RoPE on the q and k heads: returns
    the projection's bf16 ``[T, N]`` output (q heads, k heads, then v and whatever rides behind), the bits of the plain
    linear followed by ``rope_neox_`` in one launch instead of two."""
    partials.check_live()
    T, N = partials.shape
    D = cos_sin_cache.shape[1]
    _dev(positions, cos_sin_cache)
    _require(positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == T,
             "positions must be contiguous int64 [T]")
    _require(cos_sin_cache.dtype == torch.bfloat16 and cos_sin_cache.is_contiguous(), "cos_sin_cache must be contiguous bf16 [max_pos, D]")
    _require((num_heads + 2 * num_kv_heads) * D <= N, "the projection is narrower than its q, k and v heads")
    out = torch.empty(T, N, dtype=torch.bfloat16, device=partials.device)
    a = _C.RopeArgs()
    a.q, a.positions, a.cos_sin_cache = _p(out), _p(positions), _p(cos_sin_cache)
    a.num_tokens, a.num_heads, a.num_kv_heads, a.head_dim, a.rot_dim = T, num_heads, num_kv_heads, D, D
    a.max_position = cos_sin_cache.shape[0]
    a.position_offset = int(position_offset)
    a.q_stride_t, a.q_stride_h = N, D
    _C.check(_C.load().hymeta_rope_neox_partials(C.byref(a), C.byref(partials.desc), _stream(out)), "hymeta_rope_neox_partials")
    return out


def _combine_args(o_attn, o_lin, weight1, weight2, stats, out, eps):
    T, Hd = o_attn.shape
    _require(o_lin.shape == (T, Hd) and o_attn.dtype == torch.bfloat16 and o_lin.dtype == torch.bfloat16,
             "o_attn, o_lin must be bf16 [T, hidden_local]")
    _require(o_attn.stride(1) == 1 and o_lin.stride(1) == 1, "branch outputs must be row-contiguous")
    a = _C.CombineArgs()
    a.o_attn, a.o_lin, a.stats, a.out = _p(o_attn), _p(o_lin), _p(stats), _p(out)
    a.weight1, a.weight2 = _p(weight1), _p(weight2)
    a.num_tokens, a.hidden_local = T, Hd
    a.attn_stride_t, a.lin_stride_t = o_attn.stride(0), o_lin.stride(0)
    if out is not None:
        a.out_stride_t = out.stride(0)
    a.eps = eps
    return a


def rms_sumsq2(o_attn: torch.Tensor, o_lin: torch.Tensor) -> torch.Tensor:
    """Local mean of squares of both branches: fp32 [T, 2] (all-reduce it, divide by tp)."""
    _dev(o_attn, o_lin)
    stats = torch.empty(o_attn.shape[0], 2, dtype=torch.float32, device=o_attn.device)
    a = _combine_args(o_attn, o_lin, None, None, stats, None, 0.0)
    _C.check(_C.load().hymeta_rms_sumsq2(C.byref(a), _stream(o_attn)), "hymeta_rms_sumsq2")
    return stats


def hybrid_combine(o_attn: torch.Tensor, o_lin: torch.Tensor, weight1: torch.Tensor, weight2: torch.Tensor,
                   eps: float = 1e-6, stats: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16((RMSNorm1(o_attn) + RMSNorm2(o_lin)) / 2); ``stats`` = all-reduced mean squares or None (tp=1)."""
    _dev(o_attn, o_lin, weight1, weight2, stats, out)
    for w in (weight1, weight2):
        _require(w.dtype == torch.float32 and w.is_contiguous() and w.numel() == o_attn.shape[1],
                 "norm weights must be contiguous fp32 [hidden_local]")
    if stats is not None:
        _require(stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape == (o_attn.shape[0], 2),
                 "stats must be contiguous fp32 [T, 2]")
    if out is None:
        out = torch.empty_like(o_attn)
    a = _combine_args(o_attn, o_lin, weight1, weight2, stats, out, eps)
    _C.check(_C.load().hymeta_hybrid_combine(C.byref(a), _stream(o_attn)), "hymeta_hybrid_combine")
    return out


INT8_ROWS_ATTR = "_hymeta_int8_rows"


def int8_rows_of(x: torch.Tensor):
    """``(xq int8 [T, K], x_scale fp32 [T])`` a producer attached to exactly this tensor object (``add_rmsnorm(...,
    int8=True)``), or None.  A view or copy of the tensor does not carry them."""
    return getattr(x, INT8_ROWS_ATTR, None)


def add_rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, residual: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, int8: bool = False) -> torch.Tensor:
    """RMS norm of the decoder blocks.  With ``residual``: residual <- bf16(x + residual) in place
    and the norm is taken of that sum (the blocks' ``mlp_norm(hidden, residual)`` call).  ``x`` may be the
    ``PartialSums`` of a K-sliced linear (needs ``residual``): the slices are added as the input is read.

    ``int8``: the launch also writes the dynamic per-token int8 quantisation of its output (the bits ``w8a8_quantize``
    gives) and attaches ``(xq, x_scale)`` to the returned tensor (``int8_rows_of``); a W8A8 linear called on that very
    tensor skips its own quantise launch."""
    if isinstance(x, PartialSums):
        x.check_live()
        M, N = x.shape
        _dev(weight, residual, out)
        _require(residual is not None and residual.shape == (M, N) and residual.dtype == torch.bfloat16 and residual.stride(1) == 1,
                 "partial sums need a row-contiguous bf16 residual [T, hidden]")
        _require(weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.numel() == N, "weight must be contiguous bf16 [hidden]")
        if out is None:
            out = torch.empty(M, N, dtype=torch.bfloat16, device=residual.device)
        _require(out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be row-contiguous bf16 [T, hidden]")
        a = _C.AddRmsNormArgs(x=None, residual=residual.data_ptr(), weight=weight.data_ptr(), out=out.data_ptr(), num_tokens=M,
                              hidden=N, x_stride_t=0, residual_stride_t=residual.stride(0), out_stride_t=out.stride(0), eps=eps)
        if int8:
            xq = torch.empty(M, N, dtype=torch.int8, device=out.device)
            xs = torch.empty(M, dtype=torch.float32, device=out.device)
            _C.check(_C.load().hymeta_add_rmsnorm_int8(C.byref(a), C.byref(x.desc), _p(xq), _p(xs), _stream(residual)),
                     "hymeta_add_rmsnorm_int8")
            setattr(out, INT8_ROWS_ATTR, (xq, xs))
            return out
        _C.check(_C.load().hymeta_add_rmsnorm_partials(C.byref(a), C.byref(x.desc), _stream(residual)), "hymeta_add_rmsnorm_partials")
        return out
    _dev(x, weight, residual, out)
    _require(x.dim() == 2 and x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [T, hidden]")
    _require(weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.numel() == x.shape[1],
             "weight must be contiguous bf16 [hidden]")
    if residual is not None:
        _require(residual.shape == x.shape and residual.dtype == torch.bfloat16 and residual.stride(1) == 1,
                 "residual must be row-contiguous bf16 [T, hidden]")
    if out is None:
        out = torch.empty_like(x)
    _require(out.shape == x.shape and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be row-contiguous bf16 [T, hidden]")
    a = _C.AddRmsNormArgs(x=x.data_ptr(), residual=residual.data_ptr() if residual is not None else None,
                          weight=weight.data_ptr(), out=out.data_ptr(), num_tokens=x.shape[0], hidden=x.shape[1],
                          x_stride_t=x.stride(0), residual_stride_t=residual.stride(0) if residual is not None else 0,
                          out_stride_t=out.stride(0), eps=eps)
    if int8:
        xq = torch.empty(x.shape, dtype=torch.int8, device=out.device)
        xs = torch.empty(x.shape[0], dtype=torch.float32, device=out.device)
        _C.check(_C.load().hymeta_add_rmsnorm_int8(C.byref(a), None, _p(xq), _p(xs), _stream(x)), "hymeta_add_rmsnorm_int8")
        setattr(out, INT8_ROWS_ATTR, (xq, xs))
        return out
    _C.check(_C.load().hymeta_add_rmsnorm(C.byref(a), _stream(x)), "hymeta_add_rmsnorm")
    return out


LOWRANK_GATE_ROWS = 256     # above this the two library GEMMs are ahead (w0 is re-read per token here)


def lowrank_gate(x: torch.Tensor, w0: torch.Tensor, w1: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16( bf16(x w0^T) w1^T + bias ): the rank-16 gate projection of the 7B GLA layer in one launch.
    x [T, hidden]; w0 [16, hidden]; w1 [N, 16]; bias [N] -> [T, N]."""
    _dev(x, w0, w1, bias, out)
    T, hidden = x.shape
    N = w1.shape[0]
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [T, hidden]")
    _require(w0.dtype == torch.bfloat16 and w0.is_contiguous() and w0.shape == (16, hidden), "w0 must be contiguous bf16 [16, hidden]")
    _require(w1.dtype == torch.bfloat16 and w1.is_contiguous() and w1.shape == (N, 16), "w1 must be contiguous bf16 [N, 16]")
    _require(bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N),
             "bias must be contiguous bf16 [N]")
    if out is None:
        out = torch.empty(T, N, dtype=torch.bfloat16, device=x.device)
    _require(out.shape == (T, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be row-contiguous bf16 [T, N]")
    a = _C.LowrankGateArgs(x=x.data_ptr(), w0=w0.data_ptr(), w1=w1.data_ptr(), bias=_p(bias), out=out.data_ptr(),
                           num_tokens=T, hidden=hidden, rank=16, out_features=N,
                           x_stride_t=x.stride(0), out_stride_t=out.stride(0))
    _C.check(_C.load().hymeta_lowrank_gate(C.byref(a), _stream(x)), "hymeta_lowrank_gate")
    return out


# ---------------------------------------------------------------------------
# Quantised linears
# ---------------------------------------------------------------------------
def gptq_gemm(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
              bits: int, group_size: int, bias: Optional[torch.Tensor] = None, zero_offset: int = 1,
              split_k: int = 0, out: Optional[torch.Tensor] = None, defer: bool = False):
    """x bf16 [M,K] @ GPTQ(qweight int32 [K/pack,N], qzeros int32 [K/group,N/pack], scales fp16|bf16
    [K/group,N]) (+ bias bf16 [N]) -> bf16 [M,N]."""
    _dev(x, qweight, qzeros, scales, bias, out)
    M, K = x.shape
    N = qweight.shape[1]
    pack = 32 // bits
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [M, K]")
    _require(qweight.dtype == torch.int32 and qweight.is_contiguous() and qweight.shape[0] * pack == K,
             "qweight must be contiguous int32 [K/pack, N]")
    _require(qzeros.dtype == torch.int32 and qzeros.is_contiguous()
             and qzeros.shape == (K // group_size, N // pack), "qzeros must be contiguous int32 [K/group, N/pack]")
    _require(scales.dtype in (torch.float16, torch.bfloat16) and scales.is_contiguous()
             and scales.shape == (K // group_size, N), "scales must be contiguous fp16/bf16 [K/group, N]")
    if bias is not None:
        _require(bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N, "bias must be bf16 [N]")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _require(out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be bf16 [M, N]")
    lib = _C.load()
    a = _C.GptqGemmArgs()
    # split_k <= 0: the library chooses how to cut the work; its workspace bound does not depend on K
    need = int(lib.hymeta_gptq_gemm_workspace_bytes(M, N, max(split_k, 0)))
    if need:
        ws = _workspace(x.device, need, "gptq_gemm")
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    a.x, a.qweight, a.qzeros, a.scales, a.bias, a.out = _p(x), _p(qweight), _p(qzeros), _p(scales), _p(bias), _p(out)
    a.M, a.K, a.N, a.bits, a.group_size = M, K, N, bits, group_size
    a.scales_dtype = _C.HYMETA_F16 if scales.dtype == torch.float16 else _C.HYMETA_BF16
    a.zero_offset, a.split_k = zero_offset, max(split_k, 0)
    a.x_stride_m, a.out_stride_m = x.stride(0), out.stride(0)
    if defer and a.split_k == 0 and need:
        desc = _C.GemmPartials()
        _C.check(lib.hymeta_gptq_gemm_deferred(C.byref(a), C.byref(desc), _stream(x)), "hymeta_gptq_gemm_deferred")
        return _partials(desc, M, N, x.device, "gptq_gemm", bias) if desc.splits > 0 else out
    _C.check(lib.hymeta_gptq_gemm(C.byref(a), _stream(x)), "hymeta_gptq_gemm")
    return out


def gptq_dequant_nk_supported(qweight: torch.Tensor, bits: int) -> bool:
    """The transposed expansion covers whole 64-column x 32-packed-row tiles."""
    pack = 32 // bits
    return qweight.shape[1] % 64 == 0 and (qweight.shape[0] * pack) % (32 * pack) == 0


def gptq_dequant(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, bits: int, group_size: int,
                 zero_offset: int = 1, out: Optional[torch.Tensor] = None, transposed: bool = False) -> torch.Tensor:
    """GPTQ tensors -> bf16 weights [K, N] (so that y = x @ W), or with ``transposed`` [N, K] as torch stores a Linear
    weight (so that y = F.linear(x, W)): the layout the library GEMM runs faster at prefill-sized batches."""
    _dev(qweight, qzeros, scales, out)
    pack = 32 // bits
    K, N = qweight.shape[0] * pack, qweight.shape[1]
    _require(qweight.dtype == torch.int32 and qweight.is_contiguous(), "qweight must be contiguous int32 [K/pack, N]")
    _require(qzeros.dtype == torch.int32 and qzeros.is_contiguous() and qzeros.shape == (K // group_size, N // pack),
             "qzeros must be contiguous int32 [K/group, N/pack]")
    _require(scales.dtype in (torch.float16, torch.bfloat16) and scales.is_contiguous()
             and scales.shape == (K // group_size, N), "scales must be contiguous fp16/bf16 [K/group, N]")
    shape = (N, K) if transposed else (K, N)
    if out is None:
        out = torch.empty(shape, dtype=torch.bfloat16, device=qweight.device)
    _require(out.shape == shape and out.dtype == torch.bfloat16 and out.is_contiguous(),
             "out must be contiguous bf16 [K, N] ([N, K] when transposed)")
    sd = _C.HYMETA_F16 if scales.dtype == torch.float16 else _C.HYMETA_BF16
    if transposed:
        _require(gptq_dequant_nk_supported(qweight, bits), "transposed expansion needs N % 64 == 0 and K % (32 * pack) == 0")
        _C.check(_C.load().hymeta_gptq_dequant_nk(_p(qweight), _p(qzeros), _p(scales), _p(out), K, N, bits, group_size, sd,
                                                  zero_offset, _stream(qweight)), "hymeta_gptq_dequant_nk")
        return out
    _C.check(_C.load().hymeta_gptq_dequant(_p(qweight), _p(qzeros), _p(scales), _p(out), K, N, bits, group_size, sd,
                                           zero_offset, _stream(qweight)), "hymeta_gptq_dequant")
    return out


BF16_GEMM_MAX_ROWS = 128      # 65 to 128 rows: the streaming kernel only (K a multiple of 64, at least 256)
BF16_GEMM_TALL_ROWS = 256     # 129 to 256 rows: the shapes the library says it takes (hymeta_bf16_gemm_takes)


def bf16_gemm_preferred(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Where the kernel is used instead of the library GEMM: every decode-sized shape it takes (measured with
    cold weights on the 7B projections at 1 / 16 / 64 rows: ahead on all of them but the 3584 x 3584 output
    projection at 64 rows, 16 against 14 us)."""
    return bf16_gemm_supported(x, weight)


def bf16_gemm_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes the bf16 linear kernels take: decode-sized batches, and from 129 to 256 rows what the library's
    129..256-row kernel covers (``hymeta_bf16_gemm_takes``)."""
    if not (x.dim() == 2 and 0 < x.shape[0] <= BF16_GEMM_TALL_ROWS and x.is_cuda and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and x.stride(1) == 1 and weight.stride(1) == 1
            and weight.shape[1] % 32 == 0 and weight.shape[0] % 64 == 0
            and (x.shape[0] <= 64 or (weight.shape[1] % 64 == 0 and weight.shape[1] >= 256))
            and x.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0):
        return False
    if x.shape[0] <= BF16_GEMM_MAX_ROWS:
        return True
    return bool(_C.load().hymeta_bf16_gemm_takes(x.shape[0], weight.shape[1], weight.shape[0], 0))


def bf16_gemm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, split_k: int = 0, defer: bool = False):
    """x bf16 [M, K] @ weight bf16 [N, K]^T (+ bias bf16 [N]) -> bf16 [M, N] for the shapes of ``bf16_gemm_supported``;
    ``split_k`` 0: library's choice.
    ``defer``: when the library cuts K into slices, return their ``PartialSums`` instead of summing them."""
    _dev(x, weight, bias, out)
    M, K = x.shape
    N = weight.shape[0]
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [M, K]")
    _require(weight.dtype == torch.bfloat16 and weight.stride(1) == 1 and weight.shape == (N, K),
             "weight must be row-contiguous bf16 [N, K]")
    if bias is not None:
        _require(bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N, "bias must be bf16 [N]")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _require(out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be bf16 [M, N]")
    a = _C.Bf16GemmArgs()
    a.x, a.weight, a.bias, a.out = _p(x), _p(weight), _p(bias), _p(out)
    a.M, a.K, a.N = M, K, N
    a.x_stride_m, a.w_stride_n, a.out_stride_m = x.stride(0), weight.stride(0), out.stride(0)
    lib = _C.load()
    a.split_k = max(int(split_k), 0)
    nbytes = int(lib.hymeta_bf16_gemm_workspace_bytes(M, N, a.split_k))
    if nbytes > 0:
        ws = _workspace(x.device, nbytes, "bf16_gemm")
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    if defer and a.split_k == 0 and nbytes > 0:
        desc = _C.GemmPartials()
        _C.check(lib.hymeta_bf16_gemm_deferred(C.byref(a), C.byref(desc), _stream(x)), "hymeta_bf16_gemm_deferred")
        return _partials(desc, M, N, x.device, "bf16_gemm", bias) if desc.splits > 0 else out
    _C.check(lib.hymeta_bf16_gemm(C.byref(a), _stream(x)), "hymeta_bf16_gemm")
    return out


def logits_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes ``bf16_logits`` takes (any number of rows: they go 64 at a time)."""
    return (x.dim() == 2 and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.is_cuda
            and x.stride(1) == 1 and weight.stride(1) == 1 and weight.shape[1] % 64 == 0 and weight.shape[1] >= 256
            and weight.shape[0] % 64 == 0 and x.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0)


def bf16_logits(x: torch.Tensor, weight: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x bf16 [M, K] @ weight bf16 [N, K]^T -> fp32 [M, N], fp32 accumulation, result not rounded: the logits of a bf16
    lm_head (the fp32 GEMM of the operands widened to fp32, up to the order of the sums).  Rows go through the
    weight-streaming kernels 128 or 256 at a time."""
    _dev(x, weight, out)
    M, K = x.shape
    N = weight.shape[0]
    _require(logits_supported(x, weight) and weight.shape == (N, K), "bf16_logits: x bf16 [M, K], weight bf16 [N, K], K % 64 == 0, N % 64 == 0")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    _require(out.shape == (M, N) and out.dtype == torch.float32 and out.stride(1) == 1 and out.stride(0) % 4 == 0, "out must be fp32 [M, N]")
    lib = _C.load()
    r0 = 0
    while r0 < M:
        # 256 rows at a time where the 129..256-row kernel takes the matrix (one pass over the weights instead of two)
        rows = min(M - r0, BF16_GEMM_TALL_ROWS)
        if rows > BF16_GEMM_MAX_ROWS and not lib.hymeta_bf16_gemm_takes(rows, K, N, 2):
            rows = BF16_GEMM_MAX_ROWS
        xs, os_ = x[r0:r0 + rows], out[r0:r0 + rows]
        r0 += rows
        a = _C.Bf16GemmArgs()
        a.x, a.weight, a.out = _p(xs), _p(weight), _p(os_)
        a.M, a.K, a.N = xs.shape[0], K, N
        a.x_stride_m, a.w_stride_n, a.out_stride_m = x.stride(0), weight.stride(0), out.stride(0)
        _C.check(lib.hymeta_bf16_gemm_f32out(C.byref(a), _stream(x)), "hymeta_bf16_gemm_f32out")
    return out


def advance_step(num_seqs: int, num_queries: int, block_size: int, seq_lens: torch.Tensor, slot_mapping: torch.Tensor,
                 block_tables: torch.Tensor, input_tokens: Optional[torch.Tensor] = None,
                 sampled_token_ids: Optional[torch.Tensor] = None, input_positions: Optional[torch.Tensor] = None) -> None:
    """Multi-step decoding, in place and in one launch: sequence i < num_queries gets its sampled token as input, its
    position (= its length so far), length + 1 and the cache slot of that position; padding rows get slot -1."""
    _dev(seq_lens, slot_mapping, block_tables, input_tokens, sampled_token_ids, input_positions)
    _require(seq_lens.dtype == torch.int32 and seq_lens.is_contiguous() and seq_lens.numel() >= num_seqs, "seq_lens must be contiguous int32 [num_seqs]")
    _require(slot_mapping.dtype == torch.int64 and slot_mapping.is_contiguous() and slot_mapping.numel() >= num_seqs,
             "slot_mapping must be contiguous int64 [num_seqs]")
    _require(block_tables.dtype == torch.int32 and block_tables.dim() == 2 and block_tables.stride(1) == 1
             and block_tables.shape[0] >= num_queries, "block_tables must be int32 [num_seqs, blocks]")
    for name, t in (("input_tokens", input_tokens), ("input_positions", input_positions)):
        _require(t is None or (t.dtype == torch.int64 and t.is_contiguous() and t.numel() >= num_queries), f"{name} must be contiguous int64")
    if sampled_token_ids is not None:
        sampled_token_ids = sampled_token_ids.reshape(-1)
        _require(sampled_token_ids.dtype == torch.int64 and sampled_token_ids.is_contiguous() and sampled_token_ids.numel() >= num_queries,
                 "sampled_token_ids must be contiguous int64 [num_queries(, 1)]")
    _C.check(_C.load().hymeta_advance_step(num_seqs, num_queries, block_size, _p(input_tokens), _p(sampled_token_ids), _p(input_positions),
                                           _p(seq_lens), _p(slot_mapping), _p(block_tables), block_tables.stride(0), _stream(seq_lens)),
             "hymeta_advance_step")


def argmax_rows(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Greedy sampling: int64 [M] indices of the largest fp32 logit of each row (the lowest index among equals)."""
    _dev(logits, out)
    _require(logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1, "logits must be row-contiguous fp32 [M, N]")
    M, N = logits.shape
    if out is None:
        out = torch.empty(M, dtype=torch.int64, device=logits.device)
    _require(out.shape == (M,) and out.dtype == torch.int64 and out.is_contiguous(), "out must be contiguous int64 [M]")
    lib = _C.load()
    ws = _workspace(logits.device, int(lib.hymeta_argmax_rows_workspace_bytes(M)), "argmax_rows") if M > 0 else None
    _C.check(lib.hymeta_argmax_rows(_p(logits), logits.stride(0), M, N, _p(out), _p(ws), ws.numel() if ws is not None else 0,
                                    _stream(logits)), "hymeta_argmax_rows")
    return out


def w8a8_gemm(x: torch.Tensor, qweight: torch.Tensor, w_scale: torch.Tensor,
              bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, defer: bool = False):
    """x bf16 [M,K] -> dynamic per-token int8; qweight int8 [N,K]; w_scale fp32 [N] -> bf16 [M,N].
    ``defer``: when the library cuts K into slices, return their ``PartialSums`` instead of summing them."""
    _dev(x, qweight, w_scale, bias, out)
    M, K = x.shape
    N = qweight.shape[0]
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [M, K]")
    _require(qweight.dtype == torch.int8 and qweight.is_contiguous() and qweight.shape == (N, K),
             "qweight must be contiguous int8 [N, K]")
    _require(w_scale.dtype == torch.float32 and w_scale.is_contiguous() and w_scale.numel() == N,
             "w_scale must be contiguous fp32 [N]")
    if bias is not None:
        _require(bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N, "bias must be bf16 [N]")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _require(out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be bf16 [M, N]")
    ready = int8_rows_of(x)
    xq, xs = ready if ready is not None else (torch.empty(M, K, dtype=torch.int8, device=x.device),
                                              torch.empty(M, dtype=torch.float32, device=x.device))
    a = _C.W8A8GemmArgs()
    a.x, a.qweight, a.w_scale, a.bias, a.out, a.xq, a.x_scale = (
        None if ready is not None else _p(x), _p(qweight), _p(w_scale), _p(bias), _p(out), _p(xq), _p(xs))
    a.M, a.K, a.N = M, K, N
    a.x_stride_m, a.out_stride_m = x.stride(0), out.stride(0)
    lib = _C.load()
    a.split_k = 0        # the library chooses the kernel and the decomposition; the workspace bound covers every choice
    ws = _workspace(x.device, int(lib.hymeta_w8a8_gemm_workspace_bytes(M, N, 0)), "w8a8_gemm")
    a.workspace, a.workspace_bytes = _p(ws), ws.numel()
    if defer:
        desc = _C.GemmPartials()
        _C.check(lib.hymeta_w8a8_gemm_deferred(C.byref(a), C.byref(desc), _stream(x)), "hymeta_w8a8_gemm_deferred")
        return _partials(desc, M, N, x.device, "w8a8_gemm", xs, w_scale, bias) if desc.splits > 0 else out
    _C.check(lib.hymeta_w8a8_gemm(C.byref(a), _stream(x)), "hymeta_w8a8_gemm")
    return out


def gemm_silu_mul_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes the gate/up projection with the activation in its epilogue takes: decode-sized batches of a bf16
    ``[2 I, K]`` weight (gate rows, then up rows), any batch of an int8 one."""
    esize = 1 if weight.dtype == torch.int8 else 2
    rows = BF16_GEMM_TALL_ROWS if weight.dtype == torch.bfloat16 else 1 << 30      # int8: the tiled kernel above the streaming rows
    if not (x.dim() == 2 and 0 < x.shape[0] <= rows and x.is_cuda and x.dtype == torch.bfloat16
            and weight.dtype in (torch.bfloat16, torch.int8) and x.stride(1) == 1 and weight.is_contiguous()
            and (weight.shape[1] * esize) % 128 == 0 and weight.shape[1] * esize >= 512 and weight.shape[0] % 128 == 0
            and x.stride(0) % 8 == 0):
        return False
    if weight.dtype == torch.bfloat16 and x.shape[0] > BF16_GEMM_MAX_ROWS:
        return bool(_C.load().hymeta_bf16_gemm_takes(x.shape[0], weight.shape[1], weight.shape[0], 1))
    return True


def gemm_silu_mul(x: torch.Tensor, weight: torch.Tensor, w_scale: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``silu(x gate^T) * (x up^T)`` in one launch: x bf16 [M, K]; weight [2 I, K] bf16 (M <= 128, wide ones to 256), or int8 with ``w_scale``
    fp32 [2 I] (dynamic per-token int8 activations; any M) -> bf16 [M, I].  Bit for bit the linear followed by
    ``silu_and_mul``."""
    _dev(x, weight, w_scale, out)
    M, K = x.shape
    N = weight.shape[0]
    _require(gemm_silu_mul_supported(x, weight) and weight.shape == (N, K), "shape not covered: run the linear and silu_and_mul")
    if out is None:
        out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=x.device)
    _require(out.shape == (M, N // 2) and out.dtype == torch.bfloat16 and out.stride(1) == 1 and out.stride(0) % 8 == 0,
             "out must be row-contiguous bf16 [M, I]")
    lib = _C.load()
    if weight.dtype == torch.bfloat16:
        a = _C.Bf16GemmArgs()
        a.x, a.weight, a.out = _p(x), _p(weight), _p(out)
        a.M, a.K, a.N = M, K, N
        a.x_stride_m, a.w_stride_n, a.out_stride_m = x.stride(0), weight.stride(0), out.stride(0)
        _C.check(lib.hymeta_bf16_gemm_silu_mul(C.byref(a), _stream(x)), "hymeta_bf16_gemm_silu_mul")
        return out
    _require(w_scale is not None and w_scale.dtype == torch.float32 and w_scale.is_contiguous() and w_scale.numel() == N,
             "w_scale must be contiguous fp32 [2 I]")
    ready = int8_rows_of(x)
    xq, xs = ready if ready is not None else (torch.empty(M, K, dtype=torch.int8, device=x.device),
                                              torch.empty(M, dtype=torch.float32, device=x.device))
    a = _C.W8A8GemmArgs()
    a.x, a.qweight, a.w_scale, a.out, a.xq, a.x_scale = (None if ready is not None else _p(x), _p(weight), _p(w_scale), _p(out),
                                                         _p(xq), _p(xs))
    a.M, a.K, a.N = M, K, N
    a.x_stride_m, a.out_stride_m = x.stride(0), out.stride(0)
    _C.check(lib.hymeta_w8a8_gemm_silu_mul(C.byref(a), _stream(x)), "hymeta_w8a8_gemm_silu_mul")
    return out


def w8a8_quantize(x: torch.Tensor):
    """Dynamic per-token int8 quantisation: x bf16 [M,K] -> (xq int8 [M,K], x_scale fp32 [M])."""
    _dev(x)
    M, K = x.shape
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [M, K]")
    if int8_rows_of(x) is not None:
        return int8_rows_of(x)
    xq = torch.empty(M, K, dtype=torch.int8, device=x.device)
    xs = torch.empty(M, dtype=torch.float32, device=x.device)
    _C.check(_C.load().hymeta_w8a8_quantize(_p(x), _p(xq), _p(xs), M, K, x.stride(0), _stream(x)), "hymeta_w8a8_quantize")
    return xq, xs


def w8a8_scale_out(acc: torch.Tensor, x_scale: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """acc int32 [M,N] (exact sums of int8 products) -> bf16(acc * x_scale[m] * w_scale[n] + bias[n])."""
    _dev(acc, x_scale, w_scale, bias, out)
    M, N = acc.shape
    _require(acc.dtype == torch.int32 and acc.stride(1) == 1, "acc must be row-contiguous int32 [M, N]")
    _require(x_scale.dtype == torch.float32 and x_scale.is_contiguous() and x_scale.numel() == M, "x_scale must be fp32 [M]")
    _require(w_scale.dtype == torch.float32 and w_scale.is_contiguous() and w_scale.numel() == N, "w_scale must be fp32 [N]")
    if bias is not None:
        _require(bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N, "bias must be bf16 [N]")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=acc.device)
    _require(out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be bf16 [M, N]")
    _C.check(_C.load().hymeta_w8a8_scale_out(_p(acc), _p(x_scale), _p(w_scale), _p(bias), _p(out), M, N, acc.stride(0),
                                             out.stride(0), _stream(acc)), "hymeta_w8a8_scale_out")
    return out


# ---------------------------------------------------------------------------
# Sparse mixture of experts
# ---------------------------------------------------------------------------
def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None, num_rows_dev: Optional[torch.Tensor] = None
                 ) -> torch.Tensor:
    """x bf16 [R, 2I] -> silu(x[:, :I]) * x[:, I:] bf16 [R, I]; ``num_rows_dev`` (int32 device scalar)
    limits the rows actually computed."""
    _dev(x, out, num_rows_dev)
    R, two_i = x.shape
    inter = two_i // 2
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1 and two_i % 2 == 0, "x must be row-contiguous bf16 [R, 2I]")
    if out is None:
        out = torch.empty(R, inter, dtype=torch.bfloat16, device=x.device)
    _require(out.shape == (R, inter) and out.dtype == torch.bfloat16 and out.stride(1) == 1, "out must be bf16 [R, I]")
    _C.check(_C.load().hymeta_silu_and_mul(_p(x), _p(out), _p(num_rows_dev), R, inter, x.stride(0), out.stride(0),
                                           _stream(x)), "hymeta_silu_and_mul")
    return out


def silu_and_mul_int8(x: torch.Tensor, num_rows_dev: Optional[torch.Tensor] = None):
    """x bf16 [R, 2I] -> the rows ``silu_and_mul`` would give, quantised per row as ``w8a8_quantize`` quantises them:
    (int8 [R, I], fp32 scales [R]) in one launch, the bf16 rows never written."""
    _dev(x, num_rows_dev)
    R, two_i = x.shape
    inter = two_i // 2
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1 and two_i % 2 == 0, "x must be row-contiguous bf16 [R, 2I]")
    q = torch.empty(R, inter, dtype=torch.int8, device=x.device)
    sc = torch.empty(R, dtype=torch.float32, device=x.device)
    _C.check(_C.load().hymeta_silu_and_mul_int8(_p(x), _p(q), _p(sc), _p(num_rows_dev), R, inter, x.stride(0), _stream(x)),
             "hymeta_silu_and_mul_int8")
    return q, sc


class GptqExperts(NamedTuple):
    """One projection of GPTQ experts, stacked: ``qweight`` int32 [E, K / pack, N], ``qzeros`` int32 [E, K / group, N / pack],
    ``scales`` fp16 | bf16 [E, K / group, N] (each expert's tensors as ``gptq_gemm`` takes them), contiguous."""
    qweight: torch.Tensor
    qzeros: torch.Tensor
    scales: torch.Tensor
    bits: int = 4
    group_size: int = 128
    zero_offset: int = 1

    @property
    def shape(self):
        """(E, N, K): the shape of the bf16 stack this stands for."""
        return (self.qweight.shape[0], self.qweight.shape[2], self.qweight.shape[1] * (32 // self.bits))


# Rows per local expert (on average) up to which GPTQ experts run the fused de-quantising grouped GEMM; above, the local
# experts are expanded to bf16 once per call and the tiled bf16 grouped GEMM runs on the expansion (the fused kernel
# re-reads an expert's weights per 32-row tile).  Measured on the 70B TP=4 shapes (16 experts, us per block, fused /
# expanded): 1024 tokens 466 / 796, 2048: 777 / 943, 4096: 1373 / 1164, 8192: 2569 / 1557.
MOE_GPTQ_FUSED_ROWS_PER_EXPERT = int(os.environ.get("HYMETA_MOE_GPTQ_FUSED_ROWS", "160"))


def moe_gptq_expand(w: GptqExperts) -> torch.Tensor:
    """GPTQ expert stack -> bf16 [E, N, K] (one launch), in a grow-only workspace of the device and stream."""
    E, N, K = w.shape
    pack = 32 // w.bits
    _require(w.qweight.dtype == torch.int32 and w.qweight.is_contiguous() and w.qzeros.is_contiguous() and w.scales.is_contiguous(),
             "GPTQ expert tensors must be contiguous")
    _require(N % 64 == 0 and K % (16 * pack) == 0, "expansion needs N % 64 == 0 and K % (16 * pack) == 0")
    ws = _workspace(w.qweight.device, E * N * K * 2, f"moe_gptq_expand_{N}x{K}")
    out = ws[: E * N * K * 2].view(torch.bfloat16).view(E, N, K)
    sd = _C.HYMETA_F16 if w.scales.dtype == torch.float16 else _C.HYMETA_BF16
    _C.check(_C.load().hymeta_moe_gptq_expand(_p(w.qweight), _p(w.qzeros), _p(w.scales), _p(out), E, K, N, w.bits, w.group_size,
                                              sd, w.zero_offset, _stream(w.qweight)), "hymeta_moe_gptq_expand")
    return out


# Switch for A/B timing and for tests of both forms: fold the combine into the experts' down GEMM at decode sizes (see
# moe_forward).
MOE_FUSE_DOWN = os.environ.get("HYMETA_MOE_FUSE_DOWN", "1") != "0"


def moe_router_fused(x: torch.Tensor, gate_weight: torch.Tensor) -> bool:
    """Whether ``moe_forward`` routes this batch with the one-launch router (linear layer + routing)."""
    per16 = 16 if gate_weight.dtype == torch.int8 else 8      # an int8 router reads the tokens' int8 rows
    return (x.is_cuda and x.dtype == torch.bfloat16 and gate_weight.dtype in (torch.bfloat16, torch.int8)
            and gate_weight.is_contiguous() and 0 < x.shape[0] <= _C.MOE_ROUTER_MAX_TOKENS and x.shape[1] % per16 == 0
            and x.stride(0) % 8 == 0 and gate_weight.shape[0] <= 64)


def moe_forward(x: torch.Tensor, router_logits: Optional[torch.Tensor], w13: torch.Tensor, w2: torch.Tensor, top_k: int = 1,
                renormalize: bool = False, shared: Optional[torch.Tensor] = None,
                local_experts: Optional[tuple] = None, out: Optional[torch.Tensor] = None,
                gate_weight: Optional[torch.Tensor] = None, logits_out: Optional[torch.Tensor] = None,
                w13_scale: Optional[torch.Tensor] = None, w2_scale: Optional[torch.Tensor] = None,
                gate_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sparse MoE block: x bf16 [T,H], router_logits [T,E] (bf16 | fp32), w13 bf16 [E,2I,H], w2 bf16
    [E,H,I] -- or W8A8 experts: w13 / w2 int8 with per-output-channel ``w13_scale`` fp32 [E,2I] / ``w2_scale`` fp32 [E,H]
    (the tokens and the activated rows are quantised per row on the way in: exact int32 sums, scales applied once);
    ``shared`` bf16 [T,H] is added to the result; ``local_experts = (lo, hi)`` computes only
    this rank's experts (the caller all-reduces); the weight tensors may then hold just those experts
    (``[hi - lo, 2I, H]`` / ``[hi - lo, H, I]``, expert ``lo`` first: an expert-parallel rank keeps nothing else), with the
    number of experts taken from the router (``gate_weight`` rows / ``router_logits`` columns).  Returns bf16 [T,H].
    With ``router_logits=None`` and ``gate_weight`` bf16 [E,H] the block computes the logits itself: decode-sized batches
    in the routing launch (``moe_router_fused``; ``logits_out`` bf16 [T,E] then receives them), larger ones through the
    linear layer.  An int8 ``gate_weight`` with ``gate_scale`` fp32 [E] (the router of a W8A8 checkpoint) is routed in
    that launch too, on the tokens' int8 rows; larger batches must come with ``router_logits``.  GPTQ experts: ``w13`` /
    ``w2`` as ``GptqExperts`` stacks -- decode-sized batches run the fused de-quantising grouped GEMM, larger ones expand
    the local experts to bf16 once per call."""
    gptq = isinstance(w13, GptqExperts)
    if gptq:
        _require(isinstance(w2, GptqExperts), "w13 and w2 must both be GPTQ stacks")
        n_local = (local_experts[1] - local_experts[0]) if local_experts is not None else w13.shape[0]
        if x.shape[0] * top_k > MOE_GPTQ_FUSED_ROWS_PER_EXPERT * max(n_local, 1):
            held = w13.shape[0]
            total = gate_weight.shape[0] if gate_weight is not None else router_logits.shape[1]
            if held == total and local_experts is not None and tuple(local_experts) != (0, total):
                lo_, hi_ = local_experts      # expand the rank's experts only
                w13 = w13._replace(qweight=w13.qweight[lo_:hi_], qzeros=w13.qzeros[lo_:hi_], scales=w13.scales[lo_:hi_])
                w2 = w2._replace(qweight=w2.qweight[lo_:hi_], qzeros=w2.qzeros[lo_:hi_], scales=w2.scales[lo_:hi_])
            return moe_forward(x, router_logits, moe_gptq_expand(w13), moe_gptq_expand(w2), top_k, renormalize, shared,
                               local_experts, out, gate_weight, logits_out)
        _dev(x, router_logits, w13.qweight, w2.qweight, shared, out, gate_weight, gate_scale)
    else:
        _dev(x, router_logits, w13, w2, shared, out, gate_weight, gate_scale)
    T, H = x.shape
    E_held, two_i, Hw = w13.shape
    inter = two_i // 2
    _require(router_logits is not None or gate_weight is not None, "router_logits or gate_weight must be given")
    E = gate_weight.shape[0] if gate_weight is not None else router_logits.shape[1]
    lo, hi = local_experts if local_experts is not None else (0, E)
    _require(0 <= lo <= hi <= E, "local_experts must lie inside the router's experts")
    compact = E_held != E                   # only the local experts are held: slot 0 = expert lo
    _require(E_held == E or E_held == hi - lo, "w13 / w2 must hold every expert or exactly the local ones")
    fused_router = router_logits is None and T > 0 and moe_router_fused(x, gate_weight)
    if gate_weight is not None and gate_weight.dtype == torch.int8 and router_logits is None:
        _require(fused_router, "an int8 router above the one-launch router's batch size needs router_logits")
        _require(gate_scale is not None and gate_scale.dtype == torch.float32 and gate_scale.is_contiguous()
                 and gate_scale.numel() == E, "gate_scale must be contiguous fp32 [E]")
    if router_logits is None and not fused_router:
        _require(gate_weight.shape == (E, H), "gate_weight must be [E, H]")
        router_logits = torch.nn.functional.linear(x, gate_weight)
        if logits_out is not None:
            logits_out.copy_(router_logits)
    _require(x.dtype == torch.bfloat16 and x.stride(1) == 1, "x must be row-contiguous bf16 [T, H]")
    i8 = not gptq and w13.dtype == torch.int8
    if gptq:
        _require(Hw == H and tuple(w2.shape) == (E_held, H, inter), "GPTQ stacks must stand for w13 [E, 2I, H] and w2 [E, H, I]")
        for w in (w13, w2):
            pk = 32 // w.bits
            _require(w.qweight.dtype == torch.int32 and w.qweight.is_contiguous() and w.qzeros.dtype == torch.int32
                     and w.qzeros.is_contiguous() and w.scales.dtype in (torch.float16, torch.bfloat16) and w.scales.is_contiguous(),
                     "GPTQ expert tensors must be contiguous int32 / int32 / fp16 | bf16")
            _require(tuple(w.qzeros.shape) == (E_held, w.shape[2] // w.group_size, w.shape[1] // pk)
                     and tuple(w.scales.shape) == (E_held, w.shape[2] // w.group_size, w.shape[1]),
                     "qzeros / scales must be [E, K / group, N / pack] / [E, K / group, N]")
    else:
        _require(w13.dtype in (torch.bfloat16, torch.int8) and w13.is_contiguous() and Hw == H,
                 "w13 must be contiguous bf16 | int8 [E, 2I, H]")
        _require(w2.dtype == w13.dtype and w2.is_contiguous() and w2.shape == (E_held, H, inter),
                 "w2 must be contiguous [E, H, I] of w13's type")
    if i8:
        _dev(w13_scale, w2_scale)
        _require(w13_scale is not None and w13_scale.dtype == torch.float32 and w13_scale.is_contiguous()
                 and w13_scale.shape == (E_held, two_i), "w13_scale must be contiguous fp32 [E, 2I]")
        _require(w2_scale is not None and w2_scale.dtype == torch.float32 and w2_scale.is_contiguous()
                 and w2_scale.shape == (E_held, H), "w2_scale must be contiguous fp32 [E, H]")
        _require(H % 128 == 0 and inter % 64 == 0, "int8 experts need H % 128 == 0 and I % 64 == 0")
    _require(fused_router or (router_logits.shape == (T, E) and router_logits.dtype in _DT and router_logits.stride(1) == 1),
             "router_logits must be [T, E] bf16 | fp32")
    dev = x.device
    P = T * top_k
    ids = torch.empty(T, top_k, dtype=torch.int32, device=dev)
    wts = torch.empty(T, top_k, dtype=torch.float32, device=dev)
    meta = torch.empty(3 * (E + 1), dtype=torch.int32, device=dev)
    counts, cursors, offsets = meta[: E + 1], meta[E + 1: 2 * (E + 1)], meta[2 * (E + 1):]
    sorted_pair = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    pair_pos = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    if out is None:
        out = torch.empty(T, H, dtype=torch.bfloat16, device=dev)
    if T == 0:
        return out
    lib, st = _C.load(), _stream(x)
    if fused_router:
        _require(gate_weight.shape == (E, H), "gate_weight must be [E, H]")
        _require(logits_out is None or (logits_out.shape == (T, E) and logits_out.dtype == torch.bfloat16 and logits_out.is_contiguous()),
                 "logits_out must be contiguous bf16 [T, E]")
        r = _C.MoeRouterArgs()
        r.x, r.gate_weight, r.logits_out = _p(x), _p(gate_weight), _p(logits_out)
        r.x_stride_t = x.stride(0)
        if gate_weight.dtype == torch.int8:
            xq_r, xs_r = w8a8_quantize(x)       # (the rows a quantising norm attached, or one launch; the experts reuse them)
            setattr(x, INT8_ROWS_ATTR, (xq_r, xs_r))
            r.x, r.x_stride_t = _p(xq_r), xq_r.stride(0)
            r.operand_dtype, r.x_scale, r.gate_scale = _C.HYMETA_I8, _p(xs_r), _p(gate_scale)
        r.topk_ids, r.topk_weights, r.expert_offsets = _p(ids), _p(wts), _p(offsets)
        r.sorted_pair, r.pair_pos = _p(sorted_pair), _p(pair_pos)
        r.ticket = _p(_workspace(dev, 64, "moe_router"))
        r.num_tokens, r.num_experts, r.top_k, r.renormalize, r.hidden = T, E, top_k, int(bool(renormalize)), H
        r.local_expert_begin, r.local_expert_end = lo, hi
        _C.check(lib.hymeta_moe_router(C.byref(r), st), "hymeta_moe_router")
    else:
        r = _C.MoeRouteArgs()
        r.router_logits, r.topk_ids, r.topk_weights = _p(router_logits), _p(ids), _p(wts)
        r.counts, r.cursors, r.expert_offsets = _p(counts), _p(cursors), _p(offsets)
        r.sorted_pair, r.pair_pos = _p(sorted_pair), _p(pair_pos)
        r.num_tokens, r.num_experts, r.top_k, r.renormalize = T, E, top_k, int(bool(renormalize))
        r.logits_dtype = _DT[router_logits.dtype]
        r.local_expert_begin, r.local_expert_end = lo, hi
        r.logits_stride_t = router_logits.stride(0)
        _C.check(lib.hymeta_moe_route(C.byref(r), st), "hymeta_moe_route")

    h13 = torch.empty(P, two_i, dtype=torch.bfloat16, device=dev)
    if gptq:
        if shared is not None:
            _require(shared.shape == (T, H) and shared.dtype == torch.bfloat16 and shared.stride(1) == 1,
                     "shared must be bf16 [T, H]")

        def gptq_gemm_args(w: GptqExperts, xin, xout, K_, N_):
            q = _C.MoeGptqGemmArgs()
            back = lo if compact else 0           # tensors that hold the local experts only: address expert 0
            per_q, per_z, per_s = w.qweight.stride(0) * 4, w.qzeros.stride(0) * 4, w.scales.stride(0) * 2
            q.qweight = C.c_void_p(w.qweight.data_ptr() - back * per_q)
            q.qzeros = C.c_void_p(w.qzeros.data_ptr() - back * per_z)
            q.scales = C.c_void_p(w.scales.data_ptr() - back * per_s)
            q.x, q.out, q.expert_offsets = _p(xin), _p(xout), _p(offsets)
            q.max_rows, q.K, q.N = P, K_, N_
            q.bits, q.group_size, q.zero_offset = w.bits, w.group_size, w.zero_offset
            q.scales_dtype = _C.HYMETA_F16 if w.scales.dtype == torch.float16 else _C.HYMETA_BF16
            q.local_expert_begin, q.local_expert_end = lo, hi
            q.x_stride_m, q.out_stride_m = xin.stride(0), xout.stride(0)
            return q

        q13 = gptq_gemm_args(w13, x, h13, H, two_i)
        q13.a_rows, q13.a_rows_div = _p(sorted_pair), top_k
        _C.check(lib.hymeta_moe_gptq_gemm(C.byref(q13), st), "hymeta_moe_gptq_gemm (w13)")
        hact = torch.empty(P, inter, dtype=torch.bfloat16, device=dev)
        silu_and_mul(h13, out=hact)
        y = torch.empty(P, H, dtype=torch.bfloat16, device=dev)
        q2 = gptq_gemm_args(w2, hact, y, inter, H)
        q2.a_rows_div, q2.scale_rows, q2.row_scale = 1, _p(sorted_pair), _p(wts)
        _C.check(lib.hymeta_moe_gptq_gemm(C.byref(q2), st), "hymeta_moe_gptq_gemm (w2)")
        c = _C.MoeCombineArgs()
        c.y, c.shared, c.out, c.pair_pos, c.topk_weights = _p(y), _p(shared), _p(out), _p(pair_pos), _p(wts)
        c.num_tokens, c.hidden, c.top_k = T, H, top_k
        c.y_stride_m, c.out_stride_m = y.stride(0), out.stride(0)
        if shared is not None:
            c.shared_stride_m = shared.stride(0)
        _C.check(lib.hymeta_moe_combine(C.byref(c), st), "hymeta_moe_combine")
        return out
    g = _C.MoeGemmArgs()
    # the kernels index the weights by the expert's number: tensors that hold the local experts only are addressed from
    # where expert 0 would lie (only experts lo .. hi - 1 are ever read)
    esize = 1 if i8 else 2
    w13_base = C.c_void_p(w13.data_ptr() - (lo * w13.stride(0) * esize if compact else 0))
    w2_base = C.c_void_p(w2.data_ptr() - (lo * w2.stride(0) * esize if compact else 0))
    xin = x
    if i8:
        xin, xs = w8a8_quantize(x)              # per token, once for all its experts
        g.operand_dtype, g.x_scale = _C.HYMETA_I8, _p(xs)
        g.w_scale = C.c_void_p(w13_scale.data_ptr() - (lo * two_i * 4 if compact else 0))
    g.x, g.weights, g.out, g.expert_offsets = _p(xin), w13_base, _p(h13), _p(offsets)
    g.a_rows, g.a_rows_div, g.max_rows, g.K, g.N = _p(sorted_pair), top_k, P, H, two_i
    g.local_expert_begin, g.local_expert_end = lo, hi
    g.x_stride_m, g.out_stride_m, g.w_stride_e = xin.stride(0), h13.stride(0), w13.stride(0)
    _C.check(lib.hymeta_moe_gemm(C.byref(g), st), "hymeta_moe_gemm (w13)")

    if shared is not None:
        _require(shared.shape == (T, H) and shared.dtype == torch.bfloat16 and shared.stride(1) == 1,
                 "shared must be bf16 [T, H]")
    # decode-sized batches with one choice per token and every expert local: the down GEMM writes its rows straight to
    # their tokens with the shared expert added -- no combine launch
    fuse_combine = (MOE_FUSE_DOWN and top_k == 1 and (lo, hi) == (0, E)
                    and bool(lib.hymeta_moe_gemm_fuses(P, hi - lo, inter // 2 if i8 else inter, H))
                    and (shared is None or shared.stride(0) % 8 == 0))
    # slots in use are a prefix only when lo == 0; otherwise act on all P rows (unused rows are never read)
    d = _C.MoeGemmArgs()
    if i8:
        hact, hs = silu_and_mul_int8(h13)       # activation and per-row quantisation in one launch
        d.operand_dtype, d.x_scale = _C.HYMETA_I8, _p(hs)
        d.w_scale = C.c_void_p(w2_scale.data_ptr() - (lo * H * 4 if compact else 0))
    else:
        hact = torch.empty(P, inter, dtype=torch.bfloat16, device=dev)
        silu_and_mul(h13, out=hact)
    d.x, d.x_stride_m = _p(hact), hact.stride(0)
    y = out if fuse_combine else torch.empty(P, H, dtype=torch.bfloat16, device=dev)
    d.weights, d.out, d.expert_offsets = w2_base, _p(y), _p(offsets)
    d.scale_rows, d.row_scale = _p(sorted_pair), _p(wts)
    d.a_rows_div, d.max_rows, d.K, d.N = 1, P, inter, H
    d.local_expert_begin, d.local_expert_end = lo, hi
    d.out_stride_m, d.w_stride_e = y.stride(0), w2.stride(0)
    if fuse_combine:
        d.combine_tokens = _p(sorted_pair)       # one choice per token: the pair index is the token
        if shared is not None:
            d.combine_shared, d.combine_shared_stride_m = _p(shared), shared.stride(0)
    _C.check(lib.hymeta_moe_gemm(C.byref(d), st), "hymeta_moe_gemm (w2)")
    if fuse_combine:
        return out

    c = _C.MoeCombineArgs()
    c.y, c.shared, c.out, c.pair_pos, c.topk_weights = _p(y), _p(shared), _p(out), _p(pair_pos), _p(wts)
    c.num_tokens, c.hidden, c.top_k = T, H, top_k
    c.y_stride_m, c.out_stride_m = y.stride(0), out.stride(0)
    if shared is not None:
        c.shared_stride_m = shared.stride(0)
    _C.check(lib.hymeta_moe_combine(C.byref(c), st), "hymeta_moe_combine")
    return out
