"""ctypes binding of libhymeta_hip.so (declarations: include/hymeta_hip.h).

The library is the product path: if it is missing or a symbol cannot be bound,
``load()`` raises -- there is no fallback implementation behind these ops.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB_NAME = "libhymeta_hip.so"
_lib: Optional[C.CDLL] = None

HYMETA_OK = 0
HYMETA_BF16 = 0
HYMETA_F32 = 1
HYMETA_F16 = 2
HYMETA_I8 = 3

GATE_NONE = 0
GATE_7B = 1
GATE_70B_CLAMP = 2
GATE_70B_LOWER_BOUND = 3

ABI_VERSION = 3
ERR_SCRATCH_LIMIT = -5
GEMM_TICKET_BYTES = 16384
STATIC_SCRATCH_BYTES = 16 << 20

_i32, _i64, _f32, _ptr, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
_SIZE = ("struct_size", C.c_uint32)


class _Args(C.Structure):
    """An argument struct of the library: the leading ``struct_size`` is filled in on construction with the size of
    this declaration, which every entry point compares with its own."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if not self.struct_size:
            self.struct_size = C.sizeof(type(self))


class GlaDecodeArgs(_Args):
    _fields_ = [
        _SIZE,
        ("q", _ptr), ("k", _ptr), ("v", _ptr), ("g", _ptr), ("out", _ptr), ("state", _ptr),
        ("slot_idx", _ptr), ("lower_bound", _ptr), ("out_norm_weight", _ptr),
        ("batch", _i32), ("num_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
        ("q_stride_b", _i64), ("q_stride_h", _i64),
        ("k_stride_b", _i64), ("k_stride_h", _i64),
        ("v_stride_b", _i64), ("v_stride_h", _i64),
        ("g_stride_b", _i64), ("g_stride_h", _i64),
        ("out_stride_b", _i64), ("out_stride_h", _i64),
        ("state_stride_slot", _i64), ("state_stride_h", _i64),
        ("g_dtype", _i32), ("state_dtype", _i32), ("gate_mode", _i32),
        ("scale", _f32), ("gate_normalizer", _f32), ("gate_clamp", _f32), ("out_norm_eps", _f32),
        ("state_heads", _i32), ("g_w1", _ptr), ("g_bias", _ptr),
        ("qkv_slabs", _ptr), ("qkv_bias", _ptr), ("qkv_splits", _i32), ("qkv_n", _i32),
        ("q_col", _i32), ("k_col", _i32), ("v_col", _i32), ("g_col", _i32),
        ("g_w0", _ptr), ("g_hidden", _i32),
        ("qkv_x_scale", _ptr), ("qkv_w_scale", _ptr),
    ]


class GlaPrefillArgs(_Args):
    _fields_ = [
        _SIZE,
        ("q", _ptr), ("k", _ptr), ("v", _ptr), ("g", _ptr), ("out", _ptr), ("state", _ptr),
        ("slot_idx", _ptr), ("cu_seqlens", _ptr), ("context_lens", _ptr), ("meta_state", _ptr),
        ("lower_bound", _ptr), ("out_norm_weight", _ptr),
        ("num_seqs", _i32), ("total_tokens", _i32), ("max_seqlen", _i32),
        ("num_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
        ("q_stride_t", _i64), ("q_stride_h", _i64),
        ("k_stride_t", _i64), ("k_stride_h", _i64),
        ("v_stride_t", _i64), ("v_stride_h", _i64),
        ("g_stride_t", _i64), ("g_stride_h", _i64),
        ("out_stride_t", _i64), ("out_stride_h", _i64),
        ("state_stride_slot", _i64), ("state_stride_h", _i64),
        ("meta_state_stride_h", _i64),
        ("g_dtype", _i32), ("state_dtype", _i32), ("gate_mode", _i32),
        ("scale", _f32), ("gate_normalizer", _f32), ("gate_clamp", _f32), ("out_norm_eps", _f32),
        ("workspace", _ptr), ("workspace_bytes", _sz), ("num_segments", _i32),
        ("state_heads", _i32),
    ]


class ReshapeAndCacheArgs(_Args):
    _fields_ = [
        _SIZE,
        ("key", _ptr), ("value", _ptr), ("key_cache", _ptr), ("value_cache", _ptr),
        ("slot_mapping", _ptr),
        ("num_tokens", _i32), ("num_kv_heads", _i32), ("head_dim", _i32), ("block_size", _i32),
        ("elem_bytes", _i32),
        ("num_blocks", _i64),
        ("key_stride_t", _i64), ("key_stride_h", _i64),
        ("value_stride_t", _i64), ("value_stride_h", _i64),
        ("cache_stride_block", _i64), ("cache_stride_pos", _i64), ("cache_stride_h", _i64),
    ]


class PagedDecodeArgs(_Args):
    _fields_ = [
        _SIZE,
        ("q", _ptr), ("out", _ptr), ("key_cache", _ptr), ("value_cache", _ptr),
        ("block_tables", _ptr), ("cache_seqlens", _ptr), ("meta_key", _ptr), ("meta_value", _ptr),
        ("workspace", _ptr), ("workspace_bytes", _sz),
        ("batch", _i32), ("num_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
        ("block_size", _i32), ("max_blocks_per_seq", _i32), ("num_meta", _i32),
        ("window_left", _i32), ("max_seqlen_k", _i32), ("num_splits", _i32),
        ("q_stride_b", _i64), ("q_stride_h", _i64),
        ("out_stride_b", _i64), ("out_stride_h", _i64),
        ("cache_stride_block", _i64), ("cache_stride_pos", _i64), ("cache_stride_h", _i64),
        ("block_table_stride", _i64),
        ("meta_stride_t", _i64), ("meta_stride_h", _i64),
        ("scale", _f32),
        ("new_key", _ptr), ("new_value", _ptr), ("slot_mapping", _ptr),
        ("new_key_stride_b", _i64), ("new_key_stride_h", _i64),
        ("new_value_stride_b", _i64), ("new_value_stride_h", _i64),
    ]


class SwaPrefillArgs(_Args):
    _fields_ = [
        _SIZE,
        ("q", _ptr), ("key", _ptr), ("value", _ptr), ("key_cache", _ptr), ("value_cache", _ptr),
        ("block_tables", _ptr), ("cu_seqlens_q", _ptr), ("cu_seqlens_k", _ptr), ("seq_lens_k", _ptr),
        ("meta_query", _ptr), ("meta_key", _ptr), ("meta_value", _ptr), ("out", _ptr),
        ("num_seqs", _i32), ("total_q", _i32), ("max_seqlen_q", _i32),
        ("num_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
        ("num_meta", _i32), ("window_left", _i32), ("block_size", _i32), ("max_blocks_per_seq", _i32),
        ("q_stride_t", _i64), ("q_stride_h", _i64),
        ("k_stride_t", _i64), ("k_stride_h", _i64),
        ("v_stride_t", _i64), ("v_stride_h", _i64),
        ("out_stride_t", _i64), ("out_stride_h", _i64),
        ("cache_stride_block", _i64), ("cache_stride_pos", _i64), ("cache_stride_h", _i64),
        ("block_table_stride", _i64),
        ("meta_q_stride_t", _i64), ("meta_q_stride_h", _i64),
        ("meta_stride_t", _i64), ("meta_stride_h", _i64),
        ("scale", _f32),
        ("workspace", _ptr), ("workspace_bytes", _sz), ("num_splits", _i32), ("max_seqlen_k", _i32),
    ]


class RopeArgs(_Args):
    _fields_ = [
        _SIZE,
        ("q", _ptr), ("k", _ptr), ("positions", _ptr), ("cos_sin_cache", _ptr),
        ("num_tokens", _i32), ("num_heads", _i32), ("num_kv_heads", _i32), ("head_dim", _i32),
        ("rot_dim", _i32), ("max_position", _i32),
        ("position_offset", _i64),
        ("q_stride_t", _i64), ("q_stride_h", _i64), ("k_stride_t", _i64), ("k_stride_h", _i64),
    ]


class CombineArgs(_Args):
    _fields_ = [
        _SIZE,
        ("o_attn", _ptr), ("o_lin", _ptr), ("weight1", _ptr), ("weight2", _ptr), ("stats", _ptr),
        ("out", _ptr),
        ("num_tokens", _i32), ("hidden_local", _i32),
        ("attn_stride_t", _i64), ("lin_stride_t", _i64), ("out_stride_t", _i64),
        ("eps", _f32),
    ]


class AddRmsNormArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("residual", _ptr), ("weight", _ptr), ("out", _ptr),
        ("num_tokens", _i32), ("hidden", _i32),
        ("x_stride_t", _i64), ("residual_stride_t", _i64), ("out_stride_t", _i64),
        ("eps", _f32),
    ]


class LowrankGateArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("w0", _ptr), ("w1", _ptr), ("bias", _ptr), ("out", _ptr),
        ("num_tokens", _i32), ("hidden", _i32), ("rank", _i32), ("out_features", _i32),
        ("x_stride_t", _i64), ("out_stride_t", _i64),
    ]


class GptqGemmArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("qweight", _ptr), ("qzeros", _ptr), ("scales", _ptr), ("bias", _ptr), ("out", _ptr),
        ("workspace", _ptr), ("workspace_bytes", _sz),
        ("M", _i32), ("K", _i32), ("N", _i32), ("bits", _i32), ("group_size", _i32),
        ("scales_dtype", _i32), ("zero_offset", _i32), ("split_k", _i32),
        ("x_stride_m", _i64), ("out_stride_m", _i64),
    ]


class Bf16GemmArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("weight", _ptr), ("bias", _ptr), ("out", _ptr), ("workspace", _ptr), ("workspace_bytes", _sz),
        ("M", _i32), ("K", _i32), ("N", _i32), ("split_k", _i32),
        ("x_stride_m", _i64), ("w_stride_n", _i64), ("out_stride_m", _i64),
    ]


class W8A8GemmArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("qweight", _ptr), ("w_scale", _ptr), ("bias", _ptr), ("out", _ptr),
        ("xq", _ptr), ("x_scale", _ptr),
        ("M", _i32), ("K", _i32), ("N", _i32),
        ("x_stride_m", _i64), ("out_stride_m", _i64),
        ("workspace", _ptr), ("workspace_bytes", _sz), ("split_k", _i32),
    ]


class GemmPartials(_Args):
    _fields_ = [
        _SIZE,
        ("slabs", _ptr), ("splits", _i32), ("is_int32", _i32), ("x_scale", _ptr), ("w_scale", _ptr), ("bias", _ptr),
        ("M", _i32), ("N", _i32),
    ]


class MoeRouteArgs(_Args):
    _fields_ = [
        _SIZE,
        ("router_logits", _ptr), ("topk_ids", _ptr), ("topk_weights", _ptr), ("counts", _ptr), ("cursors", _ptr),
        ("expert_offsets", _ptr), ("sorted_pair", _ptr), ("pair_pos", _ptr),
        ("num_tokens", _i32), ("num_experts", _i32), ("top_k", _i32), ("renormalize", _i32), ("logits_dtype", _i32),
        ("local_expert_begin", _i32), ("local_expert_end", _i32),
        ("logits_stride_t", _i64),
    ]


class MoeRouterArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("gate_weight", _ptr), ("logits_out", _ptr), ("topk_ids", _ptr), ("topk_weights", _ptr),
        ("expert_offsets", _ptr), ("sorted_pair", _ptr), ("pair_pos", _ptr), ("ticket", _ptr),
        ("num_tokens", _i32), ("num_experts", _i32), ("top_k", _i32), ("renormalize", _i32), ("hidden", _i32),
        ("local_expert_begin", _i32), ("local_expert_end", _i32),
        ("x_stride_t", _i64),
        ("operand_dtype", _i32), ("x_scale", _ptr), ("gate_scale", _ptr),
    ]


MOE_ROUTER_MAX_TOKENS = 256


class MoeGemmArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("weights", _ptr), ("out", _ptr), ("expert_offsets", _ptr), ("a_rows", _ptr), ("scale_rows", _ptr),
        ("row_scale", _ptr),
        ("a_rows_div", _i32), ("max_rows", _i32), ("K", _i32), ("N", _i32),
        ("local_expert_begin", _i32), ("local_expert_end", _i32),
        ("x_stride_m", _i64), ("out_stride_m", _i64), ("w_stride_e", _i64),
        ("combine_tokens", _ptr), ("combine_shared", _ptr), ("combine_shared_stride_m", _i64),
        ("operand_dtype", _i32), ("x_scale", _ptr), ("w_scale", _ptr),
    ]


class MoeGptqGemmArgs(_Args):
    _fields_ = [
        _SIZE,
        ("x", _ptr), ("qweight", _ptr), ("qzeros", _ptr), ("scales", _ptr), ("out", _ptr), ("expert_offsets", _ptr),
        ("a_rows", _ptr), ("scale_rows", _ptr), ("row_scale", _ptr),
        ("a_rows_div", _i32), ("max_rows", _i32), ("K", _i32), ("N", _i32),
        ("bits", _i32), ("group_size", _i32), ("scales_dtype", _i32), ("zero_offset", _i32),
        ("local_expert_begin", _i32), ("local_expert_end", _i32),
        ("x_stride_m", _i64), ("out_stride_m", _i64),
    ]


class MoeCombineArgs(_Args):
    _fields_ = [
        _SIZE,
        ("y", _ptr), ("shared", _ptr), ("out", _ptr), ("pair_pos", _ptr), ("topk_weights", _ptr),
        ("num_tokens", _i32), ("hidden", _i32), ("top_k", _i32),
        ("y_stride_m", _i64), ("shared_stride_m", _i64), ("out_stride_m", _i64),
    ]


# symbol -> (restype, argtypes); every function include/hymeta_hip.h declares.
_SIGNATURES = {
    "hymeta_strerror": (C.c_char_p, [C.c_int]),
    "hymeta_abi_version": (C.c_int, []),
    "hymeta_stream_capture_id": (C.c_int, [_ptr, C.POINTER(C.c_uint64)]),
    "hymeta_gla_decode": (C.c_int, [C.POINTER(GlaDecodeArgs), _ptr]),
    "hymeta_gla_prefill_varlen": (C.c_int, [C.POINTER(GlaPrefillArgs), _ptr]),
    "hymeta_gla_prefill_segments": (C.c_int, [_i32, _i32, _i32]),
    "hymeta_gla_prefill_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "hymeta_reshape_and_cache": (C.c_int, [C.POINTER(ReshapeAndCacheArgs), _ptr]),
    "hymeta_meta_swa_paged_decode_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "hymeta_meta_swa_paged_decode_splits": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "hymeta_meta_swa_paged_decode": (C.c_int, [C.POINTER(PagedDecodeArgs), _ptr]),
    "hymeta_meta_swa_paged_decode_static": (C.c_int, [C.POINTER(PagedDecodeArgs), _ptr]),
    "hymeta_meta_swa_prefill": (C.c_int, [C.POINTER(SwaPrefillArgs), _ptr]),
    "hymeta_meta_swa_prefill_splits": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "hymeta_meta_swa_prefill_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "hymeta_rope_neox": (C.c_int, [C.POINTER(RopeArgs), _ptr]),
    "hymeta_rms_sumsq2": (C.c_int, [C.POINTER(CombineArgs), _ptr]),
    "hymeta_hybrid_combine": (C.c_int, [C.POINTER(CombineArgs), _ptr]),
    "hymeta_add_rmsnorm": (C.c_int, [C.POINTER(AddRmsNormArgs), _ptr]),
    "hymeta_lowrank_gate": (C.c_int, [C.POINTER(LowrankGateArgs), _ptr]),
    "hymeta_gptq_gemm_split_k": (C.c_int, [_i32, _i32, _i32, _i32]),
    "hymeta_gptq_gemm_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "hymeta_gptq_gemm": (C.c_int, [C.POINTER(GptqGemmArgs), _ptr]),
    "hymeta_gptq_dequant": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "hymeta_gptq_dequant_nk": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "hymeta_bf16_gemm": (C.c_int, [C.POINTER(Bf16GemmArgs), _ptr]),
    "hymeta_bf16_gemm_silu_mul": (C.c_int, [C.POINTER(Bf16GemmArgs), _ptr]),
    "hymeta_bf16_gemm_f32out": (C.c_int, [C.POINTER(Bf16GemmArgs), _ptr]),
    "hymeta_argmax_rows": (C.c_int, [_ptr, _i64, _i32, _i32, _ptr, _ptr, _sz, _ptr]),
    "hymeta_argmax_rows_workspace_bytes": (_sz, [_i32]),
    "hymeta_advance_step": (C.c_int, [_i32, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr]),
    "hymeta_p2p_create": (C.c_int, [C.c_int, C.c_int, _sz, C.POINTER(_ptr), _ptr]),
    "hymeta_p2p_local_buffer": (_ptr, [_ptr]),
    "hymeta_p2p_connect": (C.c_int, [_ptr, _ptr, _ptr]),
    "hymeta_p2p_verify": (C.c_int, [_ptr]),
    "hymeta_p2p_connect_pointers": (C.c_int, [_ptr, C.POINTER(_ptr)]),
    "hymeta_p2p_set_wait_limit": (C.c_int, [_ptr, C.c_uint64]),
    "hymeta_p2p_all_reduce": (C.c_int, [_ptr, _ptr, _i64, _ptr, _i64, _f32, _ptr]),
    "hymeta_p2p_all_reduce_partials": (C.c_int, [_ptr, C.POINTER(GemmPartials), _ptr, _ptr]),
    "hymeta_p2p_timeouts": (C.c_int, [_ptr, C.POINTER(C.c_uint32)]),
    "hymeta_p2p_destroy": (C.c_int, [_ptr]),
    "hymeta_bf16_gemm_deferred": (C.c_int, [C.POINTER(Bf16GemmArgs), C.POINTER(GemmPartials), _ptr]),
    "hymeta_w8a8_gemm_deferred": (C.c_int, [C.POINTER(W8A8GemmArgs), C.POINTER(GemmPartials), _ptr]),
    "hymeta_gptq_gemm_deferred": (C.c_int, [C.POINTER(GptqGemmArgs), C.POINTER(GemmPartials), _ptr]),
    "hymeta_gemm_partials_reduce": (C.c_int, [C.POINTER(GemmPartials), _ptr, _i64, _ptr]),
    "hymeta_add_rmsnorm_partials": (C.c_int, [C.POINTER(AddRmsNormArgs), C.POINTER(GemmPartials), _ptr]),
    "hymeta_rope_neox_partials": (C.c_int, [C.POINTER(RopeArgs), C.POINTER(GemmPartials), _ptr]),
    "hymeta_add_rmsnorm_int8": (C.c_int, [C.POINTER(AddRmsNormArgs), C.POINTER(GemmPartials), _ptr, _ptr, _ptr]),
    "hymeta_bf16_gemm_split_k": (C.c_int, [_i32, _i32, _i32]),
    "hymeta_bf16_gemm_takes": (C.c_int, [_i32, _i32, _i32, _i32]),
    "hymeta_bf16_gemm_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "hymeta_w8a8_gemm": (C.c_int, [C.POINTER(W8A8GemmArgs), _ptr]),
    "hymeta_w8a8_gemm_silu_mul": (C.c_int, [C.POINTER(W8A8GemmArgs), _ptr]),
    "hymeta_moe_route": (C.c_int, [C.POINTER(MoeRouteArgs), _ptr]),
    "hymeta_moe_router": (C.c_int, [C.POINTER(MoeRouterArgs), _ptr]),
    "hymeta_moe_gemm_fuses": (C.c_int, [_i32, _i32, _i32, _i32]),
    "hymeta_moe_gemm": (C.c_int, [C.POINTER(MoeGemmArgs), _ptr]),
    "hymeta_silu_and_mul": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i64, _i64, _ptr]),
    "hymeta_moe_gptq_gemm": (C.c_int, [C.POINTER(MoeGptqGemmArgs), _ptr]),
    "hymeta_moe_gptq_expand": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr]),
    "hymeta_silu_and_mul_int8": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i64, _ptr]),
    "hymeta_moe_combine": (C.c_int, [C.POINTER(MoeCombineArgs), _ptr]),
    "hymeta_w8a8_quantize": (C.c_int, [_ptr, _ptr, _ptr, _i32, _i32, _i64, _ptr]),
    "hymeta_w8a8_scale_out": (C.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i64, _i64, _ptr]),
    "hymeta_w8a8_gemm_split_k": (C.c_int, [_i32, _i32, _i32]),
    "hymeta_w8a8_gemm_workspace_bytes": (_sz, [_i32, _i32, _i32]),
}

STRUCTS = {
    "hymeta_gla_decode_args": GlaDecodeArgs,
    "hymeta_gla_prefill_args": GlaPrefillArgs,
    "hymeta_reshape_and_cache_args": ReshapeAndCacheArgs,
    "hymeta_paged_decode_args": PagedDecodeArgs,
    "hymeta_swa_prefill_args": SwaPrefillArgs,
    "hymeta_rope_args": RopeArgs,
    "hymeta_combine_args": CombineArgs,
    "hymeta_add_rmsnorm_args": AddRmsNormArgs,
    "hymeta_lowrank_gate_args": LowrankGateArgs,
    "hymeta_gptq_gemm_args": GptqGemmArgs,
    "hymeta_w8a8_gemm_args": W8A8GemmArgs,
    "hymeta_bf16_gemm_args": Bf16GemmArgs,
    "hymeta_gemm_partials": GemmPartials,
    "hymeta_moe_route_args": MoeRouteArgs,
    "hymeta_moe_router_args": MoeRouterArgs,
    "hymeta_moe_gemm_args": MoeGemmArgs,
    "hymeta_moe_gptq_gemm_args": MoeGptqGemmArgs,
    "hymeta_moe_combine_args": MoeCombineArgs,
}


def library_path() -> str:
    return os.environ.get("HYMETA_HIP_LIB") or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def exported_symbols():
    return sorted(_SIGNATURES)


def load() -> C.CDLL:
    """Load the library once and bind every declared symbol; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{_LIB_NAME} not found at {path}: build it with `make -C vllm-hymeta_amd` "
            "(or __graft_entry__.build()); the Hymeta ops have no non-HIP fallback")
    # torch carries its own copy of the HIP runtime (same soname).  The process must hold one
    # runtime only, and it has to be the one that owns torch's device memory and streams: loaded
    # the other way round, launches from this library fail with hipErrorNoDevice.
    import torch  # noqa: F401

    lib = C.CDLL(path)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError names the missing symbol
        fn.restype = restype
        fn.argtypes = argtypes
    version = lib.hymeta_abi_version()
    if version != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {version}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


class HymetaError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != HYMETA_OK:
        msg = load().hymeta_strerror(code).decode()
        raise HymetaError(f"{what}: {msg} (status {code})")
