"""Meta-token attention entry points, names and argument meaning as in the reference
(``vllm_hymeta/attention/fa_kernel_metatoken.py:287-374``), backed by the fused HIP kernels.

The reference runs two flash-attn launches (window-causal over the real keys, dense over the
meta keys) and merges them by log-sum-exp; here the meta keys are one more always-visible tile
of a single online softmax, so there is one launch and no intermediate bf16 rounding.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import hip_ops


def _check_common(causal: bool, window_size: Tuple[int, int], dropout_p: float, alibi_slopes,
                  softcap: float, return_attn_probs: bool, attn_mask) -> None:
    # the reference asserts these two (fa_kernel_metatoken.py:151, 228)
    assert causal and window_size[1] in (0, -1), "meta-token attention is causal with no right window"
    if dropout_p:
        raise NotImplementedError("dropout is not supported (inference path)")
    if alibi_slopes is not None:
        raise NotImplementedError("alibi_slopes is not supported by the HIP meta-token kernels")
    if softcap:
        raise NotImplementedError("logits soft cap is not supported by the HIP meta-token kernels")
    if return_attn_probs or attn_mask is not None:
        raise NotImplementedError("return_attn_probs / attn_mask are not supported")


def metatoken_flash_attn_varlen_func(
    q1: torch.Tensor,
    q2: Optional[torch.Tensor],
    k1: torch.Tensor,
    k2: Optional[torch.Tensor],
    v1: torch.Tensor,
    v2: Optional[torch.Tensor],
    cu_seqlens_q: Optional[torch.Tensor] = None,
    cu_seqlens_k: Optional[torch.Tensor] = None,
    max_seqlen_q: Optional[int] = None,
    max_seqlen_k: Optional[int] = None,
    num_meta_tokens: int = 128,
    dropout_p: float = 0.0,
    softmax_scale: Optional[float] = None,
    causal: bool = False,
    window_size: Tuple[int, int] = (-1, -1),
    alibi_slopes=None,
    block_tables: Optional[torch.Tensor] = None,
    softcap: float = 0.0,
    deterministic: bool = False,
    return_attn_probs: bool = False,
    attn_mask=None,
) -> torch.Tensor:
    """Packed prefill attention.  q1 ``[T,H,D]``; k1/v1 ``[Tk,Hkv,D]`` (or, with ``block_tables``,
    the paged caches ``[nblk,bs,Hkv,D]``); q2 ``[M,H,D]`` or None; k2/v2 ``[M,Hkv,D]`` or None.
    Returns ``[1, (M if q2) + T, H, D]`` with the meta-query rows first
    (reference: ``fa_kernel_metatoken.py:283``)."""
    _check_common(causal, window_size, dropout_p, alibi_slopes, softcap, return_attn_probs, attn_mask)
    if k2 is not None:
        assert num_meta_tokens == k2.shape[0]
    if q2 is not None:
        assert block_tables is None and q1.shape[0] == k1.shape[0], \
            "meta_tokens' query doesn't support decoding"
        assert num_meta_tokens == q2.shape[0]
    if max_seqlen_q is None:
        max_seqlen_q = q1.shape[0]
    if block_tables is not None:
        seq_lens_k = (cu_seqlens_k[1:] - cu_seqlens_k[:-1]).to(torch.int32).contiguous()
        out = hip_ops.meta_swa_prefill(
            q1, cu_seqlens_q, max_seqlen_q, key_cache=k1, value_cache=v1, block_tables=block_tables,
            seq_lens_k=seq_lens_k, meta_query=q2, meta_key=k2, meta_value=v2,
            window_left=window_size[0], scale=softmax_scale, max_seqlen_k=max_seqlen_k or 0)
    else:
        out = hip_ops.meta_swa_prefill(
            q1, cu_seqlens_q, max_seqlen_q, key=k1, value=v1, cu_seqlens_k=cu_seqlens_k,
            meta_query=q2, meta_key=k2, meta_value=v2, window_left=window_size[0], scale=softmax_scale,
            max_seqlen_k=max_seqlen_k or 0)
    return out.unsqueeze(0)


def metatoken_flash_attn_with_kvcache(
    q1: torch.Tensor,
    q2: Optional[torch.Tensor],
    k2: Optional[torch.Tensor],
    v2: Optional[torch.Tensor],
    key_cache: Optional[torch.Tensor] = None,
    value_cache: Optional[torch.Tensor] = None,
    cache_seqlens: Optional[torch.Tensor] = None,
    num_meta_tokens: int = 128,
    dropout_p: float = 0.0,
    softmax_scale: Optional[float] = None,
    causal: bool = False,
    window_size: Tuple[int, int] = (-1, -1),
    alibi_slopes=None,
    block_tables: Optional[torch.Tensor] = None,
    softcap: float = 0.0,
    deterministic: bool = False,
    return_attn_probs: bool = False,
    attn_mask=None,
) -> torch.Tensor:
    """Single-token paged decode.  q1 ``[B,H,D]``; caches ``[nblk,bs,Hkv,D]``; ``cache_seqlens
    [B]`` int32 counts each sequence's keys including the token being decoded; ``block_tables
    [B,max_blocks]`` int32.  Returns ``[B, 1, H, D]``."""
    _check_common(causal, window_size, dropout_p, alibi_slopes, softcap, return_attn_probs, attn_mask)
    if q2 is not None:
        raise AssertionError("meta_tokens' query doesn't support decoding")
    if k2 is not None:
        assert num_meta_tokens == k2.shape[0]
    out = hip_ops.meta_swa_paged_decode(
        q1, key_cache, value_cache, cache_seqlens, block_tables, meta_key=k2, meta_value=v2,
        window_left=window_size[0], scale=softmax_scale)
    return out.unsqueeze(1)
