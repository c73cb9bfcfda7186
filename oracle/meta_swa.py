"""Sliding-window + meta-token softmax attention: CPU restatement (test infrastructure).

Parity unpinned: the reference holds no vectors for these functions.  The
semantics follow the only executable definition in the reference, the
dense-mask ``naive_metatoken_flash_attn``
(``vllm_hymeta/attention/fa_kernel_metatoken.py:377-410``), and the two-launch
form it is meant to check: window-causal attention over the real keys
(``:235-253`` prefill, ``:157-168`` paged decode), dense attention over the
meta keys (``:174-188``, ``:261-275``), and the log-sum-exp merge
(``_update_out_and_lse``, ``:110-122``).

Conventions: token-major ``[L, H, D]``; k/v carry ``Hkv`` heads and q-head
``h`` reads kv-head ``h // (H // Hkv)``.  ``window_left`` is flash-attn's left
window: a query at absolute position ``p`` sees real keys ``j`` with
``p - window_left <= j <= p``; ``-1`` means unbounded.  The 70B backend passes
``sliding_window - 1`` (``meta_attention.py:671``); the 7B eager layer passes
``sliding_window`` itself (``hf_7B_model/window_attention.py:215``).  With
fewer queries than keys the queries are bottom-right aligned: query ``i`` sits
at position ``Lk - Lq + i`` (``fa_kernel_metatoken.py:389``).
"""

from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch


def _expand_kv(x: torch.Tensor, num_heads: int) -> torch.Tensor:
    hkv = x.shape[1]
    if hkv == num_heads:
        return x
    assert num_heads % hkv == 0
    return x.repeat_interleave(num_heads // hkv, dim=1)


def _window_mask(Lq: int, Lk: int, window_left: int) -> torch.Tensor:
    """[Lq, Lk] bool, True where the real key is visible."""
    q_id = Lk - Lq + torch.arange(Lq)
    k_id = torch.arange(Lk)
    mask = q_id[:, None] >= k_id[None, :]
    if window_left >= 0:
        mask &= (q_id[:, None] - k_id[None, :]) <= window_left
    return mask


def meta_swa_dense_ref(
    q1: torch.Tensor,
    k1: torch.Tensor,
    v1: torch.Tensor,
    k2: Optional[torch.Tensor] = None,
    v2: Optional[torch.Tensor] = None,
    q2: Optional[torch.Tensor] = None,
    window_left: int = -1,
    scale: Optional[float] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """One sequence, one softmax over {meta keys} U {visible real keys}, fp64.

    q1 ``[Lq,H,D]``; k1, v1 ``[Lk,Hkv,D]``; k2, v2 ``[M,Hkv,D]`` or None (M=0,
    the 7B layer); q2 ``[M,H,D]`` or None.  Meta queries see every meta key
    and no real key (``fa_kernel_metatoken.py:396-400``).  Returns
    ``(out, lse)`` with out ``[(M if q2)+Lq, H, D]`` -- meta-query rows FIRST,
    the vLLM-path order (``:283``) -- and lse ``[H, (M)+Lq]`` (natural log).
    """
    Lq, H, D = q1.shape
    Lk = k1.shape[0]
    if scale is None:
        scale = D ** -0.5
    M = 0 if k2 is None else k2.shape[0]
    mask = torch.ones(Lq, M + Lk, dtype=torch.bool)
    mask[:, M:] = _window_mask(Lq, Lk, window_left)
    k = _expand_kv(k1.double(), H)
    v = _expand_kv(v1.double(), H)
    if M:
        k = torch.cat([_expand_kv(k2.double(), H), k], dim=0)
        v = torch.cat([_expand_kv(v2.double(), H), v], dim=0)
    q = q1.double()
    if q2 is not None:
        assert M and q2.shape[0] == M
        q = torch.cat([q2.double(), q], dim=0)
        meta_rows = torch.zeros(M, M + Lk, dtype=torch.bool)
        meta_rows[:, :M] = True
        mask = torch.cat([meta_rows, mask], dim=0)
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    s = s.masked_fill(~mask[None], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    out = torch.einsum("hqk,khd->qhd", p, v)
    return out, lse


def lse_merge_ref(
    out: torch.Tensor, lse: torch.Tensor, block_out: torch.Tensor, block_lse: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """``_update_out_and_lse`` (``fa_kernel_metatoken.py:110-122``) in fp32.

    out/block_out ``[L,H,D]``, lse/block_lse ``[H,L]``.  Returns merged out in
    out's dtype and the merged lse (fp32).
    """
    dt = out.dtype
    o, bo = out.float(), block_out.float()
    l1, l2 = lse.float(), block_lse.float()
    new_lse = l1 - torch.nn.functional.logsigmoid(l1 - l2)
    w = torch.sigmoid(l2 - l1).transpose(0, 1)[..., None]
    o = o - w * (o - bo)
    return o.to(dt), new_lse


def _attend(q, k, v, mask, scale):
    """fp32-accumulated softmax attention returning (out in q dtype, lse fp32)."""
    H = q.shape[1]
    kk = _expand_kv(k.float(), H)
    vv = _expand_kv(v.float(), H)
    s = torch.einsum("qhd,khd->hqk", q.float(), kk) * scale
    if mask is not None:
        s = s.masked_fill(~mask[None], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    out = torch.einsum("hqk,khd->qhd", p, vv).to(q.dtype)
    return out, lse


def meta_swa_two_pass_ref(
    q1, k1, v1, k2, v2, q2=None, window_left: int = -1, scale: Optional[float] = None
) -> torch.Tensor:
    """The reference's computational structure, roundings included.

    Pass 1: window-causal attention over the real keys; pass 2: dense attention
    of all queries (real then meta) over the meta keys; both passes round their
    output to the activation dtype as flash-attn does; merge in fp32 and round
    again (``fa_kernel_metatoken.py:235-285``).  Returns ``[(M)+Lq, H, D]`` with
    meta rows first.
    """
    Lq, H, D = q1.shape
    Lk = k1.shape[0]
    if scale is None:
        scale = D ** -0.5
    out1, lse1 = _attend(q1, k1, v1, _window_mask(Lq, Lk, window_left), scale)
    q = q1 if q2 is None else torch.cat([q1, q2], dim=0)
    out2, lse2 = _attend(q, k2, v2, None, scale)
    out, _ = lse_merge_ref(out1, lse1, out2[:Lq], lse2[:, :Lq])
    if q2 is not None:
        out = torch.cat([out2[Lq:], out], dim=0)
    return out


def meta_swa_varlen_ref(
    q1: torch.Tensor,
    k1: torch.Tensor,
    v1: torch.Tensor,
    cu_seqlens_q: Sequence[int],
    cu_seqlens_k: Sequence[int],
    k2: Optional[torch.Tensor] = None,
    v2: Optional[torch.Tensor] = None,
    q2: Optional[torch.Tensor] = None,
    window_left: int = -1,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """Packed prefill batch (``metatoken_flash_attn_varlen_func``,
    ``fa_kernel_metatoken.py:329-374``).  Returns fp64 ``[(M)+T, H, D]``; the
    meta-query rows, computed once, come first (``:283``)."""
    T, H, D = q1.shape
    outs = []
    if q2 is not None:
        M = q2.shape[0]
        s = torch.einsum("qhd,khd->hqk", q2.double(), _expand_kv(k2.double(), H))
        s = s * (scale if scale is not None else D ** -0.5)
        p = torch.softmax(s, dim=-1)
        outs.append(torch.einsum("hqk,khd->qhd", p, _expand_kv(v2.double(), H)))
    for i in range(len(cu_seqlens_q) - 1):
        qs, qe = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        ks, ke = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
        if qe <= qs:
            continue
        o, _ = meta_swa_dense_ref(q1[qs:qe], k1[ks:ke], v1[ks:ke], k2, v2, None,
                                  window_left, scale)
        outs.append(o)
    if not outs:
        return torch.zeros(0, H, D, dtype=torch.float64)
    return torch.cat(outs, dim=0)


def _gather(cache: torch.Tensor, block_table: torch.Tensor, seq_len: int) -> torch.Tensor:
    """cache ``[nblk, bs, Hkv, D]`` -> ``[seq_len, Hkv, D]`` through one block table row."""
    bs = cache.shape[1]
    pos = torch.arange(seq_len)
    blk = block_table[pos // bs].long()
    return cache[blk, pos % bs]


def meta_swa_paged_decode_ref(
    q: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    cache_seqlens: Sequence[int],
    block_tables: torch.Tensor,
    k2: Optional[torch.Tensor] = None,
    v2: Optional[torch.Tensor] = None,
    window_left: int = -1,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """Single-query paged decode (``metatoken_flash_attn_with_kvcache``,
    ``fa_kernel_metatoken.py:287-326``; dispatch ``meta_attention.py:954-981``).

    q ``[B,H,D]``; caches ``[nblk, bs, Hkv, D]`` (``meta_attention.py:81-92``);
    ``cache_seqlens[b]`` counts the keys of sequence b INCLUDING the token
    being decoded (its K/V was written before the call, ``:790-799``);
    ``block_tables [B, max_blocks]``.  Returns fp64 ``[B,H,D]``; a sequence of
    length 0 with no meta keys yields zeros.
    """
    B, H, D = q.shape
    out = torch.zeros(B, H, D, dtype=torch.float64)
    for b in range(B):
        L = int(cache_seqlens[b])
        if L == 0 and k2 is None:
            continue
        kk = _gather(key_cache, block_tables[b], L)
        vv = _gather(value_cache, block_tables[b], L)
        if L == 0:
            s = torch.einsum("qhd,khd->hqk", q[b:b + 1].double(), _expand_kv(k2.double(), H))
            s = s * (scale if scale is not None else D ** -0.5)
            p = torch.softmax(s, dim=-1)
            out[b] = torch.einsum("hqk,khd->qhd", p, _expand_kv(v2.double(), H))[0]
            continue
        o, _ = meta_swa_dense_ref(q[b:b + 1], kk, vv, k2, v2, None, window_left, scale)
        out[b] = o[0]
    return out


def meta_swa_paged_prefill_ref(
    q: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    cu_seqlens_q: Sequence[int],
    seq_lens: Sequence[int],
    block_tables: torch.Tensor,
    k2: Optional[torch.Tensor] = None,
    v2: Optional[torch.Tensor] = None,
    window_left: int = -1,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """Chunked / prefix prefill through the block table.

    The new chunk's K/V are already in the cache (``meta_attention.py:790-799``);
    sequence i has ``seq_lens[i]`` keys (context + chunk) and its
    ``cu_seqlens_q`` slice of queries is bottom-right aligned against them --
    the behaviour the block-table branch at ``meta_attention.py:885-919`` is
    evidently meant to have (as written it hands the un-paged chunk K/V to a
    paged call).  Returns fp64 ``[T,H,D]``.
    """
    T, H, D = q.shape
    out = torch.zeros(T, H, D, dtype=torch.float64)
    for i in range(len(cu_seqlens_q) - 1):
        qs, qe = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        L = int(seq_lens[i])
        if qe <= qs:
            continue
        kk = _gather(key_cache, block_tables[i], L)
        vv = _gather(value_cache, block_tables[i], L)
        o, _ = meta_swa_dense_ref(q[qs:qe], kk, vv, k2, v2, None, window_left, scale)
        out[qs:qe] = o
    return out
