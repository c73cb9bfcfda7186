"""Gated-linear-attention (GLA) recurrence: CPU restatement (test infrastructure).

The single-token step (``gla_decode_step_ref``) is pinned to outputs of the reference's own
Triton kernel run under the CPU interpreter (``tests/golden/ref_gla_decode.npz``, script
``tests/golden/make_ref_fixtures.py``).  The prefill functions are parity unpinned: their kernel
is third party and the reference holds no vectors for it.  The math is restated from

* the single-token Triton step, ``vllm_hymeta/models/my_fused_recurrent.py:58-80``
  (7B variant with explicit casts: ``vllm_hymeta/model_for_7B/my_fused_recurrent.py:60-80``);
* the per-sequence chunked prefill driver,
  ``vllm_hymeta/models/modeling_hymeta.py:470-529`` (70B) and
  ``vllm_hymeta/model_for_7B/gla_attention.py:179-221`` (7B), whose third-party
  kernel ``fla.ops.gla.fused_chunk_gla`` (flash-linear-attention 0.1, not
  vendored) computes the recurrence ``S_t = diag(exp g_t) S_{t-1} + k_t^T v_t``,
  ``o_t = (scale * q_t) S_t`` and returns the final state.

Layouts are token-major, as the vLLM path holds them: q ``[T, H, D]``;
k, v, g ``[T, Hkv, D]`` with ``H % Hkv == 0`` (q-head ``h`` uses kv-head
``h // (H // Hkv)``, the effect of ``repeat_kv`` at
``model_for_7B/gla_attention.py:265-267``); state ``[slots, H, Dk, Dv]``.
"""

from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch


def _expand_kv(x: torch.Tensor, num_heads: int) -> torch.Tensor:
    """[T, Hkv, D] -> [T, H, D] by repeating each kv head for its group."""
    hkv = x.shape[1]
    if hkv == num_heads:
        return x
    assert num_heads % hkv == 0
    return x.repeat_interleave(num_heads // hkv, dim=1)


def gla_recurrent_ref(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    g: torch.Tensor,
    initial_state: Optional[torch.Tensor] = None,
    scale: Optional[float] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact (fp64) sequential GLA over one sequence.

    q ``[T,H,D]``, k/v/g ``[T,Hkv,D]``, initial_state ``[H,D,D]`` or None (zeros).
    Returns ``(o [T,H,D] fp64, final_state [H,D,D] fp64)``.  The output of step
    t uses the state *after* the update with token t
    (``my_fused_recurrent.py:72-76``).
    """
    T, H, D = q.shape
    if scale is None:
        scale = D ** -0.5  # models/my_fused_recurrent_gla.py:20-21
    q64 = q.double() * scale
    k64 = _expand_kv(k.double(), H)
    v64 = _expand_kv(v.double(), H)
    g64 = _expand_kv(g.double(), H)
    S = torch.zeros(H, D, D, dtype=torch.float64) if initial_state is None \
        else initial_state.double().clone()
    o = torch.empty(T, H, D, dtype=torch.float64)
    decay = torch.exp(g64)
    for t in range(T):
        S = decay[t][:, :, None] * S + k64[t][:, :, None] * v64[t][:, None, :]
        o[t] = torch.einsum("hk,hkv->hv", q64[t], S)
    return o, S


def gla_decode_step_ref(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    g: torch.Tensor,
    state: torch.Tensor,
    slot_idx: torch.Tensor,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """One decode token per sequence, state updated IN PLACE at ``state[slot_idx[b]]``.

    Follows the arithmetic types of ``my_fused_recurrent_fwd_kernel``
    (``my_fused_recurrent.py:58-80``) operation by operation; pinned bit for bit (fp16 and fp32
    activations, both variants of the kernel) by ``tests/golden/ref_gla_decode.npz``, which holds
    outputs of that kernel run under Triton's CPU interpreter:

    * ``q * scale`` (``:59``) is a tensor-times-scalar product: Triton casts the fp32 scalar to the
      tensor's dtype and multiplies in that dtype, so the product AND the scale itself are rounded to
      the activation dtype (bf16: ``128 ** -0.5`` becomes 0.08837890625) before ``.to(float32)`` at
      ``:75``;
    * the outer product ``k v^T`` is formed in the activation dtype (``:65``) and added in fp32
      (``:72``);
    * the decay ``exp(g)`` (g widened to fp32; the 7B variant hands g over in the activation dtype,
      ``model_for_7B/my_fused_recurrent.py:63``) and the state update are fp32 (``:69-72``);
    * the output contracts the rounded ``q * scale`` with the *updated, unrounded fp32* state
      (``:75-76``), summed here in key-channel order, the order of the interpreter's ``tl.sum``
      (on a GPU the order is the compiler's tree);
    * state and output are rounded to their storage dtypes on store (``:79-80``).

    ``exp`` is numpy's float32 routine, the one the interpreter run used (a GPU's differs from it in
    the last bit now and then, like any two ``expf``).  ``slot_idx[b] == -1`` skips the sequence:
    state untouched, output row left as it was (here: zeros) (``:33-35``).

    q ``[B,H,D]``; k, v, g ``[B,Hkv,D]``; state ``[slots,H,D,D]`` (any float
    dtype; bf16 in the reference, ``modeling_hymeta.py:1166``).  Returns
    ``o [B,H,D]`` in q's dtype.
    """
    B, H, D = q.shape
    if scale is None:
        scale = D ** -0.5
    act = q.dtype
    kx = _expand_kv(k, H).to(act)
    vx = _expand_kv(v, H).to(act)
    gx = _expand_kv(g, H).float()
    out = torch.zeros(B, H, D, dtype=act)
    scale_act = torch.tensor(scale, dtype=torch.float32).to(act)
    for b in range(B):
        slot = int(slot_idx[b])
        if slot == -1:
            continue
        kv_outer = (kx[b][:, :, None] * vx[b][:, None, :]).float()
        ratio = torch.from_numpy(np.exp(gx[b].numpy()))[:, :, None]
        s_new = kv_outer + ratio * state[slot].float()
        qs = (q[b] * scale_act).float()
        prod = qs[:, :, None] * s_new
        o = torch.zeros(H, D, dtype=torch.float32)
        for d in range(D):
            o = o + prod[:, d, :]
        state[slot] = s_new.to(state.dtype)
        out[b] = o.to(act)
    return out


def gla_prefill_varlen_ref(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    g: torch.Tensor,
    state: torch.Tensor,
    slot_idx: torch.Tensor,
    cu_seqlens: Sequence[int],
    context_lens: Optional[Sequence[int]] = None,
    meta_state: Optional[torch.Tensor] = None,
    scale: Optional[float] = None,
) -> torch.Tensor:
    """All prefill sequences of one step; state written back IN PLACE.

    Per sequence i (tokens ``cu_seqlens[i]:cu_seqlens[i+1]``) the initial state
    is

    * ``state[slot]`` when the sequence resumes a chunked prefill
      (``modeling_hymeta.py:486-490``; 7B: ``gla_attention.py:196``);
    * for a first chunk (``context_lens[i] == 0``): ``meta_state`` when given
      (70B, ``modeling_hymeta.py:501-504``), else zeros -- the effect of the
      model zeroing the slot before the layers run
      (``modeling_hymeta.py:1202-1227``, ``model_for_7B/modeling_gla_swa.py:391-414``).

    With ``context_lens=None`` every sequence starts from ``state[slot]``
    (the kernel-level contract of the 7B call site).  The final state is
    rounded to the state dtype on write-back (``modeling_hymeta.py:528``).
    ``slot == -1`` sequences are skipped (output zeros).
    Returns ``o [T,H,D]`` in q's dtype.
    """
    T, H, D = q.shape
    out = torch.zeros(T, H, D, dtype=q.dtype)
    nseq = len(cu_seqlens) - 1
    for i in range(nseq):
        s, e = int(cu_seqlens[i]), int(cu_seqlens[i + 1])
        slot = int(slot_idx[i])
        if slot == -1 or e <= s:
            continue
        first = context_lens is not None and int(context_lens[i]) == 0
        if first:
            init = None if meta_state is None else meta_state.double()
        else:
            init = state[slot].double()
        o, S = gla_recurrent_ref(q[s:e], k[s:e], v[s:e], g[s:e], init, scale)
        out[s:e] = o.to(q.dtype)
        state[slot] = S.to(state.dtype)
    return out
