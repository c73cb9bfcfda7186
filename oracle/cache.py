"""The eager models' per-layer cache: CPU restatement (test infrastructure, never on the product path).

Pinned: ``tests/golden/ref_cache.npz`` holds outputs of the reference's own ``Cache.update``
(``hf_7B_model/cache.py:40-112``, ``hf_model/cache.py:40-121``) on seeded inputs; ``tests/test_ref_fixtures.py``
checks this restatement against them bit for bit.

The eager models keep, per softmax layer, the last ``window_size`` roped keys / values in position order
(a window that rolls left when full) and, per GLA layer, the latest recurrent state; the 70B cache also keeps
the meta tokens' K/V.  The serving path keeps the same information in a paged KV cache addressed through a
block table and a slot-indexed state tensor; ``window_positions_ref`` states which positions the rolling window
holds so that tests can tie the two together."""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch


class RollingCacheRef:
    def __init__(self):
        self.states: List[Dict[str, object]] = []
        self.seen_tokens = 0

    def update(self, layer_idx: int = 0, offset: int = 1, attn_state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               recurrent_state: Optional[torch.Tensor] = None, meta_state=None, window_size: Optional[int] = None):
        """``attn_state`` = (key, value) ``[b, h, n, d]`` of the ``n`` new tokens."""
        if layer_idx == 0:
            self.seen_tokens += offset                                  # :70-71
        if len(self.states) <= layer_idx:                               # first call of the layer: :79-89
            if attn_state is not None and window_size is not None and attn_state[0].shape[-2] > window_size:
                attn_state = tuple(t[..., -window_size:, :].contiguous() for t in attn_state)
            self.states.append(dict(recurrent_state=recurrent_state, attn_state=attn_state, meta_state=meta_state))
            return self.states[layer_idx]
        st = self.states[layer_idx]
        if recurrent_state is not None:
            st["recurrent_state"] = recurrent_state                     # replaced, :92-93
        if attn_state is not None:
            n = attn_state[0].shape[-2]
            k_old, v_old = st["attn_state"]
            if window_size is not None and k_old.shape[-2] == window_size:
                # full window: shift left by n, new tokens at the end (:96-104)
                k_new, v_new = k_old.roll(-n, -2), v_old.roll(-n, -2)
                k_new[..., -n:, :] = attn_state[0]
                v_new[..., -n:, :] = attn_state[1]
            else:
                # not full yet: append, WITHOUT cutting to the window (:105-107) -- the window can overshoot
                k_new, v_new = torch.cat([k_old, attn_state[0]], -2), torch.cat([v_old, attn_state[1]], -2)
            st["attn_state"] = (k_new, v_new)
        if meta_state is not None:
            st["meta_state"] = meta_state
        return st


def window_positions_ref(chunks: List[int], window_size: int) -> List[List[int]]:
    """Token positions the rolling window holds after each update, for a sequence fed in ``chunks``."""
    held: List[int] = []
    out, t = [], 0
    for i, n in enumerate(chunks):
        new = list(range(t, t + n))
        t += n
        if i == 0:
            held = new[-window_size:] if n > window_size else new
        elif len(held) == window_size:
            held = held[n:] + new            # n <= window_size (the reference's slice assignment needs it)
        else:
            held = held + new
        out.append(list(held))
    return out
