"""Constant-size per-sequence state slots: which slot of the recurrent-state tensors a
(request, sequence) pair owns.

Host logic of the reference's cache managers (``vllm_hymeta/models/hymeta_cache.py:24-62``,
``vllm_hymeta/model_for_7B/gla_cache.py:9-33``), which inherit it from vLLM 0.10.0's
``ConstantSizeCache`` (third-party, not vendored; behaviour restated here): finished requests
release their slots first; a new request takes a free slot; a request that forks a new sequence
(parallel sampling) gets a fresh slot initialised as a copy of a sibling's state; a request that is
both scheduled and finished in the same step maps to ``PAD_SLOT_ID``.  The result lists one slot
per scheduled sequence in scheduler order, which is the batch order (prefills, then decodes) the
kernels index by.
"""

from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch

from ._compat import PAD_SLOT_ID


class SlotAllocator:
    def __init__(self, num_slots: int, copy_slot: Optional[Callable[[int, int], None]] = None):
        self.num_slots = int(num_slots)
        self._free: List[int] = list(range(self.num_slots))
        self._owned: Dict[str, Dict[int, int]] = {}
        self._copy_slot = copy_slot

    # ------------------------------------------------------------------ queries
    @property
    def num_free(self) -> int:
        return len(self._free)

    def slot_of(self, request_id: str, seq_id: int) -> Optional[int]:
        return self._owned.get(request_id, {}).get(seq_id)

    # ------------------------------------------------------------------ updates
    def release(self, finished_request_ids: Iterable[str]) -> None:
        for rid in finished_request_ids:
            slots = self._owned.pop(rid, None)
            if slots:
                self._free.extend(slots.values())

    def _assign(self, rid: str, seq_id: int, finished: Iterable[str]) -> int:
        if rid in finished:
            return PAD_SLOT_ID
        owned = self._owned.get(rid)
        if owned is None:
            if not self._free:
                raise RuntimeError("no free recurrent-state slot: more live sequences than max_num_seqs")
            slot = self._free.pop()
            self._owned[rid] = {seq_id: slot}
            return slot
        if seq_id in owned:
            return owned[seq_id]
        if not self._free:
            raise RuntimeError("no free recurrent-state slot: more live sequences than max_num_seqs")
        src = next(iter(owned.values()))
        slot = self._free.pop()
        if self._copy_slot is not None:
            self._copy_slot(src, slot)
        owned[seq_id] = slot
        return slot

    def assign(self, request_ids_to_seq_ids: Dict[str, List[int]],
               finished_request_ids: Iterable[str]) -> List[int]:
        """Release the finished requests, then return one slot per scheduled sequence."""
        finished = list(finished_request_ids)
        self.release(finished)
        return [self._assign(rid, sid, finished)
                for rid, sids in request_ids_to_seq_ids.items() for sid in sids]


class ConstantSizeStateCache:
    """State tensors shaped ``[layers, slots, ...]`` plus the slot bookkeeping.  Sub-classes give
    ``cache`` (a tuple of tensors).  Mirrors the call surface the model classes use:
    ``current_run_tensors``, ``copy_inputs_before_cuda_graphs``,
    ``get_seqlen_agnostic_capture_inputs``."""

    def __init__(self, max_batch_size: int, device="cuda"):
        self.device = torch.device(device)
        self.slots = SlotAllocator(max_batch_size, self._copy_cache)

    @property
    def cache(self) -> Tuple[torch.Tensor, ...]:
        raise NotImplementedError

    def _slot_tensors(self) -> Tuple[torch.Tensor, ...]:
        """The tensors whose dim 1 is the slot axis."""
        return self.cache

    def _copy_cache(self, from_index: int, to_index: int) -> None:
        for t in self._slot_tensors():
            t[:, to_index].copy_(t[:, from_index], non_blocking=True)

    def _indices(self, kwargs) -> List[int]:
        return self.slots.assign(kwargs["request_ids_to_seq_ids"], kwargs["finished_requests_ids"])

    def current_run_tensors(self, **kwargs):
        if "seqlen_agnostic_capture_inputs" in kwargs:     # graph-capture run: caller's buffers
            cache_tensors, state_indices_tensor = kwargs["seqlen_agnostic_capture_inputs"]
        else:
            cache_tensors = self.cache
            state_indices_tensor = torch.as_tensor(self._indices(kwargs), dtype=torch.int32,
                                                   device=self.device)
        return cache_tensors, state_indices_tensor

    def copy_inputs_before_cuda_graphs(self, input_buffers, **kwargs) -> None:
        _, index_buffer = input_buffers["seqlen_agnostic_capture_inputs"]
        idx = self._indices(kwargs)
        idx.extend([PAD_SLOT_ID] * (index_buffer.shape[0] - len(idx)))   # kernels skip slot -1
        index_buffer.copy_(torch.as_tensor(idx, dtype=torch.int32, device=index_buffer.device))

    def get_seqlen_agnostic_capture_inputs(self, batch_size: int):
        return self.cache, torch.full((batch_size,), PAD_SLOT_ID, dtype=torch.int32, device=self.device)
