"""Paged-KV indexing: CPU restatement (test infrastructure).  Integer work, bit-exact.

Parity unpinned by vectors (the reference ships none); the arithmetic is
restated from the call sites in ``vllm_hymeta/attention/meta_attention.py``:
cache shape ``:81-92``, the scatter ``reshape_and_cache_flash`` ``:790-799``,
slot-mapping construction ``:488-494`` and block-table trimming ``:474-485``.
``reshape_and_cache_flash`` and ``compute_slot_mapping`` themselves live in
vLLM 0.10.0 (``requirements.txt:8``; not vendored); their published behaviour
is: ``slot = block_table[pos // bs] * bs + pos % bs``, a negative slot
(``PAD_SLOT_ID = -1``) writes nothing.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

PAD_SLOT_ID = -1


def kv_cache_shape_ref(num_blocks: int, block_size: int, num_kv_heads: int,
                       head_size: int) -> Tuple[int, ...]:
    """``MetaAttentionBackend.get_kv_cache_shape`` (``meta_attention.py:81-92``)."""
    if block_size % 16 != 0 or block_size < 16:
        raise ValueError(
            f"Invalid block size {block_size}. Must be a multiple of 16 and at least 16.")
    return (2, num_blocks, block_size, num_kv_heads, head_size)


def reshape_and_cache_ref(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
) -> None:
    """Scatter token rows into the paged cache IN PLACE (``meta_attention.py:790-799``).

    key/value ``[T, Hkv, D]``; caches ``[nblk, bs, Hkv, D]``; ``slot_mapping
    [T]`` int64.  Token t goes to block ``slot // bs``, offset ``slot % bs``;
    negative slots are skipped.  Only the first ``len(slot_mapping)`` tokens are
    written.
    """
    bs = key_cache.shape[1]
    for t in range(slot_mapping.shape[0]):
        slot = int(slot_mapping[t])
        if slot < 0:
            continue
        key_cache[slot // bs, slot % bs] = key[t]
        value_cache[slot // bs, slot % bs] = value[t]


def slot_mapping_ref(
    block_table: Optional[Sequence[int]],
    seq_len: int,
    context_len: int,
    block_size: int,
    is_prompt: bool,
    query_len: int,
    sliding_window: Optional[int] = None,
) -> List[int]:
    """Slots for the new tokens ``[context_len, seq_len)`` of one sequence.

    Restates ``compute_slot_mapping_start_idx`` + ``compute_slot_mapping`` as
    called at ``meta_attention.py:488-494``: with a sliding window a prompt's
    tokens that fall out of the window before decoding starts are mapped to
    ``PAD_SLOT_ID``; a profile run (no block table) maps everything to PAD.
    """
    if block_table is None or len(block_table) == 0:
        return [PAD_SLOT_ID] * seq_len
    start_idx = 0
    if is_prompt and sliding_window is not None:
        start_idx = max(0, query_len - sliding_window)
    out = [PAD_SLOT_ID] * max(0, start_idx - context_len)
    for pos in range(max(start_idx, context_len), seq_len):
        out.append(int(block_table[pos // block_size]) * block_size + pos % block_size)
    return out


def gather_paged_ref(cache: torch.Tensor, block_table: Sequence[int], seq_len: int) -> torch.Tensor:
    """``[nblk, bs, Hkv, D]`` -> ``[seq_len, Hkv, D]`` through one block-table row."""
    bs = cache.shape[1]
    pos = torch.arange(seq_len)
    blk = torch.as_tensor(block_table, dtype=torch.long)[pos // bs]
    return cache[blk, pos % bs]
