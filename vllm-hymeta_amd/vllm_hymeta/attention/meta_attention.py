"""vLLM (V0) attention backend for the Hymeta 70B model: paged KV write, varlen prefill and
paged decode of sliding-window + meta-token attention, all on the HIP kernels.

Surface and semantics follow the reference backend
(``vllm_hymeta/attention/meta_attention.py``): ``MetaAttentionBackend`` (``:53-126``),
``MetaAttentionMetadata`` (``:128-409``), ``MetaAttentionMetadataBuilder`` (``:411-608``),
``MetaAttentionImpl`` (``:611-982``).  With vLLM installed the classes derive from vLLM's
abstract bases; without it (unit tests, kernel harness) from same-named stand-ins.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass
from itertools import accumulate
from typing import Any, Dict, List, Optional, Tuple, Type

import torch

from .. import hip_ops
from .._compat import (PAD_SLOT_ID, AttentionBackend, AttentionImpl, AttentionMetadata,
                       AttentionMetadataBuilder, AttentionType)
from .fa_kernel_metatoken import metatoken_flash_attn_varlen_func, metatoken_flash_attn_with_kvcache

logger = logging.getLogger(__name__)


# --------------------------------------------------------------------------- host helpers
def slot_mapping_for_sequence(out: List[int], block_table: Optional[List[int]], seq_len: int,
                              context_len: int, start_idx: int, block_size: int) -> None:
    """Append the cache slots of tokens ``[context_len, seq_len)`` of one sequence.

    ``slot = block_table[pos // block_size] * block_size + pos % block_size``; positions below
    ``start_idx`` (prompt tokens that a sliding window drops before decoding starts) and profile
    runs (no block table) map to ``PAD_SLOT_ID``.  (vLLM ``compute_slot_mapping`` as called at
    reference ``meta_attention.py:488-494``.)"""
    if not block_table:
        out.extend([PAD_SLOT_ID] * seq_len)
        return
    out.extend([PAD_SLOT_ID] * max(0, start_idx - context_len))
    for pos in range(max(start_idx, context_len), seq_len):
        out.append(block_table[pos // block_size] * block_size + pos % block_size)


def slot_mapping_start_idx(is_prompt: bool, query_len: int, sliding_window: Optional[int]) -> int:
    if is_prompt and sliding_window is not None:
        return max(0, query_len - sliding_window)
    return 0


def _h2d(data: List[int], dtype: torch.dtype, device, pin: bool) -> torch.Tensor:
    t = torch.tensor(data, dtype=dtype, pin_memory=pin and torch.cuda.is_available(), device="cpu")
    return t.to(device=device, non_blocking=True)


def _pad_block_tables(tables: List[List[int]], device) -> torch.Tensor:
    width = max((len(t) for t in tables), default=0)
    padded = torch.zeros(len(tables), width, dtype=torch.int32)
    for i, t in enumerate(tables):
        if t:
            padded[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
    return padded.to(device=device, non_blocking=True)


# --------------------------------------------------------------------------- backend
class MetaAttentionBackend(AttentionBackend):
    accept_output_buffer: bool = True

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        # the HIP kernels are built for the one head size the Hymeta models use
        return [128]

    @staticmethod
    def get_name() -> str:
        return "META_FLASH_ATTN"

    @staticmethod
    def get_impl_cls() -> Type["MetaAttentionImpl"]:
        return MetaAttentionImpl

    @staticmethod
    def get_metadata_cls() -> Type["MetaAttentionMetadata"]:
        return MetaAttentionMetadata

    @staticmethod
    def get_builder_cls() -> Type["MetaAttentionMetadataBuilder"]:
        return MetaAttentionMetadataBuilder

    @staticmethod
    def get_builder_cls_v1():
        """The V1 engine's builder (the reference serves through V0 only; see ``meta_attention_v1``)."""
        from .meta_attention_v1 import MetaAttentionMetadataBuilderV1
        return MetaAttentionMetadataBuilderV1

    @staticmethod
    def get_state_cls():
        from vllm.attention.backends.utils import CommonAttentionState
        return CommonAttentionState

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        if block_size % 16 != 0 or block_size < 16:
            raise ValueError(
                f"Invalid block size {block_size}. Must be a multiple of 16 and at least 16.")
        return (2, num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        """Copy whole blocks between two caches (host<->device swap on preemption).
        ``src_to_dst [n, 2]`` int64 pairs of block numbers."""
        pairs = src_to_dst.tolist()
        for which in (0, 1):
            for s, d in pairs:
                dst_kv_cache[which, d].copy_(src_kv_cache[which, s], non_blocking=True)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        """Copy blocks inside every layer's cache (beam search forks)."""
        if src_to_dists.numel() == 0:
            return
        src = src_to_dists[:, 0].long()
        dst = src_to_dists[:, 1].long()
        for kv in kv_caches:
            kv[:, dst] = kv[:, src]


# --------------------------------------------------------------------------- metadata
@dataclass
class MetaAttentionMetadata(AttentionMetadata):
    """Per-step attention metadata; field names as vLLM's FlashAttention metadata so the model
    runner and the model code (``query_start_loc``, ``context_lens_tensor`` ...) find them."""
    seq_lens: Optional[List[int]]
    seq_lens_tensor: Optional[torch.Tensor]
    max_prefill_seq_len: int
    max_decode_seq_len: int
    context_lens_tensor: Optional[torch.Tensor]
    block_tables: Optional[torch.Tensor]
    use_cuda_graph: bool
    max_query_len: Optional[int] = None
    max_decode_query_len: Optional[int] = None
    query_start_loc: Optional[torch.Tensor] = None
    seq_start_loc: Optional[torch.Tensor] = None
    _cached_prefill_metadata: Optional["MetaAttentionMetadata"] = None
    _cached_decode_metadata: Optional["MetaAttentionMetadata"] = None
    encoder_seq_lens: Optional[List[int]] = None
    encoder_seq_lens_tensor: Optional[torch.Tensor] = None
    encoder_seq_start_loc: Optional[torch.Tensor] = None
    max_encoder_seq_len: Optional[int] = None
    num_encoder_tokens: Optional[int] = None
    cross_slot_mapping: Optional[torch.Tensor] = None
    cross_block_tables: Optional[torch.Tensor] = None

    @property
    def is_all_encoder_attn_metadata_set(self) -> bool:
        return (self.encoder_seq_lens is not None and self.encoder_seq_lens_tensor is not None
                and self.max_encoder_seq_len is not None)

    @property
    def is_all_cross_attn_metadata_set(self) -> bool:
        return (self.is_all_encoder_attn_metadata_set and self.cross_slot_mapping is not None
                and self.cross_block_tables is not None)

    @staticmethod
    def _cut(x, sl):
        return None if x is None else x[sl]

    @property
    def prefill_metadata(self) -> Optional["MetaAttentionMetadata"]:
        """View on the first ``num_prefills`` sequences / ``num_prefill_tokens`` tokens."""
        if self.num_prefills == 0:
            return None
        if self._cached_prefill_metadata is None:
            n, nt = self.num_prefills, self.num_prefill_tokens
            self._cached_prefill_metadata = MetaAttentionMetadata(
                num_prefills=n, num_prefill_tokens=nt, num_decode_tokens=0,
                slot_mapping=self._cut(self.slot_mapping, slice(0, nt)),
                multi_modal_placeholder_index_maps=self.multi_modal_placeholder_index_maps,
                enable_kv_scales_calculation=self.enable_kv_scales_calculation,
                seq_lens=self._cut(self.seq_lens, slice(0, n)),
                seq_lens_tensor=self._cut(self.seq_lens_tensor, slice(0, n)),
                max_query_len=self.max_query_len, max_prefill_seq_len=self.max_prefill_seq_len,
                max_decode_query_len=0, max_decode_seq_len=0,
                query_start_loc=self._cut(self.query_start_loc, slice(0, n + 1)),
                seq_start_loc=self._cut(self.seq_start_loc, slice(0, n + 1)),
                context_lens_tensor=self._cut(self.context_lens_tensor, slice(0, n)),
                block_tables=self._cut(self.block_tables, slice(0, n)),
                use_cuda_graph=False,
                encoder_seq_lens=self.encoder_seq_lens, encoder_seq_lens_tensor=self.encoder_seq_lens_tensor,
                encoder_seq_start_loc=self.encoder_seq_start_loc, max_encoder_seq_len=self.max_encoder_seq_len,
                cross_slot_mapping=self.cross_slot_mapping, cross_block_tables=self.cross_block_tables)
        return self._cached_prefill_metadata

    @property
    def decode_metadata(self) -> Optional["MetaAttentionMetadata"]:
        """View on the sequences / tokens after the prefills; start locations re-based to 0."""
        if self.num_decode_tokens == 0:
            return None
        if self._cached_decode_metadata is None:
            n, nt = self.num_prefills, self.num_prefill_tokens
            qsl = None
            if self.query_start_loc is not None:
                qsl = self.query_start_loc[n:] - self.query_start_loc[n]
            self._cached_decode_metadata = MetaAttentionMetadata(
                num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.num_decode_tokens,
                slot_mapping=self._cut(self.slot_mapping, slice(nt, None)),
                multi_modal_placeholder_index_maps=None, enable_kv_scales_calculation=True,
                seq_lens=None, seq_lens_tensor=self._cut(self.seq_lens_tensor, slice(n, None)),
                max_decode_query_len=self.max_decode_query_len, max_query_len=self.max_query_len,
                max_prefill_seq_len=0, max_decode_seq_len=self.max_decode_seq_len,
                query_start_loc=qsl, seq_start_loc=self._cut(self.seq_start_loc, slice(n, None)),
                context_lens_tensor=None, block_tables=self._cut(self.block_tables, slice(n, None)),
                use_cuda_graph=self.use_cuda_graph,
                encoder_seq_lens=self.encoder_seq_lens, encoder_seq_lens_tensor=self.encoder_seq_lens_tensor,
                encoder_seq_start_loc=self.encoder_seq_start_loc, max_encoder_seq_len=self.max_encoder_seq_len,
                cross_slot_mapping=self.cross_slot_mapping, cross_block_tables=self.cross_block_tables)
        return self._cached_decode_metadata

    def advance_step(self, model_input, sampled_token_ids: Optional[torch.Tensor], block_size: int,
                     num_seqs: int, num_queries: int, turn_prefills_into_decodes: bool = False) -> None:
        """Multi-step decoding: bump every live sequence by one token in place, on device
        (reference ``meta_attention.py:338-409``; the device update restates vLLM's
        ``advance_step_flashattn``)."""
        if num_seqs != num_queries:
            assert num_seqs > num_queries and self.use_cuda_graph
        if turn_prefills_into_decodes:
            assert self.num_decode_tokens + self.num_prefills == num_seqs
            self.num_decode_tokens += self.num_prefills
            self.num_prefills = self.num_prefill_tokens = 0
            self.max_prefill_seq_len = 0
            self.max_query_len = 1
            self.slot_mapping = self.slot_mapping[:num_seqs]
        else:
            assert self.seq_lens is not None and self.max_decode_seq_len == max(self.seq_lens)
        assert self.num_prefills == 0 and self.num_prefill_tokens == 0
        assert self.num_decode_tokens == num_seqs and self.slot_mapping.shape == (num_seqs,)
        assert self.seq_lens is not None and len(self.seq_lens) == num_seqs
        assert self.seq_lens_tensor is not None and self.seq_lens_tensor.shape == (num_seqs,)
        assert self.max_query_len == 1 and self.max_prefill_seq_len == 0
        assert self.block_tables is not None and self.block_tables.shape[0] == num_seqs
        for i in range(num_queries):
            self.seq_lens[i] += 1
        self.max_decode_seq_len = max(self.seq_lens)
        q = num_queries
        tokens, positions = getattr(model_input, "input_tokens", None), getattr(model_input, "input_positions", None)
        if (self.seq_lens_tensor.is_cuda and self.slot_mapping.dtype == torch.int64 and self.seq_lens_tensor.dtype == torch.int32
                and self.block_tables.dtype == torch.int32 and self.block_tables.stride(1) == 1
                and all(t is None or (t.dtype == torch.int64 and t.is_contiguous()) for t in (tokens, positions))
                and (sampled_token_ids is None or (sampled_token_ids.dtype == torch.int64 and sampled_token_ids.is_contiguous()))):
            from .. import hip_ops       # one launch on the device
            hip_ops.advance_step(num_seqs, q, block_size, self.seq_lens_tensor, self.slot_mapping, self.block_tables,
                                 input_tokens=tokens, sampled_token_ids=sampled_token_ids, input_positions=positions)
            self._cached_decode_metadata = None
            return
        if sampled_token_ids is not None and getattr(model_input, "input_tokens", None) is not None:
            model_input.input_tokens[:q] = sampled_token_ids[:q].view(-1).to(model_input.input_tokens.dtype)
        pos = self.seq_lens_tensor[:q].long()              # position of the next token
        if getattr(model_input, "input_positions", None) is not None:
            model_input.input_positions[:q] = pos.to(model_input.input_positions.dtype)
        self.seq_lens_tensor[:q] += 1
        blk = torch.gather(self.block_tables[:q].long(), 1, (pos // block_size).unsqueeze(1)).squeeze(1)
        self.slot_mapping[:q] = blk * block_size + pos % block_size
        self._cached_decode_metadata = None


# --------------------------------------------------------------------------- builder
class MetaAttentionMetadataBuilder(AttentionMetadataBuilder):
    """Turns the scheduler's per-sequence lists into device tensors.  ``input_builder`` is vLLM's
    ``ModelInputForGPUBuilder`` (or any object with ``runner``, ``sliding_window``, ``block_size``,
    ``inter_data_list``, ``chunked_prefill_enabled``)."""

    def __init__(self, input_builder):
        self.input_builder = input_builder
        self.runner = input_builder.runner
        self.sliding_window = input_builder.sliding_window
        self.block_size = input_builder.block_size

    def prepare(self):
        self.slot_mapping: List[int] = []
        self.prefill_seq_lens: List[int] = []
        self.context_lens: List[int] = []
        self.block_tables: List[List[int]] = []
        self.curr_seq_lens: List[int] = []
        self.multimodal_placeholder_maps: Dict[str, Any] = {}
        self.num_prefills = 0
        self.num_prefill_tokens = 0
        self.num_decode_tokens = 0
        self.has_prefix_cache_hit = False

    def _add_seq_group(self, inter_data, chunked_prefill_enabled: bool, prefix_cache_hit: bool):
        is_prompt = inter_data.is_prompt
        block_tables = inter_data.block_tables
        profile_run = block_tables is None or all(not bt for bt in block_tables.values())
        for (seq_id, tokens, seq_len, curr_seq_len, query_len, context_len, window_blocks) in zip(
                inter_data.seq_ids, inter_data.input_tokens, inter_data.orig_seq_lens,
                inter_data.seq_lens, inter_data.query_lens, inter_data.context_lens,
                inter_data.curr_sliding_window_blocks):
            self.context_lens.append(context_len)
            if is_prompt:
                mm_maps = getattr(inter_data, "multi_modal_placeholder_maps", None)
                if mm_maps:
                    for modality, placeholders in mm_maps.items():
                        self.multimodal_placeholder_maps.setdefault(modality, type(placeholders)()).extend(placeholders)
                self.num_prefills += 1
                self.num_prefill_tokens += len(tokens)
                self.prefill_seq_lens.append(seq_len)
            else:
                self.num_decode_tokens += query_len
                self.curr_seq_lens.append(curr_seq_len)
            # Block table of the sequence: whole table on a prefix-cache hit; for decodes and
            # chunked prefills the table, trimmed to the window's last blocks when vLLM says so.
            table: List[int] = []
            if prefix_cache_hit:
                table = block_tables[seq_id]
            elif (chunked_prefill_enabled or not is_prompt) and block_tables is not None:
                table = block_tables[seq_id] if window_blocks == 0 else block_tables[seq_id][-window_blocks:]
            self.block_tables.append(list(table))
            start = slot_mapping_start_idx(is_prompt, query_len, self.sliding_window)
            slot_mapping_for_sequence(self.slot_mapping, None if profile_run else block_tables[seq_id],
                                      seq_len, context_len, start, self.block_size)

    def _get_graph_runner_block_tables(self, num_seqs: int, block_tables: List[List[int]]) -> torch.Tensor:
        graph_tables = self.runner.graph_block_tables[:num_seqs]
        max_blocks = graph_tables.shape[1]
        for i, table in enumerate(block_tables):
            if table:
                nb = min(len(table), max_blocks)   # look-ahead blocks beyond the graph's width are unused
                graph_tables[i, :nb] = table[:nb]
        return torch.from_numpy(graph_tables).to(device=self.runner.device, non_blocking=True)

    def build(self, seq_lens: List[int], query_lens: List[int], cuda_graph_pad_size: int,
              batch_size: int) -> MetaAttentionMetadata:
        prefix_cache_hit = any(d.prefix_cache_hit for d in self.input_builder.inter_data_list)
        for inter_data in self.input_builder.inter_data_list:
            self._add_seq_group(inter_data, self.input_builder.chunked_prefill_enabled, prefix_cache_hit)
        device = self.runner.device
        pin = getattr(self.runner, "pin_memory", False)
        use_captured_graph = cuda_graph_pad_size != -1
        max_query_len = max(query_lens)
        assert max_query_len > 0, f"query_lens: {query_lens}"
        decode_query_lens = query_lens[self.num_prefills:]
        max_decode_query_len = max(decode_query_lens) if decode_query_lens else 1
        num_decode_tokens = self.num_decode_tokens
        num_seqs = len(seq_lens)
        if use_captured_graph:
            self.slot_mapping.extend([PAD_SLOT_ID] * cuda_graph_pad_size)
            num_decode_tokens = batch_size - self.num_prefill_tokens
            block_tables = self._get_graph_runner_block_tables(num_seqs, self.block_tables)
        else:
            block_tables = _pad_block_tables(self.block_tables, device)
        return MetaAttentionMetadata(
            num_prefills=self.num_prefills,
            slot_mapping=_h2d(self.slot_mapping, torch.long, device, pin),
            num_prefill_tokens=self.num_prefill_tokens,
            num_decode_tokens=num_decode_tokens,
            seq_lens=seq_lens,
            multi_modal_placeholder_index_maps={m: p.index_map() for m, p in self.multimodal_placeholder_maps.items()},
            enable_kv_scales_calculation=True,
            seq_lens_tensor=_h2d(seq_lens, torch.int, device, pin),
            max_query_len=max_query_len,
            max_decode_query_len=max_decode_query_len,
            max_prefill_seq_len=max(self.prefill_seq_lens, default=0),
            max_decode_seq_len=max(self.curr_seq_lens, default=0),
            query_start_loc=_h2d(list(accumulate(query_lens, initial=0)), torch.int32, device, pin),
            seq_start_loc=_h2d(list(accumulate(seq_lens, initial=0)), torch.int32, device, pin),
            context_lens_tensor=_h2d(self.context_lens, torch.int, device, pin),
            block_tables=block_tables,
            use_cuda_graph=use_captured_graph,
        )


# --------------------------------------------------------------------------- impl
class MetaAttentionImpl(AttentionImpl):
    """Token layout of one step (reference ``meta_attention.py:612-639``):

        [meta tokens (warm-up only)] [prefill_0 ... prefill_N-1] [decode_0 ... decode_M-1 (+padding)]

    ``forward`` writes the KV of the step into the paged cache, then attends the prefill tokens
    (varlen; through the block table when there is cached context) and the decode tokens (paged),
    every query additionally seeing the ``num_meta_tokens`` meta keys ``key2/value2``.
    """

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int,
                 alibi_slopes: Optional[List[float]], sliding_window: Optional[int], kv_cache_dtype: str,
                 blocksparse_params: Optional[Dict[str, Any]] = None,
                 logits_soft_cap: Optional[float] = None, attn_type: str = AttentionType.DECODER,
                 kv_sharing_target_layer_name: Optional[str] = None, use_irope: bool = False,
                 num_meta_tokens: int = 128) -> None:
        if kv_sharing_target_layer_name is not None:
            raise NotImplementedError("KV sharing is not supported in V0 MetaAttention.")
        if blocksparse_params is not None:
            raise ValueError("MetaAttention does not support blocksparse attention.")
        if use_irope:
            logger.warning("Using irope in V0 is not supported yet, it will fall back to global "
                           "attention for long context.")
        if head_size not in MetaAttentionBackend.get_supported_head_sizes():
            raise ValueError(f"Head size {head_size} is not supported by MetaAttention. Supported head "
                             f"sizes are {MetaAttentionBackend.get_supported_head_sizes()}.")
        if kv_cache_dtype != "auto":
            raise NotImplementedError(f"MetaAttention does not support {kv_cache_dtype} kv-cache on this device.")
        if alibi_slopes is not None:
            raise NotImplementedError("MetaAttention (HIP) does not support ALiBi.")
        if logits_soft_cap:
            raise NotImplementedError("MetaAttention (HIP) does not support a logits soft cap.")
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_kv_heads
        self.alibi_slopes = None
        # flash-attn convention: (left, right); window W means W keys including the query itself
        self.sliding_window = (sliding_window - 1, 0) if sliding_window is not None else (-1, -1)
        self.kv_cache_dtype = kv_cache_dtype
        self.num_meta_tokens = num_meta_tokens
        self.logits_soft_cap = 0
        self.num_queries_per_kv = num_heads // num_kv_heads
        self.attn_type = attn_type

    def forward(self, layer, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                query2: Optional[torch.Tensor], key2: Optional[torch.Tensor],
                value2: Optional[torch.Tensor], kv_cache: torch.Tensor,
                attn_metadata: MetaAttentionMetadata, output: Optional[torch.Tensor] = None,
                output_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        """query ``[T,H,D]``, key/value ``[T,Hkv,D]`` (real tokens only), query2 ``[M,H,D]`` or None,
        key2/value2 ``[M,Hkv,D]``, kv_cache ``[2,nblk,bs,Hkv,D]`` (numel 0 on the profiling run).
        Writes ``output[(M if query2) + T, H, D]`` in place and returns it."""
        assert output is not None, "Output tensor must be provided."
        if output_scale is not None:
            raise NotImplementedError("fused output quantization is not supported in MetaAttention.")
        attn_type = self.attn_type
        if attn_type == AttentionType.ENCODER and not attn_metadata.is_all_encoder_attn_metadata_set:
            raise AttributeError("Encoder attention requires setting encoder metadata attributes.")
        if attn_type == AttentionType.ENCODER_DECODER and not attn_metadata.is_all_cross_attn_metadata_set:
            raise AttributeError("Encoder/decoder cross-attention requires setting cross-attention "
                                 "metadata attributes.")
        if attn_type != AttentionType.DECODER:
            raise NotImplementedError("MetaAttention (HIP) implements decoder self-attention only.")
        if layer is not None:
            ks, vs = getattr(layer, "_k_scale_float", 1.0), getattr(layer, "_v_scale_float", 1.0)
            assert ks == 1.0 and vs == 1.0, "key/value scales other than 1.0 need an fp8 cache (unsupported)"

        have_cache = kv_cache.numel() > 0
        # decode-only step with one token per sequence: the KV write rides inside the decode launch
        fuse_write = (have_cache and key is not None and value is not None
                      and attn_metadata.num_prefill_tokens == 0 and attn_metadata.num_decode_tokens > 0
                      and not (attn_metadata.max_decode_query_len or 0) > 1
                      and attn_metadata.slot_mapping.numel() == attn_metadata.num_decode_tokens)
        if have_cache and key is not None and value is not None and not fuse_write:
            hip_ops.reshape_and_cache(key, value, kv_cache[0], kv_cache[1], attn_metadata.slot_mapping.flatten())

        num_prefill_tokens = attn_metadata.num_prefill_tokens
        num_decode_tokens = attn_metadata.num_decode_tokens
        num_meta_prefill_tokens = num_prefill_tokens + (self.num_meta_tokens if query2 is not None else 0)
        decode_query = query[num_prefill_tokens:]
        prefill_query = query[:num_prefill_tokens]
        assert decode_query.shape[0] == num_decode_tokens, \
            f"decode_query.shape: {decode_query.shape}, num_decode_tokens: {num_decode_tokens}"
        window = self.sliding_window

        prefill_meta = attn_metadata.prefill_metadata
        if prefill_meta is not None:
            no_tables = (not have_cache or prefill_meta.block_tables is None
                         or prefill_meta.block_tables.numel() == 0)
            if no_tables:
                # whole prompts, no cached context: keys are the step's own key/value rows
                output[:num_meta_prefill_tokens] = metatoken_flash_attn_varlen_func(
                    q1=prefill_query, q2=query2, k1=key[:num_prefill_tokens], k2=key2,
                    v1=value[:num_prefill_tokens], v2=value2,
                    cu_seqlens_q=prefill_meta.seq_start_loc, cu_seqlens_k=prefill_meta.seq_start_loc,
                    max_seqlen_q=prefill_meta.max_prefill_seq_len, max_seqlen_k=prefill_meta.max_prefill_seq_len,
                    num_meta_tokens=self.num_meta_tokens, softmax_scale=self.scale, causal=True,
                    window_size=window)[0]
            else:
                # chunked / prefix prefill: the chunk is already in the cache; attend through the
                # block table over context + chunk, queries bottom-right aligned
                assert query2 is None, "meta-token queries are only computed on a cache-less warm-up call"
                output[:num_meta_prefill_tokens] = metatoken_flash_attn_varlen_func(
                    q1=prefill_query, q2=None, k1=kv_cache[0], k2=key2, v1=kv_cache[1], v2=value2,
                    cu_seqlens_q=prefill_meta.query_start_loc, cu_seqlens_k=prefill_meta.seq_start_loc,
                    max_seqlen_q=prefill_meta.max_query_len, max_seqlen_k=max(prefill_meta.seq_lens),
                    num_meta_tokens=self.num_meta_tokens, softmax_scale=self.scale, causal=True,
                    window_size=window, block_tables=prefill_meta.block_tables)[0]

        decode_meta = attn_metadata.decode_metadata
        if decode_meta is not None:
            if decode_meta.max_decode_query_len is not None and decode_meta.max_decode_query_len > 1:
                # several query tokens per sequence (speculative / multi-token decode)
                output[num_meta_prefill_tokens:] = metatoken_flash_attn_varlen_func(
                    q1=decode_query, q2=None, k1=kv_cache[0], k2=key2, v1=kv_cache[1], v2=value2,
                    cu_seqlens_q=decode_meta.query_start_loc, cu_seqlens_k=decode_meta.seq_start_loc,
                    max_seqlen_q=decode_meta.max_decode_query_len, max_seqlen_k=decode_meta.max_decode_seq_len,
                    num_meta_tokens=self.num_meta_tokens, softmax_scale=self.scale, causal=True,
                    window_size=window, block_tables=decode_meta.block_tables)[0]
            else:
                hip_ops.meta_swa_paged_decode(
                    decode_query, kv_cache[0], kv_cache[1], decode_meta.seq_lens_tensor,
                    decode_meta.block_tables, meta_key=key2, meta_value=value2, window_left=window[0],
                    scale=self.scale, max_seqlen_k=decode_meta.max_decode_seq_len or 0,
                    out=output[num_meta_prefill_tokens:],
                    new_key=key if fuse_write else None, new_value=value if fuse_write else None,
                    slot_mapping=attn_metadata.slot_mapping.flatten() if fuse_write else None)
        return output
