"""The 70B "hybrid" model on the HIP kernels: embedding, decoder blocks (each with a softmax branch
and a GLA branch run on the same input, 128 shared meta tokens in front of every sequence, dense
MLP or mixture of experts with a shared expert), final norm, LM head.

Reference: ``HymetaModel`` / ``HymetaForCausalLM``
(``vllm_hymeta/models/modeling_hymeta.py:1084-1691``).  Same class names, checkpoint routing,
state-cache calls and keyword arguments.

What differs in mechanism:
* the meta tokens are run through the layers ONCE, when the weights are loaded
  (``prepare_meta_caches``): every layer keeps their roped K/V and their GLA end state in the
  shared caches, and a serving step never carries the 128 extra rows.  The reference re-inserts
  them into the batch until a forward-call counter passes a threshold (``:1160-1164,
  1278-1292, 1372-1373``), which makes its first requests depend on how many warm-up runs vLLM
  happened to make;
* a new sequence's state slot is seeded from the meta state by the prefill kernel
  (``context_lens == 0``), not zeroed on the host (``_clear_prefill_cache``, ``:1202-1227``);
* tensors are ingested in checkpoint format and cut for the rank by ``vllm_hymeta.loader``.
Pipeline parallelism (the reference's 8-GPU recipe is TP4 x PP2, ``readme.md:113-114``; ``:1103-1110, 1132,
1190-1193, 1281-1296, 1356-1364, 1471-1486``): a stage holds a contiguous range of layers (vLLM's partition rule), the
first one the embedding and the meta tokens, the last one the final norm and the lm_head; a stage hands the next
``IntermediateTensors`` with ``hidden_states`` = the residual stream and ``residual`` = zeros, as the reference does.
Because the meta tokens run at load time here, the stages also hand their meta rows on once, when the weights are loaded
(``prepare_meta_caches``): over the pipeline group when it is a torch.distributed group, or through the caller.
"""

from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Set, Tuple

import torch

from .. import loader
from .._compat import HasInnerState, IntermediateTensors, IsHybrid, SupportsPP, SupportsV0Only, model_build_args
from ..linear import HymetaLinear, LinearColumns
from ..lm_head import LmHeadMixin, lm_head_logits
from ..model_for_7B.window_attention import rope_cos_sin_cache
from ..parallel import (exchange_enabled, raise_on_exchange_timeout, recv_from_stage, send_to_stage, shard_heads,
                        tp_rank_and_size)
from .attention import MetaAttention, bind_kv_cache, current_attn_metadata, forward_context, standalone_prefix
from .block import HybridBlock, HymetaRMSNorm
from .hybrid_attention import HLinearAttention, IntraHybridAttention, MetaFlashAttention
from .hymeta_cache import HymetaCacheManager, HymetaCacheParams
from .moe import HymetaMLP, HymetaMoE, expert_range


def _layer_experts(config, layer_idx: int) -> int:
    n = getattr(config, "num_layer_experts", 1)          # reference :1116-1122
    return n[layer_idx] if isinstance(n, list) else int(n)


def _shared_intermediate(config, layer_idx: int) -> int:
    s = getattr(config, "shared_intermediate_size", 0)   # reference :981-986
    if isinstance(s, list):
        return s[layer_idx] if len(s) > layer_idx else 0
    return int(s)


def _layer_window(config, layer_idx: int) -> Optional[int]:
    w = getattr(config, "interleaved_sliding_window", None)   # reference :711-722
    if w is None:
        return None
    if isinstance(w, int):
        return w
    if isinstance(w, list):
        return w[layer_idx % len(w)]
    raise ValueError(f"{type(w)} is not supported.")


def pp_layer_range(num_layers: int, pp_rank: int, pp_size: int) -> Tuple[int, int]:
    """Layers ``[start, end)`` of one pipeline stage: equal shares, the remainder one each to the stages counted back from
    the second to last (vLLM's ``get_pp_indices``, which the reference's ``make_layers`` call uses, ``:1132``): the last
    stage, which also holds the lm_head, is the last to get an extra layer."""
    if not 0 <= pp_rank < pp_size:
        raise ValueError(f"pipeline rank {pp_rank} outside a pipeline of {pp_size} stages")
    parts = [num_layers // pp_size] * pp_size
    for i in range(2, num_layers % pp_size + 2):
        parts[-i if i <= pp_size else -1] += 1
    start = sum(parts[:pp_rank])
    return start, start + parts[pp_rank]


class HymetaModel(torch.nn.Module):
    def __init__(self, config, max_num_seqs: int = 256, quant: Optional[dict] = None, tp_group=None, device="cuda",
                 cache_config=None, prefix: Optional[str] = None, expert_parallel: bool = False,
                 pp_rank: int = 0, pp_size: int = 1, pp_group=None):
        """``expert_parallel`` (``--enable-expert-parallel`` of the reference's serving commands, ``readme.md:115,167``):
        the routed experts of a mixture layer are dealt whole to the ranks of the tensor-parallel group (16 / tp each)
        instead of every expert being cut along its intermediate size; every rank still sees every token, computes the
        pairs routed to its experts, and the layer's all-reduce sums the ranks -- what vLLM's ``FusedMoE`` does under
        expert parallelism without data parallelism (``models/modeling_hymeta.py:312-324``).
        ``prefix``: name of this module in the whole model (``"model"`` under vLLM); the softmax branch of layer
        ``i`` registers its attention layer as ``<prefix>.layers.<i>.attn.vanilla_attn.attn`` (reference ``:742,
        843,955,1133``), the name under which the engine binds that layer's paged KV cache.  ``None``: a private
        name (models built directly, several per process)."""
        super().__init__()
        self.config = config
        self.prefix = standalone_prefix("model") if prefix is None else prefix
        self.device = torch.device(device)
        self.tp_group = tp_group
        self.tp_rank, self.tp_size = tp_rank_and_size(tp_group)
        self.expert_parallel = bool(expert_parallel) and self.tp_size > 1
        # pipeline stage: layers [start_layer, end_layer); pp_group: the torch.distributed group of the stages (rank =
        # stage) over which the meta rows travel at load time, or None when the caller carries them
        self.pp_rank, self.pp_size, self.pp_group = int(pp_rank), int(pp_size), pp_group
        self.start_layer, self.end_layer = pp_layer_range(config.num_hidden_layers, self.pp_rank, self.pp_size)
        self.is_first_stage, self.is_last_stage = self.pp_rank == 0, self.pp_rank == self.pp_size - 1
        self.quant = dict(bits=4, group_size=128, zero_offset=1)
        self.quant.update(quant or {})
        self.use_lower_bound = config.use_lower_bound
        self.num_meta_tokens = config.num_meta_tokens
        self.shard = shard_heads(config.num_attention_heads, config.num_key_value_heads, self.tp_size, self.tp_rank)
        L, D = self.end_layer - self.start_layer, config.head_dim      # the caches hold the stage's own layers
        self.cache_shape = (L, max_num_seqs, self.shard.num_heads, D, D)
        self.meta_linear_cache_shape = (L, self.shard.num_heads, D, D)
        self.meta_fattn_cache_shape = (L, 2, self.num_meta_tokens, self.shard.num_kv_heads, D)
        self.hymeta_cache = HymetaCacheManager(torch.bfloat16, self.cache_shape, self.meta_linear_cache_shape,
                                               self.meta_fattn_cache_shape, device=self.device)
        max_pos = min(config.max_position_embeddings, getattr(config, "max_model_len", config.max_position_embeddings))
        # real token p is roped at p + num_meta_tokens (reference :1314-1323): the table holds those rows too
        self.register_buffer("cos_sin_cache", rope_cos_sin_cache(D, max_pos + self.num_meta_tokens, int(config.rope_theta),
                                                                 device=self.device),
                             persistent=False)
        # the attention layer objects exist from construction on (the engine looks them up by name when it
        # allocates the KV cache, before and independently of the weights); load_weights builds the rest around them
        self.attn_layers = torch.nn.ModuleList([
            MetaAttention(self.shard.num_heads, D, D ** -0.5, self.shard.num_kv_heads, cache_config=cache_config,
                          per_layer_sliding_window=_layer_window(config, i) or None,
                          prefix=f"{self.prefix}.layers.{i}.attn.vanilla_attn.attn", num_meta_tokens=self.num_meta_tokens)
            for i in range(self.start_layer, self.end_layer)])
        self.embeddings: Optional[torch.Tensor] = None
        self.layers = torch.nn.ModuleList()
        self.norm: Optional[HymetaRMSNorm] = None
        self.lower_bounds: Optional[torch.Tensor] = None     # [L, hidden] fp32 AFTER softmax/cumsum (:1309-1311)
        self.meta_tokens: Optional[torch.Tensor] = None      # [M, hidden] bf16
        self.meta_hidden_out: Optional[torch.Tensor] = None  # the meta rows after this stage's layers (for the next stage)
        self.has_meta_cache = False

    def get_input_embeddings(self, input_ids: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.embedding(input_ids, self.embeddings)

    def _lower_bound(self, layer_idx: int) -> Optional[torch.Tensor]:
        return self.lower_bounds[layer_idx % 7] if self.use_lower_bound else None

    @torch.no_grad()
    def prepare_meta_caches(self, meta_hidden: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """The meta tokens' pass through the stage's layers: fills ``meta_linear_cache`` and ``meta_fattn_cache`` of each
        layer; the rows' own outputs only feed the next layer.  A stage other than the first starts from the rows the
        stage before it produced (``meta_hidden``; with a torch.distributed pipeline group they are received from that
        stage, and sent on to the next); returns the rows after this stage's layers (also kept as ``meta_hidden_out``)."""
        if self.num_meta_tokens == 0:
            self.has_meta_cache = True
            return None
        import torch.distributed as dist
        over_group = self.pp_size > 1 and self.pp_group is not None and dist.is_available() and dist.is_initialized()
        if self.is_first_stage:
            h = self.meta_tokens
        else:
            if meta_hidden is None and over_group:
                like = torch.empty(self.num_meta_tokens, self.config.hidden_size, dtype=torch.bfloat16, device=self.device)
                meta_hidden = recv_from_stage(like, self.pp_group, self.pp_rank - 1)
            if meta_hidden is None:
                raise RuntimeError("a pipeline stage after the first needs the meta rows of the stage before it: pass "
                                   "prepare_meta_caches(previous_stage.meta_hidden_out), or build the stages over a "
                                   "torch.distributed pipeline group")
            h = meta_hidden.to(self.device, torch.bfloat16)
        _, meta_linear, meta_fattn = self.hymeta_cache.cache
        for j, layer in enumerate(self.layers):
            lb = self._lower_bound(self.start_layer + j)
            h, _ = layer(h, attn_fn=lambda x, j=j, layer=layer, lb=lb: layer.attn.prepare_meta_state(
                x, meta_linear[j], meta_fattn[j], lb))
        self.meta_hidden_out = h
        if not self.is_last_stage and over_group:
            send_to_stage(h, self.pp_group, self.pp_rank + 1)
        self.has_meta_cache = True
        return h

    def forward(self, input_ids: Optional[torch.Tensor], positions: torch.Tensor, intermediate_tensors=None,
                inputs_embeds: Optional[torch.Tensor] = None, attn_metadata=None,
                kv_caches: Optional[List[torch.Tensor]] = None, **kwargs) -> Optional[torch.Tensor]:
        """Under an engine the step's metadata comes from the forward context and every layer's paged KV cache is
        the one the engine bound to its attention layer (reference ``:1243-1246``).  Callers without an engine pass
        ``attn_metadata`` and, once, ``kv_caches`` (one ``[2, nblk, bs, Hkv, D]`` tensor per layer): they are set /
        bound here and the same path runs."""
        if kv_caches is not None:
            for layer, kv in zip(self.attn_layers, kv_caches):
                bind_kv_cache(layer, kv)
        if attn_metadata is not None:
            with forward_context(attn_metadata):
                return self._forward(input_ids, positions, inputs_embeds, attn_metadata, intermediate_tensors, **kwargs)
        attn_metadata = current_attn_metadata()
        if attn_metadata is None:
            return None
        return self._forward(input_ids, positions, inputs_embeds, attn_metadata, intermediate_tensors, **kwargs)

    def _forward(self, input_ids, positions, inputs_embeds, attn_metadata, intermediate_tensors=None, **kwargs):
        if self.is_first_stage:
            if input_ids is not None and inputs_embeds is not None:
                raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
            if input_ids is None and inputs_embeds is None:
                raise ValueError("You have to specify either input_ids or inputs_embeds")
        elif intermediate_tensors is None:
            raise ValueError("a pipeline stage after the first needs the intermediate_tensors of the stage before it")
        if not self.has_meta_cache:
            raise RuntimeError("meta caches are empty: load_weights() / prepare_meta_caches() must run before forward()")
        kwargs.setdefault("request_ids_to_seq_ids", {})
        kwargs.setdefault("finished_requests_ids", [])
        params = HymetaCacheParams(*self.hymeta_cache.current_run_tensors(**kwargs))
        if self.is_first_stage:
            hidden_states = self.get_input_embeddings(input_ids) if inputs_embeds is None else inputs_embeds
        else:
            hidden_states = intermediate_tensors["hidden_states"]      # the residual stream ("residual" arrives as zeros)
        residual = None      # the residual stream travels beside the blocks' outputs (fused add + norm)
        for j, layer in enumerate(self.layers):
            c = params.at_layer_idx(j)
            hidden_states, residual = layer(hidden_states, carry_residual=True, residual=residual, positions=positions,
                                            kv_cache=None, hymeta_cache=c.hymeta_cache,
                                            meta_linear_cache=c.meta_linear_cache, meta_fattn_cache=c.meta_fattn_cache,
                                            state_indices_tensor=c.state_indices_tensor, attn_metadata=attn_metadata,
                                            lower_bound=self._lower_bound(self.start_layer + j))
        if not self.is_last_stage:
            # the next stage starts as the model does: from the residual stream itself.  The sum is the one the next
            # block's fused add + norm would have formed (K slices in order, one bf16 rounding, + residual, one rounding)
            if not isinstance(hidden_states, torch.Tensor):
                hidden_states = hidden_states.tensor()
            if residual is not None:
                hidden_states = hidden_states + residual
            return IntermediateTensors({"hidden_states": hidden_states, "residual": torch.zeros_like(hidden_states)})
        return self.norm(hidden_states, residual)[0] if residual is not None else self.norm(hidden_states)

    def make_empty_intermediate_tensors(self, batch_size: int, dtype: torch.dtype, device) -> IntermediateTensors:
        """Reference ``:1471-1486``."""
        shape = (batch_size, self.config.hidden_size)
        return IntermediateTensors({"hidden_states": torch.zeros(shape, dtype=dtype, device=device),
                                    "residual": torch.zeros(shape, dtype=dtype, device=device)})


class HymetaForCausalLM(LmHeadMixin, torch.nn.Module, HasInnerState, IsHybrid, SupportsPP, SupportsV0Only):
    packed_modules_mapping = {
        "qkv_proj": ["q_proj", "k_proj", "v_proj"],
        "gate_up_proj": ["gate_proj", "up_proj"],
    }

    def __init__(self, config=None, max_num_seqs: int = 256, quant: Optional[dict] = None, tp_group=None, device="cuda",
                 *, vllm_config=None, prefix: Optional[str] = None, expert_parallel: bool = False,
                 pp_rank: int = 0, pp_size: int = 1, pp_group=None):
        """vLLM constructs the class as ``HymetaForCausalLM(vllm_config=..., prefix=...)`` (reference ``:1386-1411``):
        the HF config, ``scheduler_config.max_num_seqs`` (state slots), the quantisation parameters, the
        tensor-parallel group and the cache config are read from ``vllm_config``.  The keyword form
        ``HymetaForCausalLM(config, max_num_seqs=..., quant=..., tp_group=..., device=...)`` builds the same model
        without an engine."""
        super().__init__()
        cache_config = None
        if vllm_config is not None:
            args = model_build_args(vllm_config, prefix or "")
            config, max_num_seqs, quant, tp_group, device, cache_config = (
                args.config, args.max_num_seqs, args.quant, args.tp_group, args.device, args.cache_config)
            expert_parallel = bool(getattr(args, "expert_parallel", False))
            pp_rank, pp_size, pp_group = args.pp_rank, args.pp_size, args.pp_group
            model_prefix = f"{prefix}.model" if prefix else "model"          # maybe_prefix(prefix, "model"), :1410
        else:
            if config is None:
                raise TypeError("HymetaForCausalLM needs vllm_config= or a model config")
            model_prefix = None if prefix is None else (f"{prefix}.model" if prefix else "model")
        self.config = config
        self.CONCAT_FFN = True
        if not hasattr(config, "sliding_window"):
            config.sliding_window = None
        self.unpadded_vocab_size = config.vocab_size
        self.model = HymetaModel(config, max_num_seqs=max_num_seqs, quant=quant, tp_group=tp_group, device=device,
                                 cache_config=cache_config, prefix=model_prefix, expert_parallel=expert_parallel,
                                 pp_rank=pp_rank, pp_size=pp_size, pp_group=pp_group)
        self.lm_head = None             # [vocab, hidden]; reads as fp32 (:1429), see lm_head.py; the last stage's only
        self.kv_cache = [torch.tensor([]) for _ in range(config.num_hidden_layers)]

    def copy_inputs_before_cuda_graphs(self, input_buffers, **kwargs):
        """The engine calls this on the host before every replay of a captured step: also the place where a timeout of
        the tensor-parallel group's one-shot exchange in an earlier step becomes an error (``raise_on_exchange_timeout``)."""
        if self.model.tp_group is not None and exchange_enabled(self.model.tp_group):
            raise_on_exchange_timeout(self.model.tp_group)
        return self.model.hymeta_cache.copy_inputs_before_cuda_graphs(input_buffers, **kwargs)

    def get_seqlen_agnostic_capture_inputs(self, batch_size: int):
        return self.model.hymeta_cache.get_seqlen_agnostic_capture_inputs(batch_size)

    def get_input_embeddings(self, input_ids: torch.Tensor) -> torch.Tensor:
        return self.model.get_input_embeddings(input_ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None, **kwargs):
        tp_group = self.model.tp_group
        if tp_group is not None and exchange_enabled(tp_group) and not torch.cuda.is_current_stream_capturing():
            raise_on_exchange_timeout(tp_group)      # eager steps: the check a replayed step gets from the hook above
        return self.model(input_ids, positions, intermediate_tensors, inputs_embeds, **kwargs)

    def make_empty_intermediate_tensors(self, batch_size: int, dtype: torch.dtype, device) -> IntermediateTensors:
        return self.model.make_empty_intermediate_tensors(batch_size, dtype, device)

    def compute_logits(self, hidden_states: torch.Tensor, sampling_metadata=None) -> torch.Tensor:
        """fp32 logits of the rows the sampler asks for (vLLM's ``LogitsProcessor`` prunes the hidden states to
        ``sampling_metadata.selected_token_indices`` first; reference ``:1459-1469``)."""
        idx = getattr(sampling_metadata, "selected_token_indices", None)
        if idx is not None:
            hidden_states = hidden_states.index_select(0, idx)
        return lm_head_logits(hidden_states, self._lm_head)

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> Set[str]:
        """Checkpoint tensors -> this rank's modules, then the meta tokens' pass.  Returns the
        destination parameter names, as the reference's ``load_weights`` does."""
        cfg, m = self.config, self.model
        q, dev = m.quant, m.device
        loaded: Set[str] = set()
        parts: Dict[str, Dict[object, Dict[str, torch.Tensor]]] = {}       # linear module -> shard id -> this rank's tensors
        experts: Dict[int, Dict[int, Dict[str, torch.Tensor]]] = {}        # layer -> expert -> projection -> this rank's slice
        plain: Dict[str, torch.Tensor] = {}
        proj_of = {"w1": "gate_proj", "w3": "up_proj", "w2": "down_proj"}
        H, KVH, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        sh = m.shard
        hs = slice(sh.head_start * D, (sh.head_start + sh.num_heads) * D)
        heads = loader.head_cuts(sh, D)
        row = loader.Cut("in", parts=m.tp_size, index=m.tp_rank)
        col = loader.Cut("out", parts=m.tp_size, index=m.tp_rank)

        def cut_of(module: str, shard_id) -> loader.Cut:
            if module.endswith(".qkv_proj"):
                return heads[shard_id]
            if module.endswith(".out_proj") or module.endswith(".down_proj"):
                return row
            if module.endswith(".gate_up_proj"):
                return col
            return loader.Cut()

        # the checkpoint is a stream: every tensor is cut for this rank as it arrives and only the piece is kept
        for name, tensor in weights:
            route = loader.route_70b(name, cfg.num_hidden_layers)
            if route is None:
                continue
            # a pipeline stage keeps its own layers; the embedding and the meta tokens belong to the first stage, the final
            # norm and the lm_head to the last (the reference skips them with is_pp_missing_parameter, :1538-1651)
            layer_of = loader._layer_of(route.param)
            if layer_of is not None and not m.start_layer <= layer_of < m.end_layer:
                continue
            if route.param in ("model.embeddings.weight", "model.meta_tokens") and not m.is_first_stage:
                continue
            if route.param in ("model.norm.weight", "lm_head.weight") and not m.is_last_stage:
                continue
            loaded.add(route.param)
            if route.expert_id is not None:
                _, leaf = loader.expert_leaf(route.param)
                if leaf not in ("weight", "weight_scale", "qweight", "qzeros", "scales", "g_idx"):
                    raise ValueError(f"{name}: unexpected expert tensor {leaf!r}")
                if leaf == "weight" and tensor.dtype not in (torch.bfloat16, torch.float16, torch.float32, torch.int8):
                    raise ValueError(f"{name}: expert weights must be floating point or int8 (W8A8), not {tensor.dtype}")
                layer = loader._layer_of(name)
                if m.expert_parallel:      # whole experts, dealt to the ranks in contiguous equal shares
                    lo, hi = expert_range(_layer_experts(cfg, layer), m.tp_size, m.tp_rank)
                    if not lo <= route.expert_id < hi:
                        continue
                    cut = loader.Cut()
                else:
                    cut = row if route.shard_id == "w2" else col          # experts are cut along their intermediate size
                piece = loader.cut_leaf(leaf, tensor, cut, q["bits"], q["group_size"])
                if piece is None:          # g_idx: checked (act-order is refused), not kept
                    continue
                piece = loader.own(piece, dev)
                if leaf == "weight" and piece.is_floating_point():
                    piece = piece.to(torch.bfloat16)
                experts.setdefault(layer, {}).setdefault(route.expert_id, {}).setdefault(proj_of[route.shard_id], {})[leaf] = piece
                continue
            module, _, leaf = route.param.rpartition(".")
            if "proj" in module and "norm" not in module:
                piece = loader.cut_leaf(leaf, tensor, cut_of(module, route.shard_id), q["bits"], q["group_size"])
                if piece is not None:
                    parts.setdefault(module, {}).setdefault(route.shard_id, {})[leaf] = loader.own(piece, dev)
            elif module.endswith(".attn.norm1") or module.endswith(".attn.norm2"):
                plain[route.param] = loader.own(tensor[hs], dev)        # the group norms act on this rank's head columns
            else:
                plain[route.param] = tensor.to(dev)

        def lin(module: str, shard=None) -> Dict[str, torch.Tensor]:
            return parts[module].pop(shard)

        def make(t: Dict[str, torch.Tensor]) -> HymetaLinear:
            return loader.make_linear(t, q["bits"], q["group_size"], q["zero_offset"], device=dev)

        def mlp(prefix: str, reduce_results: bool = True) -> HymetaMLP:
            gate_up = loader.concat_out_features([lin(f"{prefix}.gate_up_proj", 0), lin(f"{prefix}.gate_up_proj", 1)])
            return HymetaMLP(make(gate_up), make(lin(f"{prefix}.down_proj")), tp_group=m.tp_group, reduce_results=reduce_results)

        if m.is_first_stage:
            m.embeddings = plain["model.embeddings.weight"].to(torch.bfloat16)
        if m.is_last_stage:
            m.norm = HymetaRMSNorm(cfg.hidden_size, cfg.norm_eps, plain["model.norm.weight"])
            self.lm_head = plain["lm_head.weight"]
        if m.use_lower_bound:
            lb = plain["model.lower_bounds"].float().softmax(0)
            m.lower_bounds = (lb.cumsum(0) - lb[0]).contiguous()
        if m.num_meta_tokens > 0 and m.is_first_stage:
            m.meta_tokens = plain["model.meta_tokens"].to(torch.bfloat16).contiguous()
        m.layers = torch.nn.ModuleList()
        for i in range(m.start_layer, m.end_layer):
            p = f"model.layers.{i}"

            # both branches project the same normed input: one linear for the six projections (the softmax branch's
            # q | k | v columns, then the GLA branch's), each branch a view of its columns
            both = make(loader.concat_out_features([lin(f"{p}.attn.{branch}.qkv_proj", s)
                                                    for branch in ("vanilla_attn", "linear_attn") for s in ("q", "k", "v")]))
            width = (sh.num_heads + 2 * sh.num_kv_heads) * cfg.head_dim
            vanilla = MetaFlashAttention(cfg.hidden_size, H, KVH, LinearColumns(both, 0, width), m.cos_sin_cache,
                                         _layer_window(cfg, i), num_meta_tokens=m.num_meta_tokens, shard=sh,
                                         attn=m.attn_layers[i - m.start_layer])
            linear = HLinearAttention(cfg.hidden_size, H, KVH, LinearColumns(both, width, 2 * width), i,
                                      num_meta_tokens=m.num_meta_tokens, shard=sh)
            attn = IntraHybridAttention(vanilla, linear, make(lin(f"{p}.attn.out_proj")), plain[f"{p}.attn.norm1.weight"],
                                        plain[f"{p}.attn.norm2.weight"], cfg.norm_eps, tp_group=m.tp_group, merged_qkv=both)
            n_experts = _layer_experts(cfg, i)
            if n_experts == 1:
                ffn = mlp(f"{p}.mlp")
            else:
                mine = experts.pop(i)
                local = expert_range(n_experts, m.tp_size, m.tp_rank) if m.expert_parallel else None
                held = [mine[e] for e in (range(*local) if local else range(n_experts))]       # this rank's
                scales = {}
                if "qweight" in held[0]["gate_proj"]:
                    w13, w2 = loader.stack_experts_gptq(held, q["bits"], q["group_size"], q["zero_offset"])
                elif held[0]["gate_proj"]["weight"].dtype == torch.int8:
                    w13, scales["w13_scale"], w2, scales["w2_scale"] = loader.stack_experts_w8a8(held)
                else:
                    w13, w2 = loader.stack_experts([{proj: t["weight"] for proj, t in e.items()} for e in held])
                del mine, held
                # the shared expert's partial sums ride in the experts' all-reduce
                shared = mlp(f"{p}.shared_mlp", reduce_results=False) if _shared_intermediate(cfg, i) > 0 else None
                gate = plain[f"{p}.block_sparse_moe.gate.weight"]
                if gate.dtype == torch.int8:       # a W8A8 recipe over every Linear quantises the router as well
                    scales["gate_scale"] = plain[f"{p}.block_sparse_moe.gate.weight_scale"].reshape(-1).float()
                else:
                    gate = gate.to(torch.bfloat16)
                ffn = HymetaMoE(gate, w13, w2, top_k=cfg.num_experts_per_topk, renormalize=False, shared_mlp=shared,
                                group=m.tp_group, local_experts=local, **scales)
            m.layers.append(HybridBlock(HymetaRMSNorm(cfg.hidden_size, cfg.norm_eps, plain[f"{p}.attn_norm.weight"]), attn,
                                        HymetaRMSNorm(cfg.hidden_size, cfg.norm_eps, plain[f"{p}.mlp_norm.weight"]), ffn))
        # the meta tokens' pass: a later stage of a pipeline without a torch.distributed group waits for its caller to
        # bring the rows of the stage before it (prepare_meta_caches(previous.meta_hidden_out))
        if dev.type == "cuda" and (m.is_first_stage or m.pp_group is not None or m.num_meta_tokens == 0):
            m.prepare_meta_caches()
        return loaded
