"""The 7B "gla_swa" model on the HIP kernels: embedding, 28 decoder blocks (even layers gated
linear attention, odd layers sliding-window attention), final norm, LM head.

Reference: ``GLAswaModel`` / ``GLAswaForCausalLM``
(``vllm_hymeta/model_for_7B/modeling_gla_swa.py:309-724``).  Same class names, parameter names
(``packed_modules_mapping``, ``load_weights`` routing), state-cache calls and keyword arguments
(``request_ids_to_seq_ids``, ``finished_requests_ids``, ``seqlen_agnostic_capture_inputs``).

What differs in mechanism:
* the step's attention metadata and the per-layer KV caches are taken from vLLM's forward context
  when vLLM drives the model, and may be passed explicitly (``attn_metadata=``, ``kv_caches=``)
  when it does not (tests, benchmarks);
* a new sequence's state slot is not zeroed on the host (``_clear_prefill_cache``, ``:391-414``,
  three device reads per prefill): the prefill kernel starts from zeros where
  ``context_lens == 0``;
* tensors are ingested in checkpoint format (bf16, GPTQ, W8A8) and cut for the rank by
  ``vllm_hymeta.loader``; there are no vLLM parameter classes underneath.
"""

from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Set, Tuple

import torch

from .. import loader
from .._compat import HasInnerState, IntermediateTensors, IsHybrid, SupportsPP, SupportsV0Only, model_build_args
from ..linear import HymetaLinear
from ..lm_head import LmHeadMixin, lm_head_logits
from ..models.attention import bind_kv_cache, current_attn_metadata, forward_context, standalone_prefix
from ..models.block import GLAswaBlock, HymetaRMSNorm
from ..models.modeling_hymeta import pp_layer_range
from ..models.moe import HymetaMLP
from ..parallel import exchange_enabled, raise_on_exchange_timeout, shard_heads, tp_rank_and_size
from .gla_attention import GatedLinearAttention
from .gla_cache import GLACacheManager, GLACacheParams
from .window_attention import SlidingWindowAttention, rope_cos_sin_cache, swa_attention_layer

GLAswaGLU = HymetaMLP     # vllm_hymeta/model_for_7B/modeling_gla_swa.py:60-131


class GLAswaModel(torch.nn.Module):
    def __init__(self, config, max_num_seqs: int = 256, quant: Optional[dict] = None, tp_group=None, device="cuda",
                 state_per_kv_head: bool = True, cache_config=None, prefix: Optional[str] = None,
                 pp_rank: int = 0, pp_size: int = 1):
        """``prefix``: name of this module in the whole model (``"model"`` under vLLM); sliding-window layer ``i``
        registers its attention layer as ``<prefix>.layers.<i>.attn.attn`` (reference ``:201,257,346``), the name
        under which the engine binds that layer's paged KV cache.  ``None``: a private name.

        ``quant``: ``{"bits", "group_size", "zero_offset"}`` of a GPTQ checkpoint (ignored for bf16 /
        W8A8 tensors, which identify themselves by dtype).

        ``state_per_kv_head``: keep one GLA state per key/value head instead of one per query head.  The
        reference expands k, v and the gate to the query heads (``model_for_7B/gla_attention.py:246-282``)
        and caches ``[layers, slots, H, D, D]`` (``modeling_gla_swa.py:391-414``); the 7 query heads of
        a group then hold 7 identical states.  Keeping one cuts the state traffic of a decode step and
        the cache size by that factor with bit-identical outputs; ``False`` restores the reference's shape.

        ``pp_rank`` / ``pp_size``: this process's pipeline stage.  A stage holds layers ``[start_layer, end_layer)``
        (vLLM's partition, which the reference's ``make_layers`` call uses, ``:343-347``), the embedding on the first
        stage and the final norm on the last (``:324-331,383-389``); its state cache and attention layers are those of
        its own layers (the reference sizes every stage's cache for all layers, ``:348-355``)."""
        super().__init__()
        self.config = config
        self.prefix = standalone_prefix("model") if prefix is None else prefix
        self.device = torch.device(device)
        self.tp_group = tp_group
        self.tp_rank, self.tp_size = tp_rank_and_size(tp_group)
        self.quant = dict(bits=4, group_size=128, zero_offset=1)
        self.quant.update(quant or {})
        self.shard = shard_heads(config.num_attention_heads, config.num_key_value_heads, self.tp_size, self.tp_rank)
        self.pp_rank, self.pp_size = int(pp_rank), int(pp_size)
        self.start_layer, self.end_layer = pp_layer_range(config.num_hidden_layers, self.pp_rank, self.pp_size)
        self.is_first_stage, self.is_last_stage = self.pp_rank == 0, self.pp_rank == self.pp_size - 1
        # the sliding-window layers' attention objects exist from construction on: the engine looks them up by
        # name when it allocates the KV cache; load_weights builds the rest of each layer around them
        self.attn_layers = torch.nn.ModuleDict({
            str(i): swa_attention_layer(self.shard, config.head_dim, config.sliding_window,
                                        f"{self.prefix}.layers.{i}.attn.attn", cache_config=cache_config)
            for i in sorted(config.attn_layers) if self.start_layer <= i < self.end_layer})
        self.gla_layers = [i for i in range(self.start_layer, self.end_layer) if i not in config.attn_layers]
        state_heads = self.shard.num_kv_heads if state_per_kv_head else self.shard.num_heads
        self.cache_shape = (len(self.gla_layers), max_num_seqs, state_heads, config.head_dim, config.head_dim)
        self.gla_cache = GLACacheManager(dtype=torch.bfloat16, cache_shape=self.cache_shape, device=self.device)
        max_pos = min(config.max_position_embeddings, getattr(config, "max_model_len", config.max_position_embeddings))
        self.register_buffer("cos_sin_cache", rope_cos_sin_cache(config.head_dim, max_pos, config.rope_theta, device=self.device),
                             persistent=False)
        self.embeddings: Optional[torch.Tensor] = None          # [vocab, hidden] bf16, set by load_weights
        self.layers = torch.nn.ModuleList()                      # this stage's blocks, filled by load_weights
        self.norm: Optional[HymetaRMSNorm] = None

    def get_input_embeddings(self, input_ids: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.embedding(input_ids, self.embeddings)

    def forward(self, input_ids: Optional[torch.Tensor], positions: torch.Tensor, intermediate_tensors=None,
                inputs_embeds: Optional[torch.Tensor] = None, attn_metadata=None,
                kv_caches: Optional[List[torch.Tensor]] = None, **kwargs) -> Optional[torch.Tensor]:
        """Under an engine the step's metadata comes from the forward context and every sliding-window layer's
        paged KV cache is the one the engine bound to its attention layer.  Callers without an engine pass
        ``attn_metadata`` and, once, ``kv_caches`` (one ``[2, nblk, bs, Hkv, D]`` tensor per sliding-window layer, in
        layer order): they are set / bound here and the same path runs."""
        if kv_caches is not None:
            for layer, kv in zip(self.attn_layers.values(), kv_caches):
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
        kwargs.setdefault("request_ids_to_seq_ids", {})
        kwargs.setdefault("finished_requests_ids", [])
        gla_cache_tensors, state_indices_tensor = self.gla_cache.current_run_tensors(**kwargs)
        params = GLACacheParams(gla_cache_tensors, state_indices_tensor)
        if self.is_first_stage:
            hidden_states = self.get_input_embeddings(input_ids) if inputs_embeds is None else inputs_embeds
        else:
            hidden_states = intermediate_tensors["hidden_states"]      # the residual stream ("residual" arrives as zeros)

        gla_index = 0
        residual = None      # the residual stream travels beside the blocks' outputs (fused add + norm)
        for i, layer in enumerate(self.layers, self.start_layer):
            if i in self.config.attn_layers:
                hidden_states, residual = layer(hidden_states, carry_residual=True, residual=residual, positions=positions,
                                                attn_metadata=attn_metadata)
            else:
                p = params.at_layer_idx(gla_index)
                gla_index += 1
                hidden_states, residual = layer(hidden_states, carry_residual=True, residual=residual,
                                                gla_cache=p.gla_cache, state_indices_tensor=p.state_indices_tensor,
                                                attn_metadata=attn_metadata)
        if not self.is_last_stage:
            # the next stage starts as the model does, from the residual stream itself (reference ``:487-495``:
            # ``hidden_states`` = the stream, ``residual`` = zeros).  The sum is the one the next block's fused add + norm
            # would have formed (K slices in order, one bf16 rounding, + residual, one rounding)
            if not isinstance(hidden_states, torch.Tensor):
                hidden_states = hidden_states.tensor()
            if residual is not None:
                hidden_states = hidden_states + residual
            return IntermediateTensors({"hidden_states": hidden_states, "residual": torch.zeros_like(hidden_states)})
        return self.norm(hidden_states, residual)[0] if residual is not None else self.norm(hidden_states)

    def make_empty_intermediate_tensors(self, batch_size: int, dtype: torch.dtype, device) -> IntermediateTensors:
        """Reference ``:585-598``."""
        shape = (batch_size, self.config.hidden_size)
        return IntermediateTensors({"hidden_states": torch.zeros(shape, dtype=dtype, device=device),
                                    "residual": torch.zeros(shape, dtype=dtype, device=device)})


class GLAswaForCausalLM(LmHeadMixin, torch.nn.Module, HasInnerState, IsHybrid, SupportsPP, SupportsV0Only):
    packed_modules_mapping = {
        "qkv_proj": ["q_proj", "k_proj", "v_proj"],
        "gate_up_proj": ["gate_proj", "up_proj"],
    }

    def __init__(self, config=None, max_num_seqs: int = 256, quant: Optional[dict] = None, tp_group=None, device="cuda",
                 state_per_kv_head: bool = True, *, vllm_config=None, prefix: Optional[str] = None,
                 pp_rank: int = 0, pp_size: int = 1):
        """vLLM constructs the class as ``GLAswaForCausalLM(vllm_config=..., prefix=...)`` (reference
        ``modeling_gla_swa.py:510-530``); the keyword form builds the same model without an engine.  The stage of a
        pipeline comes from vLLM's pipeline group under vLLM, from ``pp_rank`` / ``pp_size`` in the keyword form."""
        super().__init__()
        cache_config = None
        if vllm_config is not None:
            args = model_build_args(vllm_config, prefix or "")
            config, max_num_seqs, quant, tp_group, device, cache_config = (
                args.config, args.max_num_seqs, args.quant, args.tp_group, args.device, args.cache_config)
            pp_rank, pp_size = args.pp_rank, args.pp_size
            model_prefix = f"{prefix}.model" if prefix else "model"
        else:
            if config is None:
                raise TypeError("GLAswaForCausalLM needs vllm_config= or a model config")
            model_prefix = None if prefix is None else (f"{prefix}.model" if prefix else "model")
        self.config = config
        self.CONCAT_FFN = True
        self.unpadded_vocab_size = config.vocab_size
        self.model = GLAswaModel(config, max_num_seqs=max_num_seqs, quant=quant, tp_group=tp_group, device=device,
                                 state_per_kv_head=state_per_kv_head, cache_config=cache_config, prefix=model_prefix,
                                 pp_rank=pp_rank, pp_size=pp_size)
        self.lm_head = None             # [vocab, hidden] on the last stage; reads as fp32 (the reference's lm_head.float())
        self.kv_cache = [torch.tensor([]) for _ in self.model.attn_layers]

    def copy_inputs_before_cuda_graphs(self, input_buffers, **kwargs):
        """The engine calls this on the host before every replay of a captured step: also the place where a timeout of
        the tensor-parallel group's one-shot exchange in an earlier step becomes an error (``raise_on_exchange_timeout``)."""
        if self.model.tp_group is not None and exchange_enabled(self.model.tp_group):
            raise_on_exchange_timeout(self.model.tp_group)
        return self.model.gla_cache.copy_inputs_before_cuda_graphs(input_buffers, **kwargs)

    def get_seqlen_agnostic_capture_inputs(self, batch_size: int):
        return self.model.gla_cache.get_seqlen_agnostic_capture_inputs(batch_size)

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
        ``sampling_metadata.selected_token_indices`` first)."""
        idx = getattr(sampling_metadata, "selected_token_indices", None)
        if idx is not None:
            hidden_states = hidden_states.index_select(0, idx)
        return lm_head_logits(hidden_states, self._lm_head)

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> Set[str]:
        """Checkpoint tensors (any of bf16 / GPTQ / W8A8 per linear) -> this rank's modules.  Returns
        the destination parameter names, as the reference's ``load_weights`` does."""
        cfg, m = self.config, self.model
        q = m.quant
        dev = m.device
        loaded: Set[str] = set()
        parts: Dict[str, Dict[object, Dict[str, torch.Tensor]]] = {}    # module -> shard id -> this rank's tensors
        plain: Dict[str, torch.Tensor] = {}
        H, KVH, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        sh = m.shard
        heads = loader.head_cuts(sh, D)
        row = loader.Cut("in", parts=m.tp_size, index=m.tp_rank)
        col = loader.Cut("out", parts=m.tp_size, index=m.tp_rank)

        def cut_of(module: str, shard_id) -> loader.Cut:
            if module.endswith(".attn.qkv_proj"):
                return heads[shard_id]
            if module.endswith(".gk_proj.linear1"):
                return heads["k"]
            if module.endswith(".o_proj") or module.endswith(".down_proj"):
                return row
            if module.endswith(".gate_up_proj"):
                return col
            return loader.Cut()

        # the checkpoint is a stream: every tensor is cut for this rank as it arrives and only the piece is kept
        for name, tensor in weights:
            route = loader.route_7b(name, cfg.num_hidden_layers)
            if route is None:
                continue
            # a pipeline stage keeps its own layers; the embedding belongs to the first stage, the final norm and the
            # lm_head to the last (the reference skips them with is_pp_missing_parameter, ``:601-724``)
            layer_of = loader._layer_of(route.param)
            if layer_of is not None and not m.start_layer <= layer_of < m.end_layer:
                continue
            if route.param == "model.embeddings.weight" and not m.is_first_stage:
                continue
            if route.param in ("model.norm.weight", "lm_head.weight") and not m.is_last_stage:
                continue
            loaded.add(route.param)
            module, _, leaf = route.param.rpartition(".")
            if leaf in ("weight", "bias", "qweight", "qzeros", "scales", "g_idx", "weight_scale") and (
                    "proj" in module or module == "lm_head" or module.endswith("embeddings")):
                piece = loader.cut_leaf(leaf, tensor, cut_of(module, route.shard_id), q["bits"], q["group_size"])
                if piece is not None:
                    parts.setdefault(module, {}).setdefault(route.shard_id, {})[leaf] = loader.own(piece, dev)
            else:
                plain[route.param] = tensor.to(dev)

        def lin(module: str, shard=None) -> Dict[str, torch.Tensor]:
            return parts[module].pop(shard)

        def make(t: Dict[str, torch.Tensor]) -> HymetaLinear:
            return loader.make_linear(t, q["bits"], q["group_size"], q["zero_offset"], device=dev)

        if m.is_first_stage:
            m.embeddings = lin("model.embeddings")["weight"].to(torch.bfloat16)
        if m.is_last_stage:
            m.norm = HymetaRMSNorm(cfg.hidden_size, cfg.norm_eps, plain["model.norm.weight"])
            self.lm_head = lin("lm_head")["weight"]
        m.layers = torch.nn.ModuleList()
        for i in range(m.start_layer, m.end_layer):
            p = f"model.layers.{i}"
            qkv = loader.concat_out_features([lin(f"{p}.attn.qkv_proj", s) for s in ("q", "k", "v")])
            o_proj = make(lin(f"{p}.attn.o_proj"))
            if i in cfg.attn_layers:
                attn = SlidingWindowAttention(cfg.hidden_size, H, KVH, cfg.sliding_window, make(qkv), o_proj, m.cos_sin_cache,
                                              shard=sh, tp_group=m.tp_group, attn=m.attn_layers[str(i)])
            else:
                attn = GatedLinearAttention(cfg.hidden_size, H, KVH, make(qkv), make(lin(f"{p}.attn.gk_proj.linear0")),
                                            make(lin(f"{p}.attn.gk_proj.linear1")), o_proj,
                                            plain[f"{p}.attn.g_norm.weight"].to(torch.bfloat16), cfg.norm_eps,
                                            cfg.gate_logit_normalizer, shard=sh, tp_group=m.tp_group)
            mlp = GLAswaGLU(make(loader.concat_out_features([lin(f"{p}.mlp.gate_up_proj", 0), lin(f"{p}.mlp.gate_up_proj", 1)])),
                            make(lin(f"{p}.mlp.down_proj")), tp_group=m.tp_group)
            m.layers.append(GLAswaBlock(HymetaRMSNorm(cfg.hidden_size, cfg.norm_eps, plain[f"{p}.attn_norm.weight"]), attn,
                                        HymetaRMSNorm(cfg.hidden_size, cfg.norm_eps, plain[f"{p}.mlp_norm.weight"]), mlp))
        return loaded
