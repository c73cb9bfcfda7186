"""vLLM base classes when vLLM is importable, minimal stand-ins with the same attribute names
when it is not (unit tests and the kernel harness run without vLLM).  Nothing here touches
the device."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional

PAD_SLOT_ID = -1

try:  # pragma: no cover - exercised only where vLLM is installed
    from vllm.attention.backends.abstract import (AttentionBackend, AttentionImpl, AttentionMetadata,
                                                  AttentionMetadataBuilder, AttentionType)
    HAVE_VLLM = True
except Exception:  # vLLM absent
    HAVE_VLLM = False

    class AttentionType:
        DECODER = "decoder"
        ENCODER = "encoder"
        ENCODER_ONLY = "encoder_only"
        ENCODER_DECODER = "encoder_decoder"

    class AttentionBackend:
        accept_output_buffer: bool = False

    class AttentionImpl:
        pass

    @dataclass
    class AttentionMetadata:
        num_prefills: int
        num_prefill_tokens: int
        num_decode_tokens: int
        slot_mapping: Any
        multi_modal_placeholder_index_maps: Optional[Dict[str, Any]]
        enable_kv_scales_calculation: bool

    class AttentionMetadataBuilder:
        pass


# --------------------------------------------------------------------------- model-class interfaces
try:  # pragma: no cover - exercised only where vLLM is installed
    from vllm.model_executor.models.interfaces import HasInnerState, IsHybrid, SupportsPP, SupportsV0Only
    from vllm.sequence import IntermediateTensors
except Exception:  # vLLM absent: same names, same class-level flags vLLM's `supports_*` / `has_inner_state` checks read

    class HasInnerState:
        has_inner_state = True

    class IsHybrid:
        is_hybrid = True

    class SupportsV0Only:
        supports_v0_only = True

    class SupportsPP:
        supports_pp = True

    class IntermediateTensors:
        """Tensors a pipeline stage hands to the next one (vLLM ``IntermediateTensors``)."""

        def __init__(self, tensors: Dict[str, Any]):
            self.tensors = tensors

        def __getitem__(self, key):
            return self.tensors[key]

        def __setitem__(self, key, value):
            self.tensors[key] = value


@dataclass
class ModelBuildArgs:
    """What the model classes need from a ``VllmConfig`` (reference ``HymetaForCausalLM.__init__``,
    ``vllm_hymeta/models/modeling_hymeta.py:1386-1411``; 7B ``model_for_7B/modeling_gla_swa.py:510-530``)."""
    config: Any
    max_num_seqs: int
    quant: Optional[Dict[str, int]]
    tp_group: Any
    device: Any
    cache_config: Any
    prefix: str
    expert_parallel: bool = False      # ``--enable-expert-parallel`` (``parallel_config.enable_expert_parallel``)
    pp_rank: int = 0                   # this process's pipeline stage
    pp_size: int = 1
    pp_group: Any = None               # torch.distributed group of the stages (rank in it = stage), or None


def quant_args(quant_config) -> Optional[Dict[str, int]]:
    """GPTQ parameters of a vLLM quantisation config (``GPTQConfig`` / ``GPTQMarlinConfig``: ``weight_bits``,
    ``group_size``, ``desc_act``); ``None`` for un-quantised and for compressed-tensors W8A8 checkpoints, whose
    int8 tensors identify themselves when they are loaded."""
    if quant_config is None or not hasattr(quant_config, "weight_bits"):
        return None
    if getattr(quant_config, "desc_act", False):
        raise NotImplementedError("GPTQ checkpoints quantised with desc_act=True (act-order g_idx) are not supported")
    fmt = str(getattr(quant_config, "checkpoint_format", "gptq"))
    return dict(bits=int(quant_config.weight_bits), group_size=int(quant_config.group_size),
                zero_offset=0 if fmt.endswith("v2") else 1)


def tensor_parallel_group(tp_size: int):
    """The process group the layers' all-reduces run over: ``None`` for one rank; vLLM's tensor-parallel device
    group under vLLM; the world group when the program itself is the launcher and the world is the TP group."""
    if tp_size <= 1:
        return None
    if HAVE_VLLM:  # pragma: no cover - only where vLLM is installed
        from vllm.distributed import get_tp_group
        return get_tp_group().device_group
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() == tp_size:
        return dist.group.WORLD
    raise RuntimeError(f"tensor_parallel_size={tp_size} needs an initialised torch.distributed world of that size")


def pipeline_parallel_stage(parallel_config, pp_size: int):
    """(stage of this process, group of the stages): vLLM's pipeline group under vLLM; otherwise the duck-typed
    ``parallel_config.pipeline_parallel_rank`` / ``.pipeline_parallel_group`` (tests, launchers of their own)."""
    if HAVE_VLLM:  # pragma: no cover - only where vLLM is installed
        from vllm.distributed import get_pp_group
        g = get_pp_group()
        return int(g.rank_in_group), g.device_group
    rank = getattr(parallel_config, "pipeline_parallel_rank", None)
    if rank is None:
        raise RuntimeError(f"pipeline_parallel_size={pp_size} without vLLM needs parallel_config.pipeline_parallel_rank")
    return int(rank), getattr(parallel_config, "pipeline_parallel_group", None)


def model_build_args(vllm_config, prefix: str = "") -> ModelBuildArgs:
    mc = vllm_config.model_config
    config = mc.hf_config
    if isinstance(getattr(mc, "max_model_len", None), int):
        config.max_model_len = mc.max_model_len                      # reference :1401-1402
    pc = getattr(vllm_config, "parallel_config", None)
    tp = int(getattr(pc, "tensor_parallel_size", 1) or 1)
    pp = int(getattr(pc, "pipeline_parallel_size", 1) or 1)
    pp_rank, pp_group = 0, None
    if pp > 1:
        pp_rank, pp_group = pipeline_parallel_stage(pc, pp)
    device = getattr(getattr(vllm_config, "device_config", None), "device", None) or "cuda"
    return ModelBuildArgs(config=config, max_num_seqs=int(vllm_config.scheduler_config.max_num_seqs),
                          quant=quant_args(getattr(vllm_config, "quant_config", None)),
                          tp_group=tensor_parallel_group(tp), device=device,
                          cache_config=getattr(vllm_config, "cache_config", None), prefix=prefix,
                          expert_parallel=bool(getattr(pc, "enable_expert_parallel", False)),
                          pp_rank=pp_rank, pp_size=pp, pp_group=pp_group)
