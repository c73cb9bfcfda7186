"""The attention layer object that owns a KV cache and the two opaque custom ops through which
compiled / graph-captured model code reaches the backend.

Reference: ``MetaAttention`` (``vllm_hymeta/models/attention.py:25-283``) and the ops
``vllm::unified_meta_attention`` / ``vllm::unified_meta_attention_with_output`` (``:317-409``).
Same constructor arguments, attributes (``impl``, ``kv_cache``, ``layer_name``, ``use_output``,
``_k_scale`` ...), call signature ``forward(query, key, value, query2, key2, value2, output_shape)``
and op schemas (``with_output`` mutates ``output``).

What differs in mechanism: the backend is this package's ``MetaAttentionBackend`` directly (the
reference asks vLLM's selector, which the platform plugin answers with the same class); the layer
registry and the step's metadata come from vLLM's forward context when vLLM drives the model and
from ``forward_context()`` of this module when it does not (tests, other serving loops); KV-transfer
connectors (``:286-314``) are vLLM's and are not restated.
"""

from __future__ import annotations

import contextlib
import threading
from typing import Any, Dict, List, Optional

import torch

from .._compat import HAVE_VLLM, AttentionType
from ..attention.meta_attention import MetaAttentionBackend

# ------------------------------------------------------------------ forward context without vLLM
_LAYERS: Dict[str, "MetaAttention"] = {}      # static_forward_context / no_compile_layers
class _Step(threading.local):
    """The step a thread is running: per thread, so that ranks which share a process (each on its own thread and
    stream) do not see each other's metadata."""
    attn_metadata: Any = None
    virtual_engine: int = 0


_STEP = _Step()


@contextlib.contextmanager
def forward_context(attn_metadata, virtual_engine: int = 0):
    """The step's attention metadata for every ``MetaAttention`` layer called inside the block
    (what ``vllm.forward_context.set_forward_context`` provides under vLLM)."""
    saved = (_STEP.attn_metadata, _STEP.virtual_engine)
    _STEP.attn_metadata, _STEP.virtual_engine = attn_metadata, virtual_engine
    try:
        yield
    finally:
        _STEP.attn_metadata, _STEP.virtual_engine = saved


_standalone = [0]


def standalone_prefix(tag: str) -> str:
    """A layer-name prefix no other model or layer of this process uses: for models and layers built
    directly (tests, benchmarks, other serving loops), where nobody hands out the names.  vLLM passes the
    reference's names (``model.layers.<i>...``) and rejects duplicates itself."""
    _standalone[0] += 1
    return f"standalone{_standalone[0]}.{tag}"


def current_attn_metadata():
    """The step's attention metadata (vLLM's forward context, or ``forward_context()`` of this module)."""
    return _context()[0]


def bind_kv_cache(layer: "MetaAttention", kv_cache: torch.Tensor, virtual_engine: int = 0) -> None:
    """What vLLM's ``bind_kv_cache`` does for every registered attention layer after it has allocated the
    paged cache: ``layer.kv_cache[virtual_engine]`` becomes the layer's ``[2, num_blocks, block_size, Hkv, D]``
    tensor.  Callers without vLLM (tests, benchmarks) bind their own tensors with it."""
    while len(layer.kv_cache) <= virtual_engine:
        layer.kv_cache.append(torch.tensor([]))
    layer.kv_cache[virtual_engine] = kv_cache


def _context():
    """``(attn_metadata, virtual_engine, layers by name)`` of the current step: what a caller set with
    ``forward_context()`` if anything, else vLLM's forward context."""
    if HAVE_VLLM and _STEP.attn_metadata is None:   # pragma: no cover - only where vLLM is installed
        from vllm.forward_context import get_forward_context
        ctx = get_forward_context()
        return ctx.attn_metadata, ctx.virtual_engine, ctx.no_compile_layers
    return _STEP.attn_metadata, _STEP.virtual_engine, _LAYERS


class MetaAttention(torch.nn.Module):
    """1. stores the step's key / value rows in the paged KV cache, 2. attends (window or full,
    plus the always-visible meta keys ``key2`` / ``value2``; ``query2`` = the meta tokens' own
    rows on a warm-up call), 3. returns ``[(M if query2) + T, H * D]``."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[List[float]] = None, cache_config=None, quant_config=None,
                 blocksparse_params: Optional[Dict[str, Any]] = None, logits_soft_cap: Optional[float] = None,
                 per_layer_sliding_window: Optional[int] = None, use_mla: bool = False, prefix: str = "",
                 attn_type: str = AttentionType.DECODER, kv_sharing_target_layer_name: Optional[str] = None,
                 num_meta_tokens: int = 128, pipeline_parallel_size: int = 1, **extra_impl_args) -> None:
        super().__init__()
        if per_layer_sliding_window is not None:
            sliding_window = per_layer_sliding_window
        elif cache_config is not None:
            sliding_window = cache_config.sliding_window
        else:
            sliding_window = None
        kv_cache_dtype = cache_config.cache_dtype if cache_config is not None else "auto"
        self.calculate_kv_scales = bool(cache_config.calculate_kv_scales) if cache_config is not None else False
        if num_kv_heads is None:
            num_kv_heads = num_heads
        if use_mla:
            raise NotImplementedError("MetaAttention has no MLA form")
        if quant_config is not None and getattr(quant_config, "get_quant_method", None) is not None and \
                kv_cache_dtype not in ("auto", "bfloat16"):
            raise ValueError("a quantised KV cache is not supported by the meta-token attention backend")

        self.kv_cache_dtype = kv_cache_dtype
        self._k_scale = torch.tensor(1.0, dtype=torch.float32)
        self._v_scale = torch.tensor(1.0, dtype=torch.float32)
        self._q_scale = torch.tensor(1.0, dtype=torch.float32)
        self._prob_scale = torch.tensor(1.0, dtype=torch.float32)
        self._k_scale_float = 1.0
        self._v_scale_float = 1.0
        self.use_mla = use_mla
        self.num_heads, self.head_size, self.num_kv_heads = num_heads, head_size, num_kv_heads
        self.sliding_window = sliding_window
        self.num_meta_tokens = num_meta_tokens

        attn_backend = MetaAttentionBackend
        self.impl = attn_backend.get_impl_cls()(num_heads, head_size, scale, num_kv_heads, alibi_slopes, sliding_window,
                                                kv_cache_dtype, blocksparse_params, logits_soft_cap, attn_type,
                                                kv_sharing_target_layer_name, num_meta_tokens=num_meta_tokens,
                                                **extra_impl_args)
        self.backend = attn_backend.get_name()
        self.dtype = torch.get_default_dtype()
        self.use_direct_call = False            # ROCm is "cuda-alike": always through the opaque ops
        self.use_output = attn_backend.accept_output_buffer
        if HAVE_VLLM:   # pragma: no cover - only where vLLM is installed
            from vllm.config import get_current_vllm_config
            cfg = get_current_vllm_config()
            registry = cfg.compilation_config.static_forward_context
            pipeline_parallel_size = cfg.parallel_config.pipeline_parallel_size
            if prefix in registry:
                raise ValueError(f"Duplicate layer name: {prefix}")
            registry[prefix] = self
            _LAYERS[prefix] = self          # for callers that set their own forward_context() beside vLLM
        else:
            if prefix in _LAYERS:
                raise ValueError(f"Duplicate layer name: {prefix}")
            _LAYERS[prefix] = self
        self.layer_name = prefix
        self.attn_type = attn_type
        self.kv_sharing_target_layer_name = kv_sharing_target_layer_name
        # placeholder per virtual engine, replaced when the KV cache is bound
        self.kv_cache = [torch.tensor([]) for _ in range(pipeline_parallel_size)]

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                query2: Optional[torch.Tensor] = None, key2: Optional[torch.Tensor] = None,
                value2: Optional[torch.Tensor] = None, output_shape: Optional[torch.Size] = None) -> torch.Tensor:
        if self.calculate_kv_scales:
            raise NotImplementedError("key/value scale calculation needs an fp8 KV cache (unsupported)")
        if output_shape is None:
            output_shape = query.shape if query2 is None else torch.Size([query.shape[0] + query2.shape[0], query.shape[1]])
        output = torch.empty(output_shape, dtype=query.dtype, device=query.device)
        hidden_size = output_shape[-1]
        query = query.view(-1, self.num_heads, self.head_size)
        output = output.view(-1, self.num_heads, self.head_size)
        if key is not None:
            key = key.view(-1, self.num_kv_heads, self.head_size)
        if value is not None:
            value = value.view(-1, self.num_kv_heads, self.head_size)
        torch.ops.vllm.unified_meta_attention_with_output(query, key, value, output, self.layer_name,
                                                          query2=query2, key2=key2, value2=value2)
        return output.view(-1, hidden_size)

    def extra_repr(self) -> str:
        return (f"head_size={self.impl.head_size}, num_heads={self.impl.num_heads}, num_kv_heads={self.impl.num_kv_heads}, "
                f"scale={self.impl.scale}, backend={self.impl.__class__.__name__}")

    def process_weights_after_loading(self, act_dtype: torch.dtype):
        if hasattr(self.impl, "process_weights_after_loading"):
            self.impl.process_weights_after_loading(act_dtype)


def release_layer(prefix: str) -> None:
    """Forget a layer name of the vLLM-free registry (tests that build models repeatedly)."""
    _LAYERS.pop(prefix, None)


def release_layers(prefix: str) -> None:
    """Forget every layer whose name starts with ``prefix`` (a whole model's layers)."""
    for name in [n for n in _LAYERS if n.startswith(prefix)]:
        del _LAYERS[name]


def registered_layers() -> Dict[str, "MetaAttention"]:
    """The vLLM-free registry (name -> layer), as vLLM's ``static_forward_context`` holds it under an engine."""
    return dict(_LAYERS)


# ------------------------------------------------------------------ the opaque ops
def _views(layer: MetaAttention, q2, k2, v2):
    v = lambda t, h: None if t is None else t.view(-1, h, layer.head_size)
    return v(q2, layer.num_heads), v(k2, layer.num_kv_heads), v(v2, layer.num_kv_heads)


def unified_meta_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, layer_name: str,
                           query2: Optional[torch.Tensor] = None, key2: Optional[torch.Tensor] = None,
                           value2: Optional[torch.Tensor] = None) -> torch.Tensor:
    attn_metadata, virtual_engine, layers = _context()
    self = layers[layer_name]
    q = query.view(-1, self.num_heads, self.head_size)
    q2, k2, v2 = _views(self, query2, key2, value2)
    rows = q.shape[0] + (q2.shape[0] if q2 is not None else 0)
    output = torch.empty(rows, self.num_heads, self.head_size, dtype=query.dtype, device=query.device)
    self.impl.forward(self, q, key.view(-1, self.num_kv_heads, self.head_size), value.view(-1, self.num_kv_heads, self.head_size),
                      q2, k2, v2, self.kv_cache[virtual_engine], attn_metadata, output)
    return output.view(rows, -1)


def unified_meta_attention_fake(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, layer_name: str,
                                query2: Optional[torch.Tensor] = None, key2: Optional[torch.Tensor] = None,
                                value2: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows = query.shape[0] + (query2.shape[0] if query2 is not None else 0)
    return query.new_empty((rows,) + tuple(query.shape[1:])).contiguous()


def unified_meta_attention_with_output(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, output: torch.Tensor,
                                       layer_name: str, query2: Optional[torch.Tensor] = None,
                                       key2: Optional[torch.Tensor] = None, value2: Optional[torch.Tensor] = None) -> None:
    attn_metadata, virtual_engine, layers = _context()
    self = layers[layer_name]
    q2, k2, v2 = _views(self, query2, key2, value2)
    self.impl.forward(self, query, key, value, q2, k2, v2, self.kv_cache[virtual_engine], attn_metadata, output)


def unified_meta_attention_with_output_fake(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, output: torch.Tensor,
                                            layer_name: str, query2: Optional[torch.Tensor] = None,
                                            key2: Optional[torch.Tensor] = None, value2: Optional[torch.Tensor] = None) -> None:
    return


def _register_ops() -> None:
    if hasattr(torch.ops.vllm, "unified_meta_attention_with_output"):
        return
    if HAVE_VLLM:   # pragma: no cover - only where vLLM is installed
        from vllm.platforms import current_platform
        from vllm.utils import direct_register_custom_op
        direct_register_custom_op(op_name="unified_meta_attention", op_func=unified_meta_attention, mutates_args=[],
                                  fake_impl=unified_meta_attention_fake, dispatch_key=current_platform.dispatch_key)
        direct_register_custom_op(op_name="unified_meta_attention_with_output", op_func=unified_meta_attention_with_output,
                                  mutates_args=["output"], fake_impl=unified_meta_attention_with_output_fake,
                                  dispatch_key=current_platform.dispatch_key)
        return
    lib = torch.library.Library("vllm", "FRAGMENT")
    lib.define("unified_meta_attention(Tensor query, Tensor key, Tensor value, str layer_name, Tensor? query2=None, "
               "Tensor? key2=None, Tensor? value2=None) -> Tensor")
    lib.define("unified_meta_attention_with_output(Tensor query, Tensor key, Tensor value, Tensor(a!) output, str layer_name, "
               "Tensor? query2=None, Tensor? key2=None, Tensor? value2=None) -> ()")
    lib.impl("unified_meta_attention", unified_meta_attention, "CompositeExplicitAutograd")
    lib.impl("unified_meta_attention_with_output", unified_meta_attention_with_output, "CompositeExplicitAutograd")
    lib._register_fake("unified_meta_attention", unified_meta_attention_fake)
    lib._register_fake("unified_meta_attention_with_output", unified_meta_attention_with_output_fake)
    globals()["_LIB"] = lib      # the registration lives as long as the module


_register_ops()
