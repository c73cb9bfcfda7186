"""Platform plugin class: tells vLLM which attention backend to use on ROCm.

Reference: ``vllm_hymeta/platform.py:10-30`` (a CUDA platform).  Here the base is vLLM's ROCm
platform; like the reference it must not initialise the device at import time, because the
plugin hook runs before Ray / multiprocessing workers fork.
"""

from __future__ import annotations

import logging

logger = logging.getLogger(__name__)

META_BACKEND = "vllm_hymeta.attention.meta_attention.MetaAttentionBackend"
# The reference runs its 7B model on vLLM's stock backend, selected with VLLM_ATTENTION_BACKEND=XFORMERS
# (platform.py:24-27).  Here both families run on the meta-token backend (the 7B sliding-window layers are
# MetaAttention layers with no meta tokens), so the switch is only kept for non-Hymeta models served by the
# same installation: it hands back vLLM's own ROCm backend.
STOCK_BACKEND = "vllm.attention.backends.rocm_flash_attn.ROCmFlashAttentionBackend"

try:  # pragma: no cover - only where vLLM is installed
    from vllm.platforms import PlatformEnum, _Backend
    from vllm.platforms.rocm import RocmPlatform as _Base
    _XFORMERS = _Backend.XFORMERS
    _ENUM = PlatformEnum.ROCM
except Exception:
    _Base = object
    _XFORMERS = "XFORMERS"
    _ENUM = "rocm"


class HymetaRocmPlatform(_Base):
    """ROCm platform that routes attention to the Hymeta meta-token backend."""

    _enum = _ENUM

    @classmethod
    def get_attn_backend_cls(cls, selected_backend, head_size, dtype, kv_cache_dtype, block_size,
                             use_v1, use_mla) -> str:
        name = getattr(selected_backend, "name", selected_backend)
        if selected_backend == _XFORMERS or name == "XFORMERS":
            logger.info("Hymeta: vLLM's stock ROCm attention backend selected (non-Hymeta models only: both Hymeta "
                        "families need the meta-token backend, unset VLLM_ATTENTION_BACKEND for them).")
            return STOCK_BACKEND
        logger.info("Hymeta: MetaAttentionBackend selected (Hymeta 70B and 7B).")
        return META_BACKEND
