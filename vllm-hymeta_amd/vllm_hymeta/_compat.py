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
