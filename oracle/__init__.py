"""CPU restatement of the Hymeta attention hot path.  TEST INFRASTRUCTURE ONLY.

This package is the parity checker for the HIP kernels.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under the product package does, and the product path raises
if its HIP library is missing rather than falling back to anything here.

Parity status: PINNED only where the reference's own code could run in the build
container, UNPINNED for the rest.  The reference ships no golden vectors, known-answer
tests or fixtures for this path, and its eager layers cannot be imported there
(``fla``, ``flash_attn`` and ``vllm`` are absent: an ordinary ``ModuleNotFoundError``;
their kernels are GPU-only).  What does load by file path -- ``hf_7B_model/norms.py``
(RMSNorm), the module-level RoPE / ``repeat_kv`` functions of
``hf_7B_model/window_attention.py``, both ``cache.py`` files and the config classes --
was run on seeded inputs (``tests/golden/make_ref_fixtures.py``,
``make_config_fixture.py``) and ``tests/test_ref_fixtures.py`` checks ``rms_norm_ref`` /
``rms_norm_tp_ref``, ``rope_neox_ref``, the GQA head mapping and ``oracle/cache.py``
against those outputs.  Every other function restates the reference math from its
source text and cites the file:line it follows; the other vectors under
``tests/golden`` are outputs of this restatement, not of the reference.
"""

from .gla import (  # noqa: F401
    gla_recurrent_ref,
    gla_decode_step_ref,
    gla_prefill_varlen_ref,
)
from .meta_swa import (  # noqa: F401
    meta_swa_dense_ref,
    lse_merge_ref,
    meta_swa_two_pass_ref,
    meta_swa_varlen_ref,
    meta_swa_paged_decode_ref,
)
from .paged import (  # noqa: F401
    reshape_and_cache_ref,
    slot_mapping_ref,
    gather_paged_ref,
    kv_cache_shape_ref,
)
