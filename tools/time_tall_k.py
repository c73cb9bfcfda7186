"""Time of the tall kernel against K for a fixed wide N (intercept = fixed cost per launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vllm-hymeta_amd"))
sys.path.insert(0, os.path.dirname(__file__))
from vllm_hymeta import hip_ops  # noqa: E402
from time_tall_gemm import timed  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 37888
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator(device="cpu").manual_seed(5)
for K in (512, 1024, 2048, 3584, 7168):
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda() for _ in range(3)]
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    st = {"i": 0}

    def ours():
        st["i"] = (st["i"] + 1) % 3
        hip_ops.bf16_gemm(x, ws[st["i"]])

    print(f"N={N} M={M} K={K}: {timed(ours):.1f} us", flush=True)
