#!/usr/bin/env python3
"""One 32k-token prompt through the 7B model a few times (for rocprofv3 --kernel-trace: where the time to first token goes).
Env QUANT (bf16 | gptq4 | w8a8), CTX."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
quant = os.environ.get("QUANT", "bf16")
L = int(os.environ.get("CTX", "32768"))
model = bench.build_model_7b(dev, quant, max_num_seqs=1)
pre = bench.PrefillRunner(model, L, 8, dev)
pre.prefill()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    pre.prefill()
    torch.cuda.synchronize()
    print(f"prefill {L} tokens: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
