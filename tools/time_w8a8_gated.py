import os, sys
sys.path.insert(0, "vllm-hymeta_amd"); sys.path.insert(0, "tools")
import torch
from tune_quant import bench
from vllm_hymeta import hip_ops
from vllm_hymeta.linear import HymetaLinear
dev = torch.device("cuda")
K, N = 3584, 37888
wq = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev)
ws = (torch.rand(N, device=dev) * 0.004 + 0.0005)
lin = HymetaLinear.from_w8a8(wq, ws)
for M in (256, 2048, 8192, 32768):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    a = bench(lambda: lin.forward_silu_mul(x), iters=5)
    b = bench(lambda: hip_ops.silu_and_mul(lin(x)), iters=5)
    print(f"M={M:6d} gate_up + silu_mul: fused {a:8.1f} us   two launches {b:8.1f} us", flush=True)
