#!/usr/bin/env python3
"""One eager decode step of the 70B model from the synthetic checkpoint (unsharded, a few layers): per-layer statistics of
the hidden states and the experts hit.  Env QUANT, LAYERS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vllm-hymeta_amd"))
import torch  # noqa: E402
import bench as B1  # noqa: E402
import bench_tp  # noqa: E402

dev = torch.device("cuda", 0)
quant = os.environ.get("QUANT", "gptq4")
layers = int(os.environ.get("LAYERS", "9"))
model, cfg = bench_tp.build_model_70b(dev, quant, 64, None, layers)
runner = B1.DecodeRunner(model, 64, 4096, 16, dev, seed=0)
from vllm_hymeta.models.moe import HymetaMoE  # noqa: E402


def stat(name):
    def hook(mod, inputs, output):
        t = output[0] if isinstance(output, tuple) else output
        if isinstance(mod, HymetaMoE):
            x = inputs[0].reshape(-1, inputs[0].shape[-1])
            lg = torch.nn.functional.linear(x, mod.gate_weight).float() if mod.w8a8_gate is None else mod.w8a8_gate(x)[:, :16].float()
            print(f"   router: experts hit {int(lg.argmax(-1).unique().numel())}, logits std over tokens {float(lg.std(0).mean()):.3f}", flush=True)
        if not isinstance(t, torch.Tensor):
            print(f"{name}: (partial sums, not a tensor)", flush=True)
            return
        t = t.float()
        print(f"{name}: finite {bool(torch.isfinite(t).all())} mean|x| {float(t.abs().mean()):.4f} row-to-row std {float(t.std(0).mean()):.4f}", flush=True)
    return hook


for i, blk in enumerate(model.model.layers):
    blk.attn.register_forward_hook(stat(f"layer {i} attn"))
    blk.mlp.register_forward_hook(stat(f"layer {i} ffn "))
runner.eager_step()
torch.cuda.synchronize()
