#!/usr/bin/env python
"""Dev probe: wall time of one SD1.5 UNet call (B = 8, 64x64 latent, bf16), eager launches, events over many calls."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import pipeline as P
from cremage_amd.synth import synth_input
dev = torch.device("cuda:0")
ldm = P.build_synthetic_ldm(device=dev, seed=1)
x = synth_input("x", (8, 4, 64, 64), 1).to(dev)
ctx = synth_input("c", (8, 77, 768), 1).to(dev)
t = torch.full((8,), 500.0, device=dev)
with torch.no_grad():
    for _ in range(3):
        ldm.model.diffusion_model(x, timesteps=t, context=ctx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ldm.model.diffusion_model(x, timesteps=t, context=ctx)
    e1.record()
    torch.cuda.synchronize()
print(f"unet call: {e0.elapsed_time(e1) / 20:.3f} ms  ({' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('CRG_'))})", flush=True)
