#!/usr/bin/env python
"""Dev probe: GPU time of the LayerNorm / GroupNorm launches at the UNet's shapes (events over many reps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
dev = "cuda:0"
def timeit(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows, dim in [(32768, 320), (8192, 640), (2048, 1280), (512, 1280), (16384, 640), (4096, 1280)]:
    x = torch.randn(rows, dim, device=dev).to(torch.bfloat16)
    g = torch.ones(dim, device=dev); b = torch.zeros(dim, device=dev)
    us = timeit(lambda: ops.layer_norm(x, g, b, 1e-5))
    print(f"ln  ({rows}, {dim}): {us:7.1f} us  {rows * dim * 4 / us / 1e6:6.2f} TB/s")
for n, c, hw in [(8, 320, 64), (8, 640, 32), (8, 960, 64), (8, 640, 64), (8, 1280, 32), (8, 1920, 32)]:
    x = torch.randn(n, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    us = timeit(lambda: ops.group_norm(x, g, b, 32, 1e-5, True))
    print(f"gn  ({n}, {c}, {hw}): {us:7.1f} us  {n * c * hw * hw * 6 / us / 1e6:6.2f} TB/s (3 passes over the tensor)")
