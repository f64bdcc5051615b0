#!/usr/bin/env python
"""Dev probe: the fp32-class (split-bf16 x3) VAE conv shapes, run alone for rocprofv3 --pmc / kernel-trace analysis."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for (n, ci, hw, co) in [(4, 128, 512, 128), (4, 256, 256, 256), (4, 512, 128, 512), (4, 512, 64, 512)]:
    x = torch.randn(n, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.02
    b = torch.zeros(co, device=dev)
    hi, lo = ops.split_bf16(x)
    for f, tag in ((lambda: ops.conv2d(hi, w, b, x_lo=lo), "planes (LDS-DMA)"),):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"x3 conv {(n, ci, hw, co)} {tag}: {us:9.1f} us  {2.0 * n * hw * hw * co * ci * 9 / us / 1e6:7.1f} TF-equivalent", flush=True)
    for _ in range(2):
        ops.conv2d(x, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"x3 conv {(n, ci, hw, co)}: {us:9.1f} us  {2.0 * n * hw * hw * co * ci * 9 / us / 1e6:7.1f} TF-equivalent", flush=True)
