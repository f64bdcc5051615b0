#!/usr/bin/env python
"""Dev probe: SDXL's FF2 (4096 x 1280 x 5120 + residual, row statistics for the next LayerNorm) under CRG_SPLIT_MAX, device time in a graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
out = []
for (M, N, K) in [(4096, 1280, 5120), (16384, 640, 2560)]:
    x = torch.randn(4, M // 4, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(4, M // 4, N, device=dev).to(torch.bfloat16)
    for st in (False, True):
        us = graph_us(lambda: ops.linear(x, w, b, residual=r, row_stats=st), n=10)
        out.append(f"{M}x{N}x{K} stats={int(st)} {us:.1f}")
print("SPLIT_MAX", os.environ.get("CRG_SPLIT_MAX"), " | ".join(out), flush=True)
