#!/usr/bin/env python
"""Dev probe: SDXL's one-round token GEMMs (M = 4096, K = 1280) under CRG_GEMM_CFG (tile / wave configuration), device time in a graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
out = []
for (M, N, K, mode) in [(4096, 1280, 1280, "res"), (4096, 1280, 1280, "res+stats"), (4096, 1280, 1280, "plain"), (4096, 3840, 1280, "plain"), (16384, 640, 640, "res+stats")]:
    x = torch.randn(4, M // 4, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(4, M // 4, N, device=dev).to(torch.bfloat16) if mode.startswith("res") else None
    us = graph_us(lambda: ops.linear(x, w, b, residual=r, row_stats=mode.endswith("stats")), n=10)
    out.append(f"{M}x{N}x{K} {mode} {us:.1f}")
print("cfg", os.environ.get("CRG_GEMM_CFG"), " | ".join(out), flush=True)
