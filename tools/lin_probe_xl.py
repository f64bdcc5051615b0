#!/usr/bin/env python
"""Dev probe: SDXL's Linear shapes (B = 4 with CFG, 1024x1024: 4096-token level C = 640, 1024-token level C = 1280), device time in a graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print({k: os.environ.get(k) for k in ("CRG_GEMM_RING", "CRG_GEMM_RING_MIN")})
tot = 0.0
S = [(16384, 640, 640, "res", 6), (16384, 1920, 640, "plain", 2), (16384, 640, 640, "plain", 4), (16384, 5120, 640, "geglu", 2), (16384, 640, 2560, "res", 2),
     (4096, 1280, 1280, "res", 20), (4096, 3840, 1280, "plain", 10), (4096, 1280, 1280, "plain", 12), (4096, 10240, 1280, "geglu", 10), (4096, 1280, 5120, "res", 10)]
for (M, N, K, mode, cnt) in S:
    x = torch.randn(4, M // 4, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(4, M // 4, N, device=dev).to(torch.bfloat16) if mode == "res" else None
    f = (lambda: ops.linear(x, w, b, act="geglu")) if mode == "geglu" else (lambda: ops.linear(x, w, b, residual=r))
    us = graph_us(f, n=10)
    tot += us * cnt
    print(f"M{M:6d} N{N:6d} K{K:5d} {mode:6s}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF  (x{cnt})", flush=True)
print(f"weighted sum: {tot:.0f} us")
