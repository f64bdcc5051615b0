#!/usr/bin/env python
"""Dev probe: the UNet's Linear / 1x1-conv shapes (B = 8), device time inside a captured graph, for whatever CRG_GEMM_CFG
the process was started with.  Columns: plain, +residual, geglu where the model uses them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print({k: os.environ.get(k) for k in ("CRG_GEMM_CFG", "CRG_SPLIT_BLOCKS")})
tot = 0.0
# (M, N, K, mode, count per UNet call)
S = [(32768, 320, 320, "res", 15), (32768, 320, 320, "plain", 5), (32768, 320, 1280, "res", 5),
     (8192, 640, 640, "res", 15), (8192, 640, 640, "plain", 5), (8192, 1920, 640, "plain", 5), (8192, 5120, 640, "geglu", 5), (8192, 640, 2560, "res", 5),
     (2048, 1280, 1280, "res", 15), (2048, 1280, 1280, "plain", 5), (2048, 3840, 1280, "plain", 5), (2048, 10240, 1280, "geglu", 5), (2048, 1280, 5120, "res", 5),
     (512, 1280, 1280, "res", 3), (512, 3840, 1280, "plain", 1), (512, 10240, 1280, "geglu", 1), (512, 1280, 5120, "res", 1),
     (8192, 640, 320, "plain", 1), (2048, 1280, 640, "plain", 1)]
for (M, N, K, mode, cnt) in S:
    x = torch.randn(8, M // 8, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(8, M // 8, N, device=dev).to(torch.bfloat16) if mode == "res" else None
    f = (lambda: ops.linear(x, w, b, act="geglu")) if mode == "geglu" else (lambda: ops.linear(x, w, b, residual=r))
    us = graph_us(f, n=10)
    tot += us * cnt
    print(f"M{M:6d} N{N:6d} K{K:5d} {mode:6s}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF  (x{cnt})", flush=True)
print(f"weighted sum per UNet call: {tot:.0f} us")
