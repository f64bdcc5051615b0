#!/usr/bin/env python
"""Dev probe: the GEGLU feed-forward projections (FF1) of the three UNet levels, device time in a graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print(os.environ.get("CRG_LIB"))
for (M, N, K) in [(32768, 2560, 320), (8192, 5120, 640), (2048, 10240, 1280)]:
    x = torch.randn(8, M // 8, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    if K == 320:
        us = graph_us(lambda: ops.ln_linear(x, g, be, 1e-5, w, b, act="geglu"), n=10)
    else:
        us = graph_us(lambda: ops.linear(x, w, b, act="geglu"), n=10)
    print(f"M{M} N{N} K{K} geglu: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
