#!/usr/bin/env python
"""Dev probe: the row-resident LayerNorm GEMMs of the 64x64 level (crg_ln_gemm), device time inside a captured graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
M, K = 32768, 320
x = torch.randn(8, M // 8, K, device=dev).to(torch.bfloat16)
g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
for N, act, tf in [(960, None, 640), (320, None, None), (2560, "geglu", None)]:
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev) if act else None
    f = lambda: ops.ln_linear(x, g, b, 1e-5, w, bias, act=act, transposed_from=tf)
    us = graph_us(f, n=10)
    print(f"ln_linear N={N} {act} vt={tf}: {us:7.1f} us  ({2.0 * M * N * K / us / 1e6:6.1f} TF)", flush=True)
