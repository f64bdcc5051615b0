#!/usr/bin/env python
"""Dev probe: the UNet's 1x1 skip convs over a virtual concat and its stride-2 convs (plain implicit-GEMM kernel), device time in a graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print({k: os.environ.get(k) for k in ("CRG_GEMM_D_MAX", "CRG_GEMM_CFG", "CRG_SPLIT_BLOCKS")})
tot = 0.0
for (N, C1, C2, Cout, hw, ks, stride, cnt) in [(8, 320, 320, 320, 64, 1, 1, 2), (8, 640, 320, 320, 64, 1, 1, 1), (8, 640, 640, 640, 32, 1, 1, 1), (8, 1280, 640, 640, 32, 1, 1, 1),
                                               (8, 640, 320, 640, 32, 1, 1, 1), (8, 1280, 1280, 1280, 16, 1, 1, 2), (8, 1280, 640, 1280, 16, 1, 1, 1), (8, 1280, 1280, 1280, 8, 1, 1, 3),
                                               (8, 320, 0, 320, 64, 3, 2, 1), (8, 640, 0, 640, 32, 3, 2, 1), (8, 1280, 0, 1280, 16, 3, 2, 1)]:
    x = torch.randn(N, hw, hw, C1, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    x2 = torch.randn(N, hw, hw, C2, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2) if C2 else None
    w = (torch.randn(Cout, C1 + C2, ks, ks, device=dev) * (ks * ks * (C1 + C2)) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    us = graph_us(lambda: ops.conv2d(x, w, b, stride=stride, padding=ks // 2, x2=x2), n=10)
    fl = 2.0 * N * (hw // stride) ** 2 * Cout * (C1 + C2) * ks * ks
    tot += us * cnt
    print(f"N{N} {C1}+{C2}->{Cout} @{hw}x{hw} k{ks} s{stride}: {us:7.1f} us {fl / us / 1e6:7.1f} TF (x{cnt})", flush=True)
print(f"weighted sum per UNet call: {tot:.0f} us")
