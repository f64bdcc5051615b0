#!/usr/bin/env python
"""Dev probe: the UNet's 3x3 convs at the 8x8 / 16x16 / 32x32 levels (split-K territory), device time inside a captured graph, for
whatever CRG_ROWHALO / CRG_SPLIT_BLOCKS / CRG_SPLIT_MAX the process was started with."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print({k: os.environ.get(k) for k in ("CRG_ROWHALO", "CRG_SPLIT_BLOCKS", "CRG_SPLIT_MAX")})
tot = 0.0
for (N, Cin, Cout, hw, cnt) in [(8, 1280, 1280, 8, 11), (8, 2560, 1280, 8, 3), (8, 1280, 1280, 16, 6), (8, 2560, 1280, 16, 2), (8, 1920, 1280, 16, 1),
                                (8, 640, 1280, 16, 1), (8, 640, 640, 32, 6), (8, 1280, 640, 32, 1), (8, 1920, 640, 32, 1), (8, 960, 640, 32, 1), (8, 320, 640, 32, 1)]:
    x = torch.randn(N, hw, hw, Cin, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    r = torch.randn(N, hw, hw, Cout, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    us = graph_us(lambda: ops.conv2d(x, w, b, padding=1, residual=r), n=10)
    fl = 2.0 * N * hw * hw * Cout * Cin * 9
    tot += us * cnt
    print(f"N{N} {Cin:5d}->{Cout:5d} @{hw:2d}x{hw:2d}: {us:7.1f} us {fl / us / 1e6:7.1f} TF  (x{cnt})", flush=True)
print(f"weighted sum per UNet call: {tot:.0f} us")
