#!/usr/bin/env python
"""Dev probe: the UNet's 3x3 convs (all levels), device time inside a captured graph, for whatever build (CRG_LIB) the process was started
with (2 = ring kernel, 5 = ping-pong kernel); prints the max abs difference against the fp32 torch conv for one shape as a sanity check."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print({k: os.environ.get(k) for k in ("CRG_LIB",)})
tot = 0.0
SH = [(8, 320, 320, 64, 12), (8, 640, 320, 64, 2), (8, 960, 320, 64, 1), (8, 640, 640, 64, 1),
      (8, 640, 640, 32, 9), (8, 1280, 640, 32, 2), (8, 1920, 640, 32, 1), (8, 960, 640, 32, 1), (8, 320, 640, 32, 1), (8, 1280, 1280, 32, 1),
      (8, 1280, 1280, 16, 9), (8, 2560, 1280, 16, 2), (8, 1920, 1280, 16, 1), (8, 640, 1280, 16, 1),
      (8, 1280, 1280, 8, 11), (8, 2560, 1280, 8, 3)]
for (N, Cin, Cout, hw, cnt) in SH:
    x = torch.randn(N, hw, hw, Cin, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    r = torch.randn(N, hw, hw, Cout, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    us = graph_us(lambda: ops.conv2d(x, w, b, padding=1, residual=r), n=10)
    fl = 2.0 * N * hw * hw * Cout * Cin * 9
    tot += us * cnt
    y = ops.conv2d(x, w, b, padding=1, residual=r).float()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, padding=1) + r.float()
    err = ((y - ref).norm() / ref.norm()).item()
    print(f"N{N} {Cin:5d}->{Cout:5d} @{hw:2d}x{hw:2d}: {us:7.1f} us {fl / us / 1e6:7.1f} TF  rel-L2 {err:.2e} (x{cnt})", flush=True)
print(f"weighted sum per UNet call: {tot:.0f} us")
