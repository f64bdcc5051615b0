#!/usr/bin/env python
"""Dev probe: the thin convs (conv_in / conv_out of the UNet and the VAE), device time inside a captured graph + error vs torch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
for (N, Cin, Cout, hw, dt) in [(8, 4, 320, 64, torch.bfloat16), (8, 320, 4, 64, torch.bfloat16), (4, 4, 512, 64, torch.float32), (4, 128, 3, 512, torch.float32)]:
    x = torch.randn(N, hw, hw, Cin, device=dev).to(dt).permute(0, 3, 1, 2)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5).to(dt)
    b = torch.randn(Cout, device=dev)
    us = graph_us(lambda: ops.conv2d(x, w, b, padding=1), n=10)
    y = ops.conv2d(x, w, b, padding=1).float()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, padding=1)
    err = ((y - ref).norm() / ref.norm()).item()
    print(f"N{N} {Cin:4d}->{Cout:4d} @{hw}x{hw} {str(dt)[6:]}: {us:7.1f} us  rel-L2 {err:.2e}", flush=True)
