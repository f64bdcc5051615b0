#!/usr/bin/env python
"""Dev helper for PMC passes: a few launches of one UNet 3x3 conv shape (argv: Cin Cout hw; default 640 640 64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
ci, co, hw = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (640, 640, 64)))
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(8, hw, hw, ci, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
w = (torch.randn(co, ci, 3, 3, device=dev) * (9 * ci) ** -0.5).to(torch.bfloat16)
b = torch.randn(co, device=dev)
for _ in range(6):
    y = ops.conv2d(x, w, b, padding=1)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
