#!/usr/bin/env python
"""Dev probe: GroupNorm + SiLU on conv outputs that carry their producer's statistics (crg_groupnorm_pre: tile fold, or finalise + apply),
device time of the GroupNorm launch(es) alone inside a captured graph.  Alternate with CRG_LIB=<other build> for an A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
out = []
for (N, cin, C, hw) in [(8, 320, 320, 64), (8, 320, 640, 64), (8, 640, 640, 32), (8, 640, 1280, 32), (8, 320, 320, 32)]:
    x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn(C, cin, 3, 3, device=dev) * (9 * cin) ** -0.5).to(torch.bfloat16)
    y = ops.conv2d(x, w, None, padding=1, gn_stats=True)
    g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    us = graph_us(lambda: ops.group_norm(y, g, be, 32, 1e-5, silu=True), n=10)
    out.append(f"{N}x{C}x{hw}^2 rows{ops._gn_rows_of(y)} {us:.1f}")
print(os.environ.get("CRG_LIB", "tree"), " | ".join(out), flush=True)
