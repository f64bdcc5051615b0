#!/usr/bin/env python
"""Dev probe: 3x3 conv (statistics side channel) -> GroupNorm + SiLU at the 64x64 level, device time of the pair inside a captured graph;
run with CRG_GN_TILE=0 / 1 (32-row partials + finalise launch / tile partials folded by the normalising launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
out = []
for (N, cin, cout, hw) in [(8, 320, 320, 64), (8, 640, 320, 64), (4, 640, 640, 64)]:
    x = torch.randn(N, hw, hw, cin, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(cout, device=dev)
    g, be = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    t_c = graph_us(lambda: ops.conv2d(x, w, b, padding=1, gn_stats=True), n=10)
    t_p = graph_us(lambda: ops.group_norm(ops.conv2d(x, w, b, padding=1, gn_stats=True), g, be, 32, 1e-5, silu=True), n=10)
    out.append(f"{N}x{cin}->{cout}@{hw}: conv {t_c:.1f} conv+gn {t_p:.1f}")
print("GN_TILE", os.environ.get("CRG_GN_TILE", "1"), " | ".join(out), flush=True)
