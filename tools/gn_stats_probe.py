#!/usr/bin/env python
"""Dev probe: conv -> GroupNorm(+SiLU) pairs at the UNet's 64x64 / 32x32 production shapes, with the producer-side statistics
(default) and without (CRG_GN_STATS=0 in a second process); device time inside a captured graph (tools/gt.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
print("CRG_GN_STATS =", os.environ.get("CRG_GN_STATS", "1"))
for (N, Cin, Cout, hw) in [(8, 320, 320, 64), (8, 640, 640, 32), (8, 1280, 640, 32), (8, 640, 320, 64)]:
    x = torch.randn(N, hw, hw, Cin, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    g, be = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    r = torch.randn(N, hw, hw, Cout, device=dev).to(torch.bfloat16).permute(0, 3, 1, 2)
    conv = lambda: ops.conv2d(x, w, b, padding=1, residual=r, gn_stats=True)
    y = conv()
    if os.environ.get("PROBE_EAGER"):  # under rocprofv3 --kernel-trace: plain launches, read the per-kernel device times from the trace
        for _ in range(10):
            ops.group_norm(conv(), g, be, 32, 1e-5, silu=True)
        torch.cuda.synchronize()
        continue
    t_conv = graph_us(conv)
    t_pair = graph_us(lambda: ops.group_norm(conv(), g, be, 32, 1e-5, silu=True))
    t_gn = graph_us(lambda: ops.group_norm(y, g, be, 32, 1e-5, silu=True))
    print(f"N{N} {Cin}->{Cout} @{hw}x{hw}: conv {t_conv:7.1f} us   conv+gn {t_pair:7.1f} us   gn alone {t_gn:7.1f} us   stats={'yes' if getattr(y, '_crg_gn', None) is not None else 'no'}", flush=True)
