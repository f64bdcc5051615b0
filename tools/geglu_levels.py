#!/usr/bin/env python
"""Dev probe: the GEGLU projection of every transformer level (B = 8), plain and with the LayerNorm-epilogue correction, and the row-resident
kernel at K = 320 - device time inside a captured graph.  Run it alternately with CRG_LIB=<other build> for an A/B in one gpurun call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
ops.LN_EPI_320 = 2
out = []
for (T, C) in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
    x0 = torch.randn(8, T, C, device=dev).to(torch.bfloat16)
    w0 = (torch.randn(C, C, device=dev) * C ** -0.5).to(torch.bfloat16)
    r0 = torch.randn(8, T, C, device=dev).to(torch.bfloat16)
    x = ops.linear(x0, w0, None, residual=r0, row_stats=True)
    ln = torch.nn.LayerNorm(C).to(dev)
    w = (torch.randn(8 * C, C, device=dev) * C ** -0.5).to(torch.bfloat16)
    b = torch.randn(8 * C, device=dev)
    xn = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
    t_g = graph_us(lambda: ops.linear(xn, w, b, act="geglu"), n=10)
    t_e = graph_us(lambda: ops.linear(x, w, b, act="geglu", ln=(ln.weight, ln.bias, ln.eps)), n=10)
    s = f"M={8 * T} K={C}: plain {t_g:.1f} ln-epilogue {t_e:.1f}"
    if C == 320:
        s += f" row-resident {graph_us(lambda: ops.ln_linear(x, ln.weight, ln.bias, ln.eps, w, b, act='geglu'), n=10):.1f}"
    out.append(s)
print(os.environ.get("CRG_LIB", "tree"), " | ".join(out), flush=True)
