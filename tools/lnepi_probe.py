#!/usr/bin/env python
"""Dev probe: LayerNorm + consumer GEMM at every transformer level - stand-alone layernorm + linear vs the epilogue-correction route
(and the row-resident kernel at K = 320), and what the row statistics cost their producer.  Device time inside a captured graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
ops.LN_EPI_320 = 2
B = int(os.environ.get("PROBE_B", "8"))
tot = {"sep": 0.0, "epi": 0.0}
for (T, C, cnt) in [(4096, 320, 5), (1024, 640, 5), (256, 1280, 5), (64, 1280, 1)]:
    M = B * T
    x0 = torch.randn(B, T, C, device=dev).to(torch.bfloat16)
    w0 = (torch.randn(C, C, device=dev) * C ** -0.5).to(torch.bfloat16)
    b0 = torch.randn(C, device=dev)
    r0 = torch.randn(B, T, C, device=dev).to(torch.bfloat16)
    p_plain = graph_us(lambda: ops.linear(x0, w0, b0, residual=r0), n=10)
    p_stats = graph_us(lambda: ops.linear(x0, w0, b0, residual=r0, row_stats=True), n=10)
    x = ops.linear(x0, w0, b0, residual=r0, row_stats=True)
    ln = torch.nn.LayerNorm(C).to(dev)
    print(f"level M={M} C={C}: producer to_out+res {p_plain:.1f} -> {p_stats:.1f} us with row statistics", flush=True)
    tot["sep"] += 3 * cnt * p_plain
    tot["epi"] += 3 * cnt * p_stats
    for name, N, act, vt in [("qkv", 3 * C, None, True), ("to_q", C, None, False), ("geglu", 8 * C, "geglu", False)]:
        w = (torch.randn(N, C, device=dev) * C ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev) if act else None
        n0 = 2 * C if (vt and ops.linear_transposed_ok(x, w, 2 * C)) else None
        t_ln = graph_us(lambda: ops.layer_norm(x, ln.weight, ln.bias, ln.eps), n=10)
        xn = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
        t_g = graph_us(lambda: ops.linear(xn, w, b, act=act, transposed_from=n0), n=10)
        t_e = graph_us(lambda: ops.linear(x, w, b, act=act, transposed_from=n0, ln=(ln.weight, ln.bias, ln.eps)), n=10)
        extra = ""
        if C == 320 and ops.ln_linear_ok(x, w, act, n0):
            t_r = graph_us(lambda: ops.ln_linear(x, ln.weight, ln.bias, ln.eps, w, b, act=act, transposed_from=n0), n=10)
            extra = f"   row-resident {t_r:.1f}"
            tot["sep"] += cnt * t_r
        else:
            tot["sep"] += cnt * (t_ln + t_g)
        tot["epi"] += cnt * t_e
        print(f"   {name:6s} N={N:6d}: layernorm {t_ln:.1f} + gemm {t_g:.1f} = {t_ln + t_g:.1f} us   epilogue route {t_e:.1f} us{extra}", flush=True)
print(f"per UNet call: round-3 routes {tot['sep']:.0f} us, epilogue route everywhere {tot['epi']:.0f} us")
