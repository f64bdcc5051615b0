#!/usr/bin/env python
"""Dev probe: fused Q|K|V projection + self-attention, V row-major (round 2) vs V^T out of the GEMM epilogue (round 3), device time in a graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
for (B, T, C, heads) in [(8, 1024, 640, 8), (4, 2304, 640, 8), (4, 4096, 640, 10), (4, 1024, 1280, 20)]:
    d = C // heads
    x = torch.randn(B, T, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(torch.bfloat16)
    def old():
        qkv = ops.linear(x, w)
        return ops.attention_rows_v(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5)
    def new():
        qk, vt = ops.linear(x, w, transposed_from=2 * C)
        return ops.attention(qk[..., :C], qk[..., C:], vt, heads, T, d ** -0.5)
    a, b = old().float(), new().float()
    err = ((a - b).norm() / a.norm()).item()
    t_old, t_new = graph_us(old, n=10), graph_us(new, n=10)
    g_old, g_new = graph_us(lambda: ops.linear(x, w), n=10), graph_us(lambda: ops.linear(x, w, transposed_from=2 * C), n=10)
    print(f"B{B} T{T} C{C} h{heads}: qkv+attn {t_old:7.1f} -> {t_new:7.1f} us   (projection alone {g_old:6.1f} -> {g_new:6.1f})  rel diff {err:.1e}", flush=True)
