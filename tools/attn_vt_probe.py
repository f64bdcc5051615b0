#!/usr/bin/env python
"""Dev probe: self-attention with V transposed ([B, C, keys]: LDS-DMA / pipelined kernels) vs V row-major as a slice of the fused
Q|K|V projection output (register-staged kernel with transposing LDS reads), device time inside a captured graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
for (B, N, heads, d) in [(8, 4096, 8, 40), (8, 1024, 8, 80), (8, 256, 8, 160), (4, 2304, 8, 80), (4, 4096, 10, 64), (4, 1024, 20, 64)]:
    C = heads * d
    qkv = torch.randn(B, N, 3 * C, device=dev).to(torch.bfloat16)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    vt = v.transpose(1, 2).contiguous()
    t_v = graph_us(lambda: ops.attention_rows_v(q, k, v, heads, d ** -0.5), n=10)
    t_vt = graph_us(lambda: ops.attention(q, k, vt, heads, N, d ** -0.5), n=10)
    fl = 4.0 * B * heads * N * N * d
    print(f"B{B} N{N} h{heads} d{d}: row-major V {t_v:7.1f} us ({fl / t_v / 1e6:6.1f} TF)   transposed V {t_vt:7.1f} us ({fl / t_vt / 1e6:6.1f} TF)", flush=True)
