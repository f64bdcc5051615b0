#!/usr/bin/env python
"""Dev probe: attention time against the number of blocks per CU (batch sweep at N = 4096, 8 heads x 40): how the resident blocks
of a CU share it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
N, heads, d = 4096, 8, 40
C = heads * d
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    k = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    vt = torch.randn(B, C, N, device=dev).to(torch.bfloat16)
    f = lambda: ops.attention(q, k, vt, heads, N, d ** -0.5)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    print(f"B {B:2d}: {B * heads} heads  {us:8.1f} us  {us / B:7.1f} us per batch item  {4.0 * B * heads * N * N * d / us / 1e6:7.1f} TF", flush=True)
