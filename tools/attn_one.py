#!/usr/bin/env python
"""Dev: a few launches of the N = 4096, 8 x 40 self-attention (batch from argv) for rocprofv3 --pmc passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, heads, d = 4096, 8, 40
C = heads * d
torch.manual_seed(0)
q = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
k = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, C, N, device="cuda").to(torch.bfloat16)
for _ in range(5):
    ops.attention(q, k, vt, heads, N, d ** -0.5)
torch.cuda.synchronize()
