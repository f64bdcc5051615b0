#!/usr/bin/env python
"""Dev probe: what one more DEPENDENT kernel costs inside a captured graph (launch + drain of a trivial kernel), and the same for a tiny
LayerNorm / GroupNorm: the floor under every small op of the UNet."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
x = torch.randn(64, device=dev).to(torch.bfloat16)
y = torch.randn(1, 64, 320, device=dev).to(torch.bfloat16)
g, b = torch.ones(320, device=dev), torch.zeros(320, device=dev)
print(f"silu(64 elements), dependent chain : {graph_us(lambda: ops.silu(x), n=50):6.2f} us per node", flush=True)
print(f"layer_norm(64 x 320)               : {graph_us(lambda: ops.layer_norm(y, g, b), n=50):6.2f} us per node", flush=True)
big = torch.randn(8, 4096, 320, device=dev).to(torch.bfloat16)
print(f"layer_norm(32768 x 320)            : {graph_us(lambda: ops.layer_norm(big, g, b), n=20):6.2f} us per node", flush=True)
def chain():
    t = x
    for _ in range(4):
        t = ops.silu(t)
    return t
print(f"silu x4 chained (true dependency)  : {graph_us(chain, n=20) / 4:6.2f} us per node", flush=True)
