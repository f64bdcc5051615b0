#!/usr/bin/env python
"""Dev probe: attention with few keys (cross-attention against the 77-token context, 8x8 self-attention) on the shapes of the SD1.5 /
SDXL UNets, K | V as column slices of one fused projection output (the layout the model uses).  Prints time per call and the
error against an fp32 reference.  A/B: run once with CRG_ATTN_CTX=0 (general kernels) and once with the default."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
print("CRG_ATTN_CTX =", os.environ.get("CRG_ATTN_CTX", "1"))
for (B, N, M, heads, d) in [(8, 4096, 77, 8, 40), (8, 1024, 77, 8, 80), (8, 256, 77, 8, 160), (8, 64, 77, 8, 160), (8, 64, 64, 8, 160),
                            (8, 4096, 81, 8, 40), (8, 4096, 4, 8, 40), (4, 4096, 77, 10, 64), (4, 1024, 77, 20, 64), (4, 9216, 77, 8, 40)]:
    C = heads * d
    q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, M, 2 * C, device=dev).to(torch.bfloat16)
    k, v = kv[..., :C], kv[..., C:]
    f = lambda: ops.attention_rows_v(q, k, v, heads, d ** -0.5)
    got = f().float()
    qh = q.float().view(B, N, heads, d).transpose(1, 2)
    kh = k.float().reshape(B, M, heads, d).transpose(1, 2)
    vh = v.float().reshape(B, M, heads, d).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).transpose(1, 2).reshape(B, N, C)
    err = ((got - ref).norm() / ref.norm()).item()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"attn B{B} N{N} M{M} h{heads} d{d}: {us:8.1f} us  rel-L2 {err:.2e}", flush=True)
