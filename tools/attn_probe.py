#!/usr/bin/env python
"""Dev probe: flash attention timings (events over many reps) - transposed-V entry (crg_attention) vs row-major-V entry
(crg_attention_v) on the SD1.5 / SDXL self- and cross-attention shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
QS = float(os.environ.get("QSCALE", "1.0"))  # < 1: flat softmax (the running max rarely moves), like the synthetic-weight model
for (B, N, M, heads, d) in [(8, 4096, 4096, 8, 40), (8, 1024, 1024, 8, 80), (8, 256, 256, 8, 160), (8, 4096, 77, 8, 40), (4, 4096, 4096, 10, 64),
                            (4, 1024, 1024, 20, 64)]:
    C = heads * d
    q = (QS * torch.randn(B, N, C, device=dev)).to(torch.bfloat16)
    k = (QS * torch.randn(B, M, C, device=dev)).to(torch.bfloat16)
    v = torch.randn(B, M, C, device=dev).to(torch.bfloat16)
    vt = torch.nn.functional.pad(v.transpose(1, 2), (0, (-M) % 8)).contiguous()
    qkv = torch.cat([q, k, v], -1) if N == M else None
    fl = 4.0 * B * heads * N * M * d
    for tag, f in [("vt ", lambda: ops.attention(q, k, vt, heads, M, d ** -0.5)), ("v  ", lambda: ops.attention_rows_v(q, k, v, heads, d ** -0.5))] + \
                  ([("qkv", lambda: ops.attention_rows_v(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads, d ** -0.5))] if qkv is not None else []):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"attn {tag} B{B} N{N} M{M} h{heads} d{d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF", flush=True)
