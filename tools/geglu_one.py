#!/usr/bin/env python
"""Dev helper for PMC passes / timing: the GEGLU projection GEMM of one level (argv: M N K; default 8192 5120 640) - a few launches for a
counter pass, or with `time` as 4th argument the device time inside a captured graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (8192, 5120, 640)))
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(8, M // 8, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev)
if len(sys.argv) > 4 and sys.argv[4] == "time":
    from tools.gt import graph_us
    print(f"M{M} N{N} K{K} geglu SUPER={os.environ.get('CRG_GEMM_RING_SUPER', '1')}: {graph_us(lambda: ops.linear(x, w, b, act='geglu'), n=10):.1f} us")
else:
    for _ in range(6):
        y = ops.linear(x, w, b, act="geglu")
    torch.cuda.synchronize()
    print("ok", float(y.float().abs().mean()))
