#!/usr/bin/env python
"""Dev probe: LayerNorm + Linear as two / three launches vs the fused row-resident kernel (crg_ln_gemm), SD1.5 64x64-level shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
M, K = 32768, 320
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)


def timeit(f, reps=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for N, act in [(960, None), (320, None), (2560, "geglu")]:
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev) if act else None
    t2 = timeit(lambda: ops.linear(ops.layer_norm(x, g, b, 1e-5), w, bias, act=act))
    t1 = timeit(lambda: ops.ln_linear(x, g, b, 1e-5, w, bias, act=act))
    fl = 2.0 * M * N * K
    print(f"ln+linear N={N} {act}: separate {t2:7.1f} us   fused {t1:7.1f} us  ({fl / t1 / 1e6:6.1f} TF)", flush=True)
