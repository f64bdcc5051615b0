#!/usr/bin/env python
"""Dev probe (library built with wall_clock64 laps in the K loop of gemm_glds_kernel): where a wave's time goes per k-tile:
0 waiting for its own DMA (counted vmcnt), 1 barrier, 2 issuing the next DMA batch, 3 fragment reads + MFMAs."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops, _lib as L
lib = C.CDLL(L.LIB_PATH)
for kind, shape in [("gemm", (32768, 320, 2880)), ("conv", (8, 320, 64, 320)), ("gemm", (4096, 1280, 1280))]:
    if kind == "gemm":
        m, n, k = shape
        x = torch.randn(m, k, device="cuda").to(torch.bfloat16); w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
        f = lambda: ops.linear(x, w)
        nk = k // 64
    else:
        nb, ci, hw, co = shape
        x = torch.randn(nb, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 3, 3, device="cuda") * 0.02).to(torch.bfloat16); b = torch.zeros(co, device="cuda")
        f = lambda: ops.conv2d(x, w, b)
        nk = ci * 9 // 64
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 4 * 4))()
    lib.crg_debug_read(buf, 256 * 4 * 4)
    t = torch.tensor(list(buf), dtype=torch.float64).reshape(-1, 4) / 100.0
    t = t[t.sum(1) > 0]
    tot = t.sum(1).median().item()
    names = ["own DMA wait", "barrier", "DMA issue", "reads + MFMA"]
    print(f"{kind} {shape}: K loop {tot:.1f} us per wave, {1e3 * tot / nk:.0f} ns per k-tile: " +
          "  ".join(f"{nm} {100 * t[:, i].median().item() / tot:.0f}%" for i, nm in enumerate(names)))
