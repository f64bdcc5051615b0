#!/usr/bin/env python
"""Dev probe (needs a library built with in-kernel wall_clock64 stamps, tools/ab/libcrg_timing.so): where a lone GEMM block
spends its fixed latency chain.  Stamps: 0 kernel entry, 1 before the first DMA issue, 2 after it, 3 first tile landed (+barrier),
4 K loop done, 5 epilogue stores retired, 6 epilogue stores issued, 7 after block_to_tile."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops, _lib as L
dev = "cuda:0"
lib = C.CDLL(L.LIB_PATH)
for (m, n, k) in [(2048, 1280, 1280), (8192, 640, 640), (32768, 320, 320), (2048, 1280, 64)]:
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(n, device=dev)
    r = torch.randn(m, n, device=dev).to(torch.bfloat16)
    for _ in range(3):
        ops.linear(x, w, b, residual=r)
    torch.cuda.synchronize()
    ops.linear(x, w, b, residual=r)
    torch.cuda.synchronize()
    nb = 256
    buf = (C.c_ulonglong * (8 * nb))()
    lib.crg_debug_read(buf, 8 * nb)
    ts = torch.tensor(list(buf), dtype=torch.float64).reshape(nb, 8)[:, :8]
    t0 = ts[:, 0].min()
    d = (ts - ts[:, :1]) / 100.0   # wall_clock64 ticks at 100 MHz -> us
    print(f"{(m, n, k)}: per-block median us since entry: " + "  ".join(f"s{i}={d[:, i].median().item():.2f}" for i in (7, 2, 1, 6, 3, 4, 5)) +
          f" | block entry spread {((ts[:, 0].max() - t0) / 100.0).item():.2f} us, last stamp {((ts[:, 5].max() - t0) / 100.0).item():.2f} us")
