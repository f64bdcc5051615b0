#!/usr/bin/env python
"""Dev probe (library built with -DCRG_PP_STAMPS, tools/build_one_variant.sh stamps conv_pp -DCRG_PP_STAMPS; run with
CRG_LIB=tools/ab/libcrg_stamps.so): where a conv3_pp_kernel launch spends its time per block (wave 0): entry -> first DMA issued ->
first operands landed -> K loop done -> epilogue stores issued.  100 MHz wall clock (10 ns resolution)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops, _lib as L
lib = C.CDLL(L.LIB_PATH)
for (nb, ci, hw, co) in [(8, 320, 64, 320), (8, 640, 64, 320), (8, 640, 32, 640), (8, 1280, 16, 1280), (8, 1280, 8, 1280)]:
    x = torch.randn(nb, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.zeros(co, device="cuda")
    f = lambda: ops.conv2d(x, w, b)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 8))()
    lib.crg_debug_read_pp(buf, 1024 * 8)
    t = torch.tensor(list(buf), dtype=torch.float64).reshape(1024, 8) / 100.0  # us
    t = t[t[:, 4] > 0]
    t0 = t[:, 0].min()
    ph = [(t[:, i + 1] - t[:, i]) for i in range(4)]
    print(f"conv {nb}x{ci}x{hw}x{hw} -> {co}: {len(t)} blocks, launch (events) {e0.elapsed_time(e1) * 1e3 / 20:.1f} us; "
          f"entry spread {t[:, 0].max() - t0:.1f} us, first / last exit {t[:, 4].min() - t0:.1f} / {t[:, 4].max() - t0:.1f} us; per block (median / max): "
          f"setup {ph[0].median():.2f} / {ph[0].max():.2f}  first DMA wait {ph[1].median():.2f} / {ph[1].max():.2f}  "
          f"K loop {ph[2].median():.1f} / {ph[2].max():.1f}  epilogue {ph[3].median():.2f} / {ph[3].max():.2f} us", flush=True)
