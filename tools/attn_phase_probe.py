#!/usr/bin/env python
"""Dev probe (library built with wall_clock64 laps in attn_kernel): time per phase of the flash loop, summed over the key tiles
of one wave (block x, head 0): 0 prefetch issue, 1 QK^T, 2 softmax, 3 PV, 4 commit + barrier."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops, _lib as L
lib = C.CDLL(L.LIB_PATH)
B, N, Cc, H = 8, 4096, 320, 8
q = torch.randn(B, N, Cc, device="cuda").to(torch.bfloat16); k = torch.randn(B, N, Cc, device="cuda").to(torch.bfloat16)
vt = torch.randn(B, Cc, N, device="cuda").to(torch.bfloat16)
import time
t0 = time.time()
while time.time() - t0 < float(os.environ.get("WARM_S", "2.0")):  # DVFS settles under sustained load
    for _ in range(20):
        ops.attention(q, k, vt, H, N, 40 ** -0.5)
    torch.cuda.synchronize()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 32))()
lib.crg_debug_read_attn(buf, 8 * 32)
raw = torch.tensor(list(buf), dtype=torch.float64).reshape(32, 8)
raw = raw[raw[:, :5].sum(1) > 0]  # blocks that exist (256-query blocks: 16 per head)
if raw[:, 6].min() > 0:
    print(f"in-kernel clock {(raw[:, 5] / raw[:, 6]).median().item() * 100:.0f} MHz (s_memtime / s_memrealtime)")
t = raw[:, :5] / 100.0  # us
names = ["prefetch", "QK^T", "softmax", "PV", "commit+barrier"]
tot = t.sum(1).median().item()
print(f"wave total {tot:.1f} us over {N // 64} tiles = {tot / (N // 64) * 1000:.0f} ns per tile")
for i, nme in enumerate(names):
    print(f"  {nme:16s} {t[:, i].median().item():7.1f} us  {100 * t[:, i].median().item() / tot:5.1f} %")
