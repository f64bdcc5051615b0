#!/usr/bin/env python
"""Dev probe: the fp32-class VAE 3x3 convs as bf16 x 3 (split planes) and as CRG_PREC_F16MX (fp16 + MX cross terms): device time inside a
captured graph and rel-L2 against an fp32 torch conv on a sub-sample."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
from tools.gt import graph_us
dev = "cuda:0"
torch.manual_seed(0)
tot = {"x3": 0.0, "mx": 0.0}
for (n, ci, hw, co, cnt) in [(4, 128, 512, 128, 7), (4, 256, 256, 256, 5), (4, 256, 512, 128, 1), (4, 512, 128, 512, 6), (4, 512, 256, 256, 1), (4, 512, 64, 512, 9)]:
    x = torch.randn(n, hw, hw, ci, device=dev).permute(0, 3, 1, 2)
    w = torch.randn(co, ci, 3, 3, device=dev) * (9 * ci) ** -0.5
    b = torch.randn(co, device=dev)
    hi, lo = ops.split_bf16(x)
    x16, x8 = ops.split_mx(x)
    y3 = ops.conv2d(hi, w, b, x_lo=lo, gn_stats=True)
    ym = ops.conv2d(x16, w, b, x_mx=x8, gn_stats=True)
    ref = F.conv2d(x[:1, :, :34, :34].float().cpu(), w.cpu(), b.cpu(), padding=1)[:, :, 1:33, 1:33]
    e3 = ((y3[:1, :, 1:33, 1:33].cpu() - ref).norm() / ref.norm()).item()
    em = ((ym[:1, :, 1:33, 1:33].cpu() - ref).norm() / ref.norm()).item()
    t3 = graph_us(lambda: ops.conv2d(hi, w, b, x_lo=lo, gn_stats=True), n=5, reps=3)
    tm = graph_us(lambda: ops.conv2d(x16, w, b, x_mx=x8, gn_stats=True), n=5, reps=3)
    tot["x3"] += cnt * t3
    tot["mx"] += cnt * tm
    fl = 2.0 * n * hw * hw * co * ci * 9
    print(f"{(n, ci, hw, co)} x{cnt}: bf16 x 3 {t3:8.1f} us ({fl / t3 / 1e6:6.1f} TF-eq, rel {e3:.1e})   MX {tm:8.1f} us ({fl / tm / 1e6:6.1f} TF-eq, rel {em:.1e})   {tm / t3:.3f}", flush=True)
print(f"decoder 3x3 convs behind a GroupNorm, weighted: bf16 x 3 {tot['x3'] / 1e3:.2f} ms, MX {tot['mx'] / 1e3:.2f} ms per batch of four")
