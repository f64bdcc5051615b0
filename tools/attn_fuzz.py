#!/usr/bin/env python
"""Dev check (GPU): random attention shapes through crg_attention against an fp32 torch reference - exercises the kernel selection
(pipelined / LDS-DMA / register-staged), odd grids (no XCD remap), query tails, key tails, strided q / k slices."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
random.seed(int(os.environ.get("SEED", "0")))
torch.manual_seed(0)
dev = "cuda:0"
worst = 0.0
for it in range(int(os.environ.get("ITERS", "60"))):
    d = random.choice([40, 40, 40, 80, 64, 160, 48, 32])
    H = random.choice([1, 2, 3, 5, 8, 10])
    B = random.choice([1, 2, 3])
    Nq = random.choice([1, 31, 64, 100, 128, 255, 256, 257, 300, 512, 777, 1024])
    Nk = random.choice([4, 64, 77, 81, 128, 154, 192, 256, 320, 384, 512, 640, 1024, 1100])
    C = H * d
    pad_q, pad_k = random.choice([0, 8, 24]), random.choice([0, 8, 16])
    wq = torch.randn(B, Nq, C + 2 * pad_q, device=dev)
    wk = torch.randn(B, Nk, C + 2 * pad_k, device=dev)
    q = wq.to(torch.bfloat16)[..., pad_q:pad_q + C]
    k = wk.to(torch.bfloat16)[..., pad_k:pad_k + C]
    v = torch.randn(B, Nk, C, device=dev).to(torch.bfloat16)
    ld = (Nk + 7) // 8 * 8
    vt = torch.full((B, C, ld), float("nan"), device=dev, dtype=torch.bfloat16)
    vt[:, :, :Nk] = v.transpose(1, 2)
    got = ops.attention(q, k, vt, H, Nk, d ** -0.5).float()
    sp = lambda t: t.float().reshape(B, t.shape[1], H, d).permute(0, 2, 1, 3)
    s = torch.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * d ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(B, Nq, C)
    rel = ((got - ref).norm() / ref.norm()).item()
    worst = max(worst, rel)
    ok = torch.isfinite(got).all().item() and rel < 1e-2
    print(f"{'ok ' if ok else 'BAD'} B{B} H{H} d{d} Nq{Nq} Nk{Nk} pads {pad_q}/{pad_k}: rel {rel:.2e}", flush=True)
    assert ok
print(f"all ok, worst rel {worst:.2e}")
