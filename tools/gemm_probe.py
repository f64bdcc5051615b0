#!/usr/bin/env python
"""Dev probe: time one conv / GEMM shape (events over many reps, host queue kept full) - used with CRG_GEMM8=0/1 and
under rocprofv3 --pmc to see what bounds the MFMA kernels."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="conv")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--only", type=int, default=-1, help="run only the last N shapes")
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
shapes = {
    "conv": [(8, 320, 64, 320), (8, 640, 32, 640), (8, 1280, 16, 1280), (8, 640, 64, 320), (8, 1280, 8, 1280), (8, 2560, 16, 1280),
             (8, 1280, 32, 640)],
    "gemm": [(32768, 320, 2880), (32768, 2560, 320), (8192, 640, 640), (8192, 5120, 640), (2048, 1280, 1280), (32768, 320, 320),
             (2048, 10240, 1280), (2048, 1280, 5120), (512, 10240, 1280), (512, 1280, 5120), (8192, 640, 2560),
             (512, 1280, 1280), (512, 2560, 1280), (616, 1280, 768), (616, 640, 768), (2048, 2560, 1280), (8192, 1280, 640)],
    "mid": [(4096, 1280, 1280), (8192, 640, 640), (2048, 1280, 1280), (4096, 1280, 640), (16384, 640, 640), (4096, 640, 1280), (8192, 1280, 640)],
    "res": [(32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280)],
    "ksweep": [(2048, 1280, k) for k in (64, 128, 320, 640, 1280, 2560)] + [(8192, 640, k) for k in (64, 128, 320, 640, 1280, 2560)],
}[a.kind]
for sh in (shapes[-a.only:] if a.only > 0 else shapes):
    if a.kind == "conv":  # noqa
        n, ci, hw, co = sh
        x = torch.randn(n, ci, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 3, 3, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(co, device=dev)
        f = lambda: ops.conv2d(x, w, b)
        fl = 2.0 * n * hw * hw * co * ci * 9
    else:
        m, nn, k = sh
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(nn, k, device=dev) * 0.02).to(torch.bfloat16)
        f = lambda: ops.linear(x, w)
        fl = 2.0 * m * nn * k
        if a.kind == "res":
            bias = torch.zeros(nn, device=dev)
            res = torch.randn(m, nn, device=dev).to(torch.bfloat16)
            variants = [("plain", f), ("bias", lambda: ops.linear(x, w, bias)), ("bias+res", lambda: ops.linear(x, w, bias, residual=res))]
            for tag, g in variants:
                for _ in range(3):
                    g()
                torch.cuda.synchronize()
                torch.zeros(1, device=dev)  # separator kernel for tools/trace_groups.py
                for _ in range(a.reps):
                    g()
                torch.cuda.synchronize()
                torch.zeros(1, device=dev)
                print(f"res {sh} {tag}", flush=True)
            continue
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    print(f"{a.kind} {sh}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF   (XCD_PART={os.environ.get('CRG_XCD_PART')} SPLIT_BELOW={os.environ.get('CRG_SPLIT_BELOW')})", flush=True)
