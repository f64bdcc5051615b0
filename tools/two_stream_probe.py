#!/usr/bin/env python
"""Dev probe: one SD1.5 UNet call on B = 8 as ONE captured stream vs as TWO captured streams of B = 4 (the CFG halves) that the
GPU may run concurrently - do the per-kernel fixed costs (prologue, output drain, tail) of one chain hide under the other?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import _lib as L, pipeline as P
from cremage_amd.synth import synth_input
dev = torch.device("cuda:0")
ldm = P.build_synthetic_ldm(device=dev, seed=1)
m = ldm.model.diffusion_model
x = synth_input("x", (8, 4, 64, 64), 1).to(dev)
ctx = synth_input("c", (8, 77, 768), 1).to(dev)
t = torch.full((8,), 500.0, device=dev)
xa, xb, ca, cb, ta, tb = x[:4].contiguous(), x[4:].contiguous(), ctx[:4].contiguous(), ctx[4:].contiguous(), t[:4].contiguous(), t[4:].contiguous()
for lane in (0, 1, 2):
    L.set_lane(lane)
    h = L.ctx(0)
    L.check(L.load().crg_ctx_reserve(h, 1 << 30), h, "reserve")
L.set_lane(0)


def run_one():
    return m(x, timesteps=t, context=ctx)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run_two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        L.set_lane(1)
        ya = m(xa, timesteps=ta, context=ca)
    with torch.cuda.stream(s2):
        L.set_lane(2)
        yb = m(xb, timesteps=tb, context=cb)
    L.set_lane(0)
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return ya, yb


def timeit(f, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for _ in range(3):
        run_one()
        run_two()
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        y1 = run_one()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        y2 = run_two()
    torch.cuda.synchronize()
    print(f"one stream  B=8      : {timeit(g1.replay):.3f} ms", flush=True)
    print(f"two streams B=4 + B=4: {timeit(g2.replay):.3f} ms", flush=True)
    print(f"one stream  B=8      : {timeit(g1.replay):.3f} ms", flush=True)
    print(f"two streams B=4 + B=4: {timeit(g2.replay):.3f} ms", flush=True)
    g1.replay(); g2.replay(); torch.cuda.synchronize()
    d = (torch.cat([y2[0], y2[1]]) - y1).abs().max().item()
    print("max |one - two| =", d)
