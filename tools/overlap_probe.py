#!/usr/bin/env python
"""Dev probe: is the whole-job throughput of the C2 workload higher when the fp32-class VAE decode of batch k runs on a second HIP
stream (own crg context = own scratch, `_lib.set_lane(1)`) while the 20 UNet calls of batch k + 1 are enqueued on the main stream?
Prints ms per batch for the sequential and the pipelined form over the same number of batches (all work inside the timed region)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import _lib, pipeline as P
from cremage_amd.synth import synth_input
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ldm = P.build_synthetic_ldm(device=dev, seed=1234)
ldm.model.enable_hip_graph(True)
b = 4
c = torch.stack([synth_input(f"bench.c{i}", (77, 768), 7) for i in range(b)]).to(dev)
uc = synth_input("bench.uc", (1, 77, 768), 7).expand(b, -1, -1).contiguous().to(dev)
g = torch.Generator(device=dev).manual_seed(42)
rn = lambda *a: torch.randn((b, 4, 64, 64), generator=g, device=dev)
side = torch.cuda.Stream(device=dev)


def sample():
    _, z = P.txt2img(ldm, c, uc, steps=20, sampler="euler_a", cfg_scale=7.5, x0=rn(), noise_sampler=rn, decode=False)
    return z


def seq(n):
    for _ in range(n):
        img = P.decode_images(ldm, sample())
    return img


def pipe(n):
    img = None
    for _ in range(n):
        z = sample()
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            prev = _lib.set_lane(1)
            img = P.decode_images(ldm, z)
            _lib.set_lane(prev)
            z.record_stream(side)
    torch.cuda.current_stream().wait_stream(side)
    return img


N = int(os.environ.get("NB", "6"))
for name, f in [("sequential", seq), ("pipelined", pipe), ("sequential", seq), ("pipelined", pipe)]:
    f(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = f(N)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    print(f"{name:10s}: {1e3 * dt / N:8.2f} ms per batch of {b}  ({b * N / dt:6.2f} img/s)", flush=True)
