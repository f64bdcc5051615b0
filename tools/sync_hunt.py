#!/usr/bin/env python
"""Dev helper: where does the host synchronise with the GPU inside one bench batch?  torch's sync debug mode warns at every blocking call;
the host-side duration of a batch (no explicit synchronisation) against its GPU duration shows whether the host runs ahead."""
import os, sys, time, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import dist as D, ops, pipeline as P
from cremage_amd.synth import synth_input
dev = torch.device("cuda", 0)
ldm, _ = P.build_ldm_sharded(0, dev, unet_dtype=ops.HALF, vae_dtype=torch.float32, seed=1234)
ldm.model.enable_hip_graph(True)
b = 4
c = torch.stack([synth_input(f"bench.c{i}", (77, 768), 7) for i in range(b)]).to(dev)
uc = synth_input("bench.uc", (1, 77, 768), 7).expand(b, -1, -1).contiguous().to(dev)
gens = [torch.Generator(device=dev).manual_seed(42 + i) for i in range(b)]
step = lambda: P.txt2img(ldm, c, uc, steps=20, sampler="euler_a", cfg_scale=7.5, height=512, width=512, generators=gens)[0]
step(); step()
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = step()
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"batch {k}: host returned after {1e3 * th:.1f} ms, GPU time {e0.elapsed_time(e1):.1f} ms", flush=True)
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
    torch.cuda.set_sync_debug_mode("default")
    seen = {}
    for x in w:
        key = (str(x.message)[:90], x.filename.split("/")[-1], x.lineno)
        seen[key] = seen.get(key, 0) + 1
    for k, n in seen.items():
        print(n, k)
print("sync warnings:", len(w))
