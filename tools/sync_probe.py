#!/usr/bin/env python
"""Dev probe: list the host<->device synchronisations of one txt2img batch (torch sync debug mode) and time the host side."""
import os, sys, time, warnings, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cremage_amd import pipeline as P
from cremage_amd.synth import synth_input
dev = "cuda:0"
ldm = P.build_synthetic_ldm(device=dev)
ldm.model.enable_hip_graph(True)
b = 4
c = torch.stack([synth_input(f"bench.c{i}", (77, 768), 7) for i in range(b)]).to(dev)
uc = synth_input("bench.uc", (1, 77, 768), 7).expand(b, -1, -1).contiguous().to(dev)
for _ in range(2):
    P.txt2img(ldm, c, uc, steps=20, sampler="euler_a", cfg_scale=7.5)
torch.cuda.synchronize()
sites = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/cremage_amd/" in f.filename or f.filename.endswith("sync_probe.py")]
    sites[(str(message)[:60], tuple(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:]))] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
t0 = time.perf_counter()
P.txt2img(ldm, c, uc, steps=20, sampler="euler_a", cfg_scale=7.5)
t_host = time.perf_counter() - t0
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host returned after {1e3 * t_host:.1f} ms, GPU done after {1e3 * t_all:.1f} ms")
for k, v in sites.most_common(20):
    print(v, k)
