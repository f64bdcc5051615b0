#!/usr/bin/env python
"""Per-shape timing of every conv / GEMM / attention / norm launch of one SD1.5 UNet call (B=8, 64x64 latent)
and one VAE decode: records the call shapes by wrapping cremage_amd.ops, then replays every unique shape
in isolation (events on the launch stream) and prints a table sorted by total time.  Dev tool (GPU box)."""
import argparse
import collections
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cremage_amd import ops, pipeline as P  # noqa: E402
from cremage_amd.synth import synth_input  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="unet", choices=["unet", "vae"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--graph", action="store_true", help="device time inside a captured hipGraph (tools/gt.py) instead of eager launches "
                                                         "bracketed by events (which are host-bound below ~15 us)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ldm = P.build_synthetic_ldm(device=dev, seed=1)
    calls = []
    orig = {n: getattr(ops, n) for n in ["conv2d", "linear", "linear_transposed", "attention", "attention_rows_v", "group_norm", "layer_norm", "ln_linear"]}

    def wrap(name):
        f = orig[name]

        def g(*args, **kw):
            calls.append((name, args, kw))
            return f(*args, **kw)
        return g

    for n in orig:
        setattr(ops, n, wrap(n))
    with torch.no_grad():
        if a.what == "unet":
            x = synth_input("x", (a.batch, 4, 64, 64), 1).to(dev)
            ctx = synth_input("c", (a.batch, 77, 768), 1).to(dev)
            t = torch.full((a.batch,), 500.0, device=dev)
            ldm.model.diffusion_model(x, timesteps=t, context=ctx)
        else:
            z = synth_input("z", (a.batch, 4, 64, 64), 1).to(dev)
            ldm.decode_first_stage(z)
    for n in orig:
        setattr(ops, n, orig[n])

    def sig(name, args, kw):
        def sh(v):
            if torch.is_tensor(v):
                return (tuple(v.shape), str(v.dtype).replace("torch.", ""))
            if isinstance(v, (tuple, list)):
                return tuple(sh(u) for u in v)
            return v
        return (name, tuple(sh(v) for v in args), tuple(sorted((k, sh(v)) for k, v in kw.items())))

    groups = collections.OrderedDict()
    for name, args, kw in calls:
        s = sig(name, args, kw)
        if s not in groups:
            groups[s] = [0, name, args, kw]
        groups[s][0] += 1

    def flops(name, args, kw, out):
        if name == "conv2d":
            x, w = args[0], args[1]
            return 2.0 * out.shape[0] * out.shape[2] * out.shape[3] * w.shape[0] * w[0].numel()
        if name in ("linear", "linear_transposed"):
            x, w = args[0], args[1]
            return 2.0 * (x.numel() // x.shape[-1]) * w.shape[0] * w[0].numel()
        if name == "ln_linear":
            x, w = args[0], args[4]
            return 2.0 * (x.numel() // x.shape[-1]) * w.shape[0] * w[0].numel()
        if name == "attention":
            q, k = args[0], args[1]
            return 4.0 * q.shape[0] * q.shape[1] * args[4] * q.shape[2]
        if name == "attention_rows_v":
            q, k = args[0], args[1]
            return 4.0 * q.shape[0] * q.shape[1] * k.shape[1] * q.shape[2]
        return 0.0

    def nbytes(name, args, kw, out):
        outs = out if isinstance(out, tuple) else (out,)
        tot = sum(o.numel() * o.element_size() for o in outs)
        for v in list(args) + list(kw.values()):
            if torch.is_tensor(v):
                tot += v.numel() * v.element_size()
        return tot

    rows = []
    with torch.no_grad():
        for s, (cnt, name, args, kw) in groups.items():
            f = orig[name]
            out = f(*args, **kw)
            torch.cuda.synchronize()
            if a.graph:
                from tools.gt import graph_us
                us = graph_us(lambda: f(*args, **kw), n=a.reps, reps=3)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    f(*args, **kw)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.reps
            fl = flops(name, args, kw, out)
            by = nbytes(name, args, kw, out)
            desc = f"{name} " + " ".join(str(x) for x in s[1] if x is not None) + " " + " ".join(f"{k}={v}" for k, v in s[2] if v is not None and v is not False)
            rows.append(dict(desc=desc[:150], count=cnt, us=us, total_us=us * cnt, tflops=fl / us / 1e6 if fl else 0.0, gbs=by / us / 1e3))
    rows.sort(key=lambda r: -r["total_us"])
    tot = sum(r["total_us"] for r in rows)
    print(f"# {a.what} batch {a.batch}: {len(calls)} launches-level calls, {len(rows)} unique shapes, sum {tot/1e3:.2f} ms")
    for r in rows:
        print(f"{r['total_us']/1e3:8.3f} ms  x{r['count']:<3d} {r['us']:9.1f} us  {r['tflops']:7.1f} TF  {r['gbs']:8.1f} GB/s  {r['desc']}")
    if a.out:
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
