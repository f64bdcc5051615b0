"""Child process of test_linear_ring_gemm (tests/test_hip_ops.py): started with CRG_GEMM_RING=2 CRG_GEMM_RING_MIN=50 so that every
eligible bf16 GEMM below runs on gemm_ring_kernel; prints ONE JSON line {case: [rel-L2, max-abs, max|ref|, reproducible]}.
Not collected by pytest (leading underscore)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cremage_amd import ops  # noqa: E402

BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def main():
    assert os.environ.get("CRG_GEMM_RING") == "2"
    dev = torch.device("cuda:0")
    q = lambda t: t.to(BF).float()
    out = {}
    for (M, N, K, mode) in [(32768, 320, 320, "res"), (8192, 1920, 640, "plain"), (8192, 5120, 640, "geglu"), (8200, 640, 2560, "res"),
                            (16384, 1000, 256, "bias"), (65536, 160, 128, "res"), (2048, 10240, 1280, "geglu"), (50000, 320, 192, "plain")]:
        x, w, b = rnd(M, K, seed=301), rnd(N, K, seed=302, scale=K ** -0.5), rnd(N, seed=303)
        xd, wd, bd = x.to(dev).to(BF), w.to(dev), b.to(dev)
        if mode == "geglu":
            f = lambda: ops.linear(xd, wd, bd, act="geglu")
            h = F.linear(q(x), q(w), b)
            ref = h[:, :N // 2] * F.gelu(h[:, N // 2:])
        elif mode == "res":
            r = rnd(M, N, seed=304)
            rd = r.to(dev).to(BF)
            f = lambda: ops.linear(xd, wd, bd, residual=rd)
            ref = F.linear(q(x), q(w), b) + q(r)
        else:
            bb = bd if mode == "bias" else None
            f = lambda: ops.linear(xd, wd, bb)
            ref = F.linear(q(x), q(w), b if mode == "bias" else None)
        got = f()
        same = bool(torch.equal(got, f()))
        g = got.float().cpu()
        out[f"{M}x{N}x{K}_{mode}"] = [((g - ref).norm() / ref.norm()).item(), (g - ref).abs().max().item(), ref.abs().max().item(), same]
    print("RING_GEMM_RESULT " + json.dumps(out))


if __name__ == "__main__":
    with torch.no_grad():
        main()
