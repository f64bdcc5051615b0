"""Child process of test_conv_schedules (tests/test_hip_ops.py): started with one CRG_RING value (read once per process), runs the
256-pixel-tile 3x3 conv shapes and prints ONE JSON line {case: [rel-L2 vs fp32 torch, sha256 of the output bytes]}.
Not collected by pytest (leading underscore)."""
import hashlib
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
    dev = torch.device("cuda:0")
    q = lambda t: t.to(BF).float()
    nhwc = lambda t: t.to(dev).to(BF).contiguous(memory_format=torch.channels_last)
    out = {}
    # (N, C1, C2, H, W, Cout, upsample): one round without split-K, concat, 2 / 4 / 16 K slices, linear row buffer, upsample, 128-wide
    # tiles, two image rows per tile, row segments
    for (N, C1, C2, H, W, Co, up) in [(8, 320, 0, 64, 64, 320, False), (8, 320, 320, 64, 64, 320, False), (8, 640, 0, 32, 32, 640, False),
                                      (8, 1280, 0, 16, 16, 1280, False), (8, 1280, 0, 8, 8, 1280, False), (8, 640, 0, 32, 32, 640, True),
                                      (8, 192, 64, 64, 64, 128, False), (2, 128, 0, 128, 128, 320, False), (1, 128, 0, 64, 512, 320, False),
                                      (8, 64, 0, 64, 64, 160, False),    # three (chunk, kernel row) groups cut into K slices of two and ONE group: a slice that is all prologue + last group
                                      (4, 64, 0, 64, 64, 320, False)]:   # three groups, no split: one steady group
        C = C1 + C2
        x = rnd(N, C1, H, W, seed=180)
        x2 = rnd(N, C2, H, W, seed=181) if C2 else None
        w, b = rnd(Co, C, 3, 3, seed=182, scale=(9 * C) ** -0.5), rnd(Co, seed=183)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        res, cvec = rnd(N, Co, Ho, Wo, seed=184), rnd(N, Co, seed=185)
        xin = q(x) if x2 is None else torch.cat([q(x), q(x2)], 1)
        if up:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        ref = F.conv2d(xin, q(w), b, padding=1) + cvec[:, :, None, None] + q(res)
        got = ops.conv2d(nhwc(x), w.to(dev), b.to(dev), x2=nhwc(x2) if C2 else None, upsample2x=up, cvec=cvec.to(dev), residual=nhwc(res))
        g = got.float().cpu()
        digest = hashlib.sha256(got.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).contiguous().cpu().view(torch.int16).numpy().tobytes()).hexdigest()
        out[f"{N}x{C1}+{C2}x{H}x{W}->{Co}{'u' if up else ''}"] = [((g - ref).norm() / ref.norm()).item(), digest]
    print("CONV_SCHED_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
