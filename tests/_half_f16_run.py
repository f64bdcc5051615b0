"""Child process of test_fp16_operand_build (tests/test_hip_models.py): runs with CRG_HALF=f16, i.e. against libcrg_hip_f16.so - the same
kernels compiled for IEEE fp16 operands (crg_common.h) - and prints ONE JSON line: per-op errors against fp32 PyTorch on fp16-rounded
inputs, and the full-size C1 trajectory (20-step Euler, 512x512, CFG 7.5) against the fixture the reference's fp32 CPU path produced.
Not collected by pytest (leading underscore); the environment variable must be set before cremage_amd is imported."""
import json
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from cremage_amd import _lib, ops, pipeline as P  # noqa: E402
from cremage_amd.synth import synth_input  # noqa: E402
from tests.conftest import GOLD, load_golden, rel_l2  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def main():
    assert _lib.HALF_F16 and ops.HALF == torch.float16 and _lib.load().crg_half_kind() == 1
    dev = torch.device("cuda:0")
    H = torch.float16
    q = lambda t: t.to(H).float()
    out = {}
    # ---- per-op: linear (+residual), 3x3 conv (ring kernel shape), flash attention, GroupNorm + SiLU, LayerNorm + GEMM ----
    x, w, b, r = rnd(4096, 640, seed=1), rnd(640, 640, seed=2, scale=640 ** -0.5), rnd(640, seed=3), rnd(4096, 640, seed=4)
    got = ops.linear(x.to(dev).to(H), w.to(dev), b.to(dev), residual=r.to(dev).to(H)).float().cpu()
    out["linear"] = rel_l2(got, F.linear(q(x), q(w), b) + q(r))
    xc, wc = rnd(2, 320, 64, 64, seed=5), rnd(320, 320, 3, 3, seed=6, scale=(9 * 320) ** -0.5)
    got = ops.conv2d(xc.to(dev).to(H).contiguous(memory_format=torch.channels_last), wc.to(dev), None, padding=1, gn_stats=True)
    out["conv3x3"] = rel_l2(got.float().cpu().contiguous(), F.conv2d(q(xc), q(wc), None, padding=1))
    gam, bet = (1 + 0.1 * rnd(320, seed=7)).to(dev), (0.1 * rnd(320, seed=8)).to(dev)
    gn = ops.group_norm(got, gam, bet, 32, 1e-5, silu=True)
    out["groupnorm_pre"] = rel_l2(gn.float().cpu().contiguous(), F.silu(F.group_norm(got.float().cpu(), 32, gam.cpu(), bet.cpu(), 1e-5)))
    qq, kk, vv = rnd(2, 4096, 320, seed=9), rnd(2, 4096, 320, seed=10), rnd(2, 4096, 320, seed=11)
    vt = vv.transpose(1, 2).contiguous()
    got = ops.attention(qq.to(dev).to(H), kk.to(dev).to(H), vt.to(dev).to(H), 8, 4096, 40 ** -0.5).float().cpu()
    hd = lambda t: q(t).view(2, 4096, 8, 40).transpose(1, 2)
    ref = (torch.softmax(hd(qq) @ hd(kk).transpose(-1, -2) * 40 ** -0.5, -1) @ hd(vv)).transpose(1, 2).reshape(2, 4096, 320)
    out["attention_4096_d40"] = rel_l2(got, ref)
    xl, wl = rnd(2, 4096, 320, seed=12) + 0.5, rnd(960, 320, seed=13, scale=320 ** -0.5)
    g2, b2 = 1 + 0.1 * rnd(320, seed=14), 0.1 * rnd(320, seed=15)
    got = ops.ln_linear(xl.to(dev).to(H), g2.to(dev), b2.to(dev), 1e-5, wl.to(dev)).float().cpu()
    out["ln_linear"] = rel_l2(got, F.linear(F.layer_norm(q(xl), (320,), g2, b2, 1e-5).to(H).float(), q(wl)))
    # ---- C1 at full size: fp16 UNet (the reference's own GPU dtype) + fp32-class VAE against the fp32 CPU reference ----
    if os.path.exists(os.path.join(GOLD, "traj_c1_sd15_full.npz")):
        meta, g = load_golden("traj_c1_sd15_full")
        ldm = P.build_synthetic_ldm(meta["unet"], meta["dd"], dev, unet_dtype=H, vae_dtype=torch.float32, seed=meta["seed"])
        B, L, seed = meta["B"], meta["L"], meta["seed"]
        c = synth_input("c1.c", (B, 77, 768), seed).to(dev)
        uc = synth_input("c1.uc", (B, 77, 768), seed).to(dev)
        x0 = synth_input("c1.x0", (B, 4, L, L), seed).to(dev)
        images, xf = P.txt2img(ldm, c, uc, steps=meta["S"], sampler="euler", cfg_scale=meta["cfg"], height=8 * L, width=8 * L, x0=x0)
        ref_pix = ((g["img_f16"].float() + 1) / 2).clamp(0, 1)
        pix = (images.cpu() - ref_pix).abs()
        out["c1_latent_rel_l2"] = rel_l2(xf.float().cpu(), g["x"])
        out["c1_pixel_linf"] = pix.max().item()
        out["c1_pixel_mean_abs"] = pix.mean().item()
    print("HALF_F16_RESULT " + json.dumps(out))


if __name__ == "__main__":
    with torch.no_grad():
        main()
