import sys, torch
sys.path.insert(0, '/root/repo')
from tests.conftest import load_golden
from cremage_amd.synth import synth_fill_, synth_input
from cremage_amd.ldm_hip.vae import AutoencoderKL
from cremage_amd import ops
meta, g = load_golden("vae_sd15_full_decode")
for dt in (torch.bfloat16, torch.float32):
    m = synth_fill_(AutoencoderKL(meta["dd"], None, 4), meta["seed"], prefix=meta["prefix"]).to(dt).to("cuda:0").eval()
    z = synth_input("vae_full.z", (1, 4, 64, 64), meta["seed"]).to("cuda:0")
    with torch.no_grad():
        dec = m.decode(z / meta["scale_factor"]).cpu()
    sub = (dec[:, :, ::8, ::8] - g["dec_sub"]).abs()
    pix = ((dec+1)/2).clamp(0,1)[:, :, ::8, ::8] - ((g["dec_sub"]+1)/2).clamp(0,1)
    print(dt, "dec max err", sub.max().item(), "pixel Linf", pix.abs().max().item(), "pixel mean abs", pix.abs().mean().item(), "ref range", g["dec_sub"].min().item(), g["dec_sub"].max().item())
