"""ORACLE tooling - generate the golden vectors under tests/golden/ by importing and
running the REFERENCE's own modules (read-only mount /root/reference) on CPU in fp32.

TEST INFRASTRUCTURE ONLY; runs in the build container only (the GPU box has no
/root/reference and only consumes the committed fixtures).  No reference source is
copied: this script only *imports* it, feeds name-keyed synthetic parameters
(cremage_amd.synth) and records inputs/outputs.

Recipe (verified there):
  * sys.path gets /root/reference/modules; bytecode writing is disabled (RO mount);
  * third-party packages that are absent offline and only needed at import time
    (omegaconf.ListConfig, pytorch_lightning.LightningModule, torchvision, torchdiffeq,
    torchsde) are registered as empty `sys.modules` entries - none of them does any
    arithmetic on this path;
  * GPU_DEVICE=cpu selects `CrossAttentionOriginal` (attention.py:877-883);
  * `torch.cuda.is_available` is forced True ONLY around forwards, to skip the
    Apple-MPS fp16 casts (openaimodel.py:85-90,794-795,814-815; autoencoder.py:334-335)
    so that the path really runs in fp32.

Usage:  python oracle/gen_golden.py [--only NAME ...] [--full]
"""
import argparse
import contextlib
import json
import os
import sys
import time
import types

os.environ["GPU_DEVICE"] = "cpu"
os.environ.setdefault("LOGLEVEL", "WARNING")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference/modules"
GOLD = os.path.join(REPO, "tests", "golden")

from cremage_amd.synth import synth_fill_, synth_input  # noqa: E402


def _install_import_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class ListConfig(list):
        pass

    oc = mod("omegaconf", ListConfig=ListConfig, OmegaConf=object)
    oc.listconfig = mod("omegaconf.listconfig", ListConfig=ListConfig)

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

    pl = mod("pytorch_lightning", LightningModule=LightningModule)
    pl.utilities = mod("pytorch_lightning.utilities")
    pl.utilities.distributed = mod("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", make_grid=None)
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")
    mod("torchdiffeq", odeint=None)
    mod("torchsde")


_install_import_stubs()
sys.path.insert(0, REF)

from ldm.modules.diffusionmodules import openaimodel as R_unet  # noqa: E402
from ldm.modules.diffusionmodules import model as R_vae  # noqa: E402
from ldm.modules.diffusionmodules import util as R_util  # noqa: E402
from ldm.modules import attention as R_attn  # noqa: E402
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution  # noqa: E402
from ldm.models.autoencoder import AutoencoderKL  # noqa: E402
from ldm.models.diffusion.ddpm import LatentDiffusion  # noqa: E402
from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from ldm.models.diffusion import k_diffusion_samplers as R_ks  # noqa: E402
from k_diffusion import external as R_kext  # noqa: E402
from k_diffusion import sampling as R_ksamp  # noqa: E402

assert R_attn.BasicTransformerBlock.ATTENTION_MODES["softmax-original"] is R_attn.CrossAttentionOriginal


@contextlib.contextmanager
def fp32_forward():
    """Scoped `torch.cuda.is_available() -> True` so the reference skips its fp16 casts."""
    orig = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        with torch.no_grad():
            yield
    finally:
        torch.cuda.is_available = orig


def save(name, meta, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), meta=json.dumps(meta), **out)
    sz = os.path.getsize(os.path.join(GOLD, name + ".npz"))
    print(f"[golden] {name}: {sz/1024:.1f} KiB  {list(out)}")


SEED = 1234

# ---------------------------------------------------------------------------- leaf ops


def g_groupnorm():
    for tag, cls_eps, C, hw in [("gn32_e5", 1e-5, 64, 8), ("gn32_e5_c320", 1e-5, 320, 4), ("gn_e6", 1e-6, 128, 6)]:
        if cls_eps == 1e-5:
            m = R_util.normalization(C)
        else:
            m = R_attn.Normalize(C)
        synth_fill_(m, SEED, prefix=tag + ".")
        x = synth_input(tag, (2, C, hw, hw), SEED, 1.5) + 0.3
        with fp32_forward():
            y = m(x)
            ys = torch.nn.SiLU()(y)
        save("op_" + tag, dict(C=C, hw=hw, eps=cls_eps, seed=SEED, prefix=tag + "."), y=y, y_silu=ys)


def g_timestep_embedding():
    t = torch.tensor([0.0, 1.0, 17.25, 499.5, 998.9999, 999.0])
    with fp32_forward():
        e320 = R_util.timestep_embedding(t, 320)
        e64 = R_util.timestep_embedding(t, 64)
    save("op_timestep_embedding", dict(), t=t, e320=e320, e64=e64)


def g_resblock():
    for tag, cin, cout in [("res_same", 64, 64), ("res_skip", 64, 128), ("res_320", 320, 320)]:
        hw = 8 if cin < 320 else 4
        m = R_unet.ResBlock(cin, 256, 0, out_channels=cout)
        synth_fill_(m, SEED, prefix=tag + ".")
        x = synth_input(tag + ".x", (2, cin, hw, hw), SEED)
        emb = synth_input(tag + ".emb", (2, 256), SEED)
        with fp32_forward():
            y = m(x, emb)
        save("blk_" + tag, dict(cin=cin, cout=cout, hw=hw, emb=256, seed=SEED, prefix=tag + "."), y=y)


def g_updown():
    m = R_unet.Downsample(64, True, out_channels=64)
    synth_fill_(m, SEED, prefix="down.")
    x = synth_input("down.x", (2, 64, 10, 10), SEED)
    with fp32_forward():
        y = m(x)
    save("blk_downsample", dict(C=64, hw=10, seed=SEED, prefix="down."), y=y)
    m = R_unet.Upsample(64, True, out_channels=64)
    synth_fill_(m, SEED, prefix="up.")
    x = synth_input("up.x", (2, 64, 5, 5), SEED)
    with fp32_forward():
        y = m(x)
    save("blk_upsample", dict(C=64, hw=5, seed=SEED, prefix="up."), y=y)


def g_attention():
    cases = [
        ("ca_d40_m77", 320, 768, 8, 40, 64, 77),
        ("ca_d80_m154", 640, 768, 8, 80, 36, 154),
        ("ca_d160_self", 1280, None, 8, 160, 64, None),
        ("ca_d64_m77", 128, 96, 2, 64, 100, 77),
        ("ca_d40_self", 320, None, 8, 40, 200, None),
    ]
    for tag, qd, cd, heads, dh, n, mctx in cases:
        m = R_attn.CrossAttentionOriginal(qd, cd, heads=heads, dim_head=dh)
        synth_fill_(m, SEED, prefix=tag + ".")
        x = synth_input(tag + ".x", (2, n, qd), SEED)
        ctx = synth_input(tag + ".ctx", (2, mctx, cd), SEED) if mctx else None
        with fp32_forward():
            y = m(x, context=ctx)
        save("op_" + tag, dict(query_dim=qd, context_dim=cd, heads=heads, dim_head=dh, n=n, m=mctx, seed=SEED,
                               prefix=tag + "."), y=y)
    # LoRA rank-4 branch + IP-Adapter FaceID branch (attention.py:616-641,660-683)
    tag = "ca_lora_ipa"
    m = R_attn.CrossAttentionOriginal(128, 96, heads=4, dim_head=32, lora_ranks=[4], lora_weights=[0.7],
                                      ipa_scale=0.6, ipa_num_tokens=4)
    synth_fill_(m, SEED, prefix=tag + ".")
    with torch.no_grad():
        for name, p in m.named_parameters():
            if "_lora_" in name and p.ndim > 0:
                p.copy_(synth_input(tag + "." + name, p.shape, SEED, 0.2))
    x = synth_input(tag + ".x", (2, 50, 128), SEED)
    ctx = synth_input(tag + ".ctx", (2, 81, 96), SEED)
    with fp32_forward():
        y = m(x, context=ctx)
    save("op_" + tag, dict(query_dim=128, context_dim=96, heads=4, dim_head=32, n=50, m=81, lora_ranks=[4],
                           lora_weights=[0.7], ipa_scale=0.6, ipa_num_tokens=4, seed=SEED, prefix=tag + "."), y=y)


def g_transformer():
    tag = "ff"
    m = R_attn.FeedForward(64, glu=True)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 30, 64), SEED)
    with fp32_forward():
        y = m(x)
    save("op_ff_geglu", dict(dim=64, n=30, seed=SEED, prefix=tag + "."), y=y)

    tag = "btb"
    m = R_attn.BasicTransformerBlock(128, 4, 32, context_dim=96, checkpoint=False)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 36, 128), SEED)
    ctx = synth_input(tag + ".ctx", (2, 77, 96), SEED)
    with fp32_forward():
        y = m(x, context=ctx)
    save("blk_basic_transformer", dict(dim=128, heads=4, dim_head=32, context_dim=96, n=36, m=77, seed=SEED,
                                       prefix=tag + "."), y=y)

    tag = "st"
    m = R_attn.SpatialTransformer(128, 4, 32, depth=1, context_dim=96, use_checkpoint=False)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 128, 6, 6), SEED)
    ctx = synth_input(tag + ".ctx", (2, 77, 96), SEED)
    with fp32_forward():
        y = m(x, context=ctx)
    save("blk_spatial_transformer", dict(C=128, heads=4, dim_head=32, context_dim=96, hw=6, m=77, seed=SEED,
                                         prefix=tag + "."), y=y)


# ---------------------------------------------------------------------------- UNet

TINY_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[2, 1],
                 num_res_blocks=1, channel_mult=[1, 2], num_heads=4, use_spatial_transformer=True,
                 transformer_depth=1, context_dim=96, use_checkpoint=False, legacy=False)
SMALL_SD_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
                     num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                     transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False)
SD15_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                 transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False)


def _unet_case(name, cfg, B, L, mctx, tvals):
    m = R_unet.UNetModel(**cfg)
    synth_fill_(m, SEED, prefix="unet.")
    x = synth_input(name + ".x", (B, 4, L, L), SEED)
    ctx = synth_input(name + ".ctx", (B, mctx, cfg["context_dim"]), SEED)
    t = torch.tensor(tvals, dtype=torch.float32)
    t0 = time.time()
    with fp32_forward():
        y = m(x, timesteps=t, context=ctx)
    dt = time.time() - t0
    keys = sorted(k for k, _ in m.named_parameters())
    save(name, dict(cfg=cfg, B=B, L=L, m=mctx, seed=SEED, prefix="unet.", n_params=sum(p.numel() for p in m.parameters()),
                    n_keys=len(keys), key_sample=keys[:: max(1, len(keys) // 40)], ref_cpu_seconds=dt,
                    threads=torch.get_num_threads()), t=t, y=y)
    return m


def g_unet_tiny():
    _unet_case("unet_tiny", TINY_UNET, 2, 16, 77, [10.0, 731.25])


def g_unet_small_sd():
    _unet_case("unet_small_sd", SMALL_SD_UNET, 2, 16, 77, [3.5, 900.0])


def g_unet_sd15_full():
    # Full-size SD1.5 UNet (859.52 M params), B=2 (one image x CFG), L=64 - config 1's unit of work.
    _unet_case("unet_sd15_full", SD15_UNET, 2, 64, 77, [981.5, 981.5])


# ---------------------------------------------------------------------------- VAE

TINY_DD = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2],
               num_res_blocks=1, attn_resolutions=[], dropout=0.0)
SD15_DD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
               num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def g_vae_blocks():
    tag = "vres"
    m = R_vae.ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 64, 8, 8), SEED)
    with fp32_forward():
        y = m(x, None)
    save("blk_vae_resnet", dict(cin=64, cout=128, hw=8, seed=SEED, prefix=tag + "."), y=y)

    tag = "vattn"
    m = R_vae.AttnBlock(64)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 64, 6, 6), SEED)
    with fp32_forward():
        y = m(x)
    save("blk_vae_attn", dict(C=64, hw=6, seed=SEED, prefix=tag + "."), y=y)

    tag = "vdown"
    m = R_vae.Downsample(64, True)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 64, 10, 10), SEED)
    with fp32_forward():
        y = m(x)
    save("blk_vae_downsample", dict(C=64, hw=10, seed=SEED, prefix=tag + "."), y=y)

    tag = "vup"
    m = R_vae.Upsample(64, True)
    synth_fill_(m, SEED, prefix=tag + ".")
    x = synth_input(tag + ".x", (2, 64, 5, 5), SEED)
    with fp32_forward():
        y = m(x)
    save("blk_vae_upsample", dict(C=64, hw=5, seed=SEED, prefix=tag + "."), y=y)


def _make_ae(dd):
    ae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
    synth_fill_(ae, SEED, prefix="vae.")
    return ae.eval()


def g_vae_tiny():
    ae = _make_ae(TINY_DD)
    z = synth_input("vae_tiny.z", (2, 4, 8, 8), SEED)
    img = synth_input("vae_tiny.img", (2, 3, 16, 16), SEED, 0.5).clamp(-1, 1)
    noise = synth_input("vae_tiny.noise", (2, 4, 8, 8), SEED)
    with fp32_forward():
        dec = ae.decode(z)
        # AutoencoderKL.encode (autoencoder.py:324-331) minus its unconditional x.half() at :327
        moments = ae.quant_conv(ae.encoder(img))
        post = DiagonalGaussianDistribution(moments)
        sample = post.mean + post.std * noise
    save("vae_tiny", dict(dd=TINY_DD, seed=SEED, prefix="vae."), dec=dec, moments=moments, sample=sample)


def g_vae_sd15_full():
    ae = _make_ae(SD15_DD)
    z = synth_input("vae_full.z", (1, 4, 64, 64), SEED)
    t0 = time.time()
    with fp32_forward():
        dec = ae.decode(z / 0.18215)
    dt = time.time() - t0
    img = torch.clamp((dec + 1.0) / 2.0, 0.0, 1.0)  # image_generator.py:1013
    save("vae_sd15_full_decode", dict(dd=SD15_DD, seed=SEED, prefix="vae.", ref_cpu_seconds=dt,
                                      threads=torch.get_num_threads(), scale_factor=0.18215),
         dec_f16=dec.half(), dec_sub=dec[:, :, ::8, ::8].contiguous(), pix_stats=torch.tensor(
             [img.mean().item(), img.std().item(), dec.abs().max().item()]))
    x = synth_input("vae_full.img", (1, 3, 256, 256), SEED, 0.5).clamp(-1, 1)
    with fp32_forward():
        moments = ae.quant_conv(ae.encoder(x))
    save("vae_sd15_full_encode", dict(dd=SD15_DD, seed=SEED, prefix="vae.", hw=256), moments=moments)


# ---------------------------------------------------------------------------- schedules & trajectories


def _tiny_ldm():
    ldm = LatentDiffusion(first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                                              "params": dict(ddconfig=TINY_DD, lossconfig={"target": "torch.nn.Identity"},
                                                             embed_dim=4)},
                          cond_stage_config={"target": "torch.nn.Identity"},
                          unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": TINY_UNET},
                          linear_start=0.00085, linear_end=0.012, timesteps=1000, conditioning_key="crossattn",
                          scale_factor=0.18215, use_ema=False, cond_stage_trainable=False, first_stage_key="jpg",
                          cond_stage_key="txt", image_size=16, channels=4)
    synth_fill_(ldm.model.diffusion_model, SEED, prefix="unet.")
    synth_fill_(ldm.first_stage_model, SEED, prefix="vae.")
    return ldm.eval()


def g_schedules():
    ldm = _tiny_ldm()
    acp = ldm.alphas_cumprod
    den = R_kext.CompVisDenoiser(ldm, False)
    sig20 = den.get_sigmas(20)
    sig5 = den.get_sigmas(5)
    probe = torch.tensor([0.03, 0.5, 1.0, 3.3, 14.6146, 20.0])
    s2t = den.sigma_to_t(probe)
    dd = DDIMSampler(ldm)
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        orig = torch.cuda.is_available
        torch.cuda.is_available = lambda: True  # fp32 tables (ddim.py:43-46); model is on cpu so .to(device) is a no-op
        torch.Tensor.cuda_orig = None
        try:
            dd.register_buffer = lambda name, attr: setattr(dd, name, attr)
            dd.make_schedule(ddim_num_steps=20, ddim_eta=0.0, verbose=False)
        finally:
            torch.cuda.is_available = orig
    save("schedules", dict(linear_start=0.00085, linear_end=0.012, timesteps=1000),
         alphas_cumprod=acp, sigmas_table=den.sigmas, get_sigmas_20=sig20, get_sigmas_5=sig5, sigma_probe=probe,
         sigma_to_t=s2t, ddim_timesteps_20=np.asarray(dd.ddim_timesteps), ddim_alphas_20=np.asarray(dd.ddim_alphas),
         ddim_alphas_prev_20=np.asarray(dd.ddim_alphas_prev))


def g_trajectories():
    """5-step Euler / Euler-a through the reference's own sampler stack
    (EulerSampler -> sample_euler -> LDMWrapperForKDiffusion -> CompVisDenoiser ->
    LatentDiffusion.apply_model -> DiffusionWrapper -> UNetModel), tiny UNet, CFG 7.5,
    then decode_first_stage; plus a DDIM img2img (S=20, t_enc=3)."""
    ldm = _tiny_ldm()
    B, L = 2, 16
    c = synth_input("traj.c", (B, 77, 96), SEED)
    uc = synth_input("traj.uc", (B, 77, 96), SEED)
    x0 = synth_input("traj.x0", (B, 4, L, L), SEED)
    noises = [synth_input(f"traj.noise{i}", (B, 4, L, L), SEED) for i in range(5)]

    R_ks.trange = R_ksamp.trange = lambda *a, **k: range(*a)  # silence tqdm

    for cls, nm in [(R_ks.EulerSampler, "euler"), (R_ks.EulerAncestralSampler, "euler_a")]:
        s = cls(ldm)
        it = iter(noises)
        orig_randn_like = torch.randn_like
        if nm == "euler_a":
            torch.randn_like = lambda x, **k: next(it)  # captured noise instead of the global RNG (sampling.py:61-62)
        try:
            with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
                # sample() opens torch.autocast(device_type=GPU_DEVICE) when cuda "is available"
                # (k_diffusion_samplers.py:244-249); on CPU autocast would drop to bf16, so call the
                # two halves of sample() directly: _sample_common_prep then do_sample (:196-253).
                s._sample_common_prep(S=5, batch_size=B, shape=[4, L, L], conditioning=c,
                                      unconditional_guidance_scale=7.5, unconditional_conditioning=uc, x0=x0)
                x, _ = s.do_sample()
                img = ldm.decode_first_stage(x)
        finally:
            torch.randn_like = orig_randn_like
        save("traj_" + nm, dict(B=B, L=L, S=5, cfg=7.5, seed=SEED, unet=TINY_UNET, dd=TINY_DD), sigmas=s.sigmas, x=x, img=img)

    dd = DDIMSampler(ldm)
    dd.register_buffer = lambda name, attr: setattr(dd, name, attr)
    orig = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        dd.make_schedule(ddim_num_steps=20, ddim_eta=0.0, verbose=False)
    finally:
        torch.cuda.is_available = orig
    import ldm.models.diffusion.ddim as R_ddim
    R_ddim.tqdm = lambda it, **k: it
    img_in = synth_input("traj.img", (B, 3, 32, 32), SEED, 0.5).clamp(-1, 1)
    enc_noise = synth_input("traj.encnoise", (B, 4, L, L), SEED)
    fwd_noise = synth_input("traj.fwdnoise", (B, 4, L, L), SEED)
    with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
        moments = ldm.first_stage_model.quant_conv(ldm.first_stage_model.encoder(img_in))
        post = DiagonalGaussianDistribution(moments)
        init_latent = 0.18215 * (post.mean + post.std * enc_noise)  # ddpm.py:575-582
        z_enc = dd.stochastic_encode(init_latent, torch.tensor([3] * B), noise=fwd_noise)
        x = dd.decode(z_enc, c, 3, unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
        img = ldm.decode_first_stage(x)
    save("traj_ddim_img2img", dict(B=B, L=L, S=20, t_enc=3, cfg=7.5, seed=SEED, unet=TINY_UNET, dd=TINY_DD),
         init_latent=init_latent, z_enc=z_enc, x=x, img=img)


def g_hires_latent():
    """Hires-fix, latent upscaler (image_generator.py:958-999): txt2img at the base size, F.interpolate(bilinear) of the final
    latents, k-diffusion stochastic_encode to t_enc = int(strength * S) (k_diffusion_samplers.py:255-296) and a partial Euler
    denoise over the last t_enc + 1 sigmas (img2img_sampling :227-246, `denoising_steps`), through the reference's sampler stack."""
    ldm = _tiny_ldm()
    B, L, S, factor, strength = 2, 8, 6, 2, 0.5
    c = synth_input("hires.c", (B, 77, 96), SEED)
    uc = synth_input("hires.uc", (B, 77, 96), SEED)
    x0 = synth_input("hires.x0", (B, 4, L, L), SEED)
    noise = synth_input("hires.noise", (B, 4, factor * L, factor * L), SEED)
    R_ks.trange = R_ksamp.trange = lambda *a, **k: range(*a)
    t_enc = int(strength * S)
    # the constructor keeps its schedule buffers in fp32 only when cuda "is available" (k_diffusion_samplers.py:97-100; fp16
    # otherwise) and register_buffer then moves them to "cuda" (:113-121): build it under the shim with that move neutralised,
    # so that stochastic_encode uses the fp32 coefficients of the GPU configuration
    orig_rb = R_ks.KDiffusionSamplerBase.register_buffer
    R_ks.KDiffusionSamplerBase.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    try:
        with fp32_forward():
            s = R_ks.EulerSampler(ldm)
    finally:
        R_ks.KDiffusionSamplerBase.register_buffer = orig_rb
    assert s.sqrt_alphas_cumprod.dtype == torch.float32
    with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
        s._sample_common_prep(S=S, batch_size=B, shape=[4, L, L], conditioning=c, unconditional_guidance_scale=7.5,
                              unconditional_conditioning=uc, x0=x0)
        base, _ = s.do_sample()
        up = torch.nn.functional.interpolate(base, scale_factor=factor, mode="bilinear", align_corners=False)
        z_enc = s.stochastic_encode(up, torch.tensor([t_enc] * B), sampling_steps=S, noise=noise)
        s._sample_common_prep(S=S, batch_size=B, shape=[4, factor * L, factor * L], conditioning=c, unconditional_guidance_scale=7.5,
                              unconditional_conditioning=uc, x0=z_enc, denoising_steps=t_enc)
        x, _ = s.do_sample()
        img = ldm.decode_first_stage(x)
    save("traj_hires_latent", dict(B=B, L=L, S=S, factor=factor, strength=strength, t_enc=t_enc, cfg=7.5, seed=SEED, unet=TINY_UNET, dd=TINY_DD),
         base=base, up=up, z_enc=z_enc, sigmas=s.sigmas, x=x, img=img)


def g_alphas_doc():
    """The reference's only numeric known-answer artefact for this path: 1000 float64 alphas_cumprod values
    printed in docs/developers/ddpm_cumprod_alpha_example_values.md (data only, no source)."""
    import re
    vals = {}
    for line in open("/root/reference/docs/developers/ddpm_cumprod_alpha_example_values.md"):
        m = re.match(r"^\[(\d+)\]\s+([0-9.eE+-]+)\s*$", line)
        if m:
            vals[int(m.group(1))] = float(m.group(2))
    assert sorted(vals) == list(range(1000))
    save("alphas_cumprod_doc", dict(linear_start=0.00085, linear_end=0.012, timesteps=1000,
                                    source="docs/developers/ddpm_cumprod_alpha_example_values.md"),
         alphas_cumprod=np.asarray([vals[i] for i in range(1000)], dtype=np.float64))


def g_param_contract():
    """Names and shapes of every parameter of the reference's SD1.5 UNet (plain and with two LoRA ranks + FaceID, the
    way image_generator.py:314-320 constructs it) and of its AutoencoderKL: the load_state_dict contract."""
    import hashlib

    def digest(m):
        items = sorted(f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items())
        return hashlib.sha1("\n".join(items).encode()).hexdigest(), len(items)

    with torch.device("meta"):
        u = R_unet.UNetModel(**SD15_UNET)
        ul = R_unet.UNetModel(**dict(SD15_UNET, lora_ranks=[4, 16], lora_weights=[1.0, 0.5], ipa_scale=0.7, ipa_num_tokens=4))
        ae = AutoencoderKL(ddconfig=SD15_DD, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
    du, nu = digest(u)
    dl, nl = digest(ul)
    da, na = digest(ae)
    save("param_contract", dict(unet_sha1=du, unet_n=nu, unet_lora_sha1=dl, unet_lora_n=nl, vae_sha1=da, vae_n=na,
                                unet_cfg=SD15_UNET, vae_dd=SD15_DD, lora_ranks=[4, 16], lora_weights=[1.0, 0.5], ipa_scale=0.7,
                                ipa_num_tokens=4), dummy=np.zeros(1))


def g_controlnet_hook():
    """ControlNet's hook into the UNet: the reference's ControlledUnetModel (cldm.py:28-70, a UNetModel subclass whose
    forward adds the control residuals after the middle block and to every skip) on the tiny UNet with synthetic
    control tensors."""
    from cldm.cldm import ControlledUnetModel
    m = ControlledUnetModel(**TINY_UNET)
    synth_fill_(m, SEED, prefix="unet.")
    B, L = 2, 16
    x = synth_input("cn.x", (B, 4, L, L), SEED)
    ctx = synth_input("cn.ctx", (B, 77, TINY_UNET["context_dim"]), SEED)
    t = torch.tensor([250.0, 600.5])
    shapes = [(64, 16, 16), (64, 16, 16), (64, 8, 8), (128, 8, 8), (128, 8, 8)]  # 4 skips + middle (popped first)
    control = [synth_input(f"cn.control{i}", (B,) + s, SEED, 0.3) for i, s in enumerate(shapes)]
    # cldm.py:54-55,67-68 cast to fp16 whenever the tensor is not on a 'cuda' device (Mac path, keyed on device.type, not
    # on is_available): neutralise Tensor.half for the duration of the forward so the path stays fp32
    orig_half = torch.Tensor.half
    torch.Tensor.half = lambda self, *a, **k: self
    try:
        with fp32_forward():
            y = m(x, timesteps=t, context=ctx, control=[c.clone() for c in control], only_mid_control=False)
            y_mid = m(x, timesteps=t, context=ctx, control=[c.clone() for c in control], only_mid_control=True)
    finally:
        torch.Tensor.half = orig_half
    save("hook_controlnet", dict(cfg=TINY_UNET, B=B, L=L, seed=SEED, prefix="unet.", shapes=shapes), t=t, y=y, y_mid=y_mid)



def g_controlnet():
    """The reference's ControlNet (cldm.py:73-342) feeding its ControlledUnetModel (:28-70) exactly as
    ControlLDM.apply_model does (:374-393, control_scales all 1.0 and a non-trivial set), tiny and SD-shaped small."""
    from cldm.cldm import ControlledUnetModel, ControlNet
    for name, cfg, B, L, tvals in [("controlnet_tiny", TINY_UNET, 2, 16, [250.0, 600.5]), ("controlnet_small_sd", SMALL_SD_UNET, 2, 16, [40.0, 905.5])]:
        ccfg = {k: v for k, v in cfg.items() if k != "out_channels"}
        cn = ControlNet(hint_channels=3, **ccfg)
        un = ControlledUnetModel(**cfg)
        synth_fill_(cn, SEED, prefix="cn.")
        synth_fill_(un, SEED, prefix="unet.")
        x = synth_input("cnet.x", (B, 4, L, L), SEED)
        hint = synth_input("cnet.hint", (B, 3, 8 * L, 8 * L), SEED, 0.5).clamp(-1, 1) * 0.5 + 0.5
        ctx = synth_input("cnet.ctx", (B, 77, cfg["context_dim"]), SEED)
        t = torch.tensor(tvals)
        n_ctrl = len(cn.input_blocks) + 1
        scales = [0.6 + 0.05 * i for i in range(n_ctrl)]
        orig_half = torch.Tensor.half
        torch.Tensor.half = lambda self, *a, **k: self  # cldm.py:54-55,67-68,322-323 cast on non-'cuda' devices
        try:
            with fp32_forward():
                control = cn(x=x, hint=hint, timesteps=t, context=ctx)
                eps = un(x=x, timesteps=t, context=ctx, control=[c.clone() for c in control], only_mid_control=False)
                eps_scaled = un(x=x, timesteps=t, context=ctx, control=[c * s for c, s in zip(control, scales)], only_mid_control=False)
                eps_mid = un(x=x, timesteps=t, context=ctx, control=[c.clone() for c in control], only_mid_control=True)
        finally:
            torch.Tensor.half = orig_half
        assert len(control) == n_ctrl
        arrays = {f"control{i}": c for i, c in enumerate(control)}
        # parameter-name contract of the full-size ControlNet (cldm_v15.yaml control_stage_config), built on the meta device
        import hashlib
        with torch.device("meta"):
            full = ControlNet(hint_channels=3, **{k: v for k, v in SD15_UNET.items() if k != "out_channels"})
        items = sorted(f"{k}:{tuple(v.shape)}" for k, v in full.state_dict().items())
        save(name, dict(cfg=cfg, B=B, L=L, seed=SEED, cn_prefix="cn.", unet_prefix="unet.", n_control=n_ctrl, scales=scales,
                        cn_sd15_sha1=hashlib.sha1("\n".join(items).encode()).hexdigest(), cn_sd15_n=len(items)),
             t=t, eps=eps, eps_scaled=eps_scaled, eps_mid=eps_mid, **arrays)


def g_controlnet_sd15_full():
    """Full-size ControlNet (361 M params, cldm_v15.yaml) + ControlledUnetModel at B=2 (one image x CFG), L=64, 512x512 hint:
    eps in full, every control tensor sub-sampled (stride 4 spatially) to keep the fixture small."""
    from cldm.cldm import ControlledUnetModel, ControlNet
    cfg = SD15_UNET
    cn = ControlNet(hint_channels=3, **{k: v for k, v in cfg.items() if k != "out_channels"})
    un = ControlledUnetModel(**cfg)
    synth_fill_(cn, SEED, prefix="cn.")
    synth_fill_(un, SEED, prefix="unet.")
    B, L = 2, 64
    x = synth_input("cnet.x", (B, 4, L, L), SEED)
    hint = synth_input("cnet.hint", (B, 3, 8 * L, 8 * L), SEED, 0.5).clamp(-1, 1) * 0.5 + 0.5
    ctx = synth_input("cnet.ctx", (B, 77, cfg["context_dim"]), SEED)
    t = torch.tensor([801.0, 801.0])
    orig_half = torch.Tensor.half
    torch.Tensor.half = lambda self, *a, **k: self
    try:
        with fp32_forward():
            control = cn(x=x, hint=hint, timesteps=t, context=ctx)
            eps = un(x=x, timesteps=t, context=ctx, control=[c.clone() for c in control], only_mid_control=False)
    finally:
        torch.Tensor.half = orig_half
    arrays = {f"control{i}_sub": c[:, :, ::4, ::4].contiguous() for i, c in enumerate(control)}
    save("controlnet_sd15_full", dict(cfg=cfg, B=B, L=L, seed=SEED, cn_prefix="cn.", unet_prefix="unet.", n_control=len(control),
                                      n_params=sum(p.numel() for p in cn.parameters())), t=t, eps=eps, **arrays)


# ---------------------------------------------------------------------------- SDXL (sgm)

TINY_SGM_UNET = dict(adm_in_channels=96, num_classes="sequential", use_checkpoint=False, in_channels=4, out_channels=4,
                     model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1, channel_mult=[1, 2], num_head_channels=32,
                     use_linear_in_transformer=True, transformer_depth=[1, 2], context_dim=128,
                     spatial_transformer_attn_type="softmax-xformers")
SMALL_SDXL_UNET = dict(adm_in_channels=2816, num_classes="sequential", use_checkpoint=False, in_channels=4, out_channels=4,
                       model_channels=64, attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64,
                       use_linear_in_transformer=True, transformer_depth=[1, 2, 10], context_dim=2048,
                       spatial_transformer_attn_type="softmax-xformers")
SDXL_UNET = dict(SMALL_SDXL_UNET, model_channels=320)


def _import_sgm():
    SG = "/root/reference/modules/sdxl"
    if SG not in sys.path:
        sys.path.insert(0, SG)
    for name, path in [("sgm", SG + "/sgm"), ("sgm.modules", SG + "/sgm/modules"),
                       ("sgm.modules.diffusionmodules", SG + "/sgm/modules/diffusionmodules")]:
        if name not in sys.modules:  # bare packages: their __init__ pull in lightning / open_clip
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    from sgm.modules.diffusionmodules import denoiser, discretizer, guiders, openaimodel, sampling, wrappers
    return openaimodel, denoiser, discretizer, guiders, sampling, wrappers


def _sgm_unet_case(name, cfg, B, L, mctx, tvals):
    SU = _import_sgm()[0]
    m = SU.UNetModel(**cfg)
    synth_fill_(m, SEED, prefix="sgm_unet.")
    x = synth_input(name + ".x", (B, 4, L, L), SEED)
    ctx = synth_input(name + ".ctx", (B, mctx, cfg["context_dim"]), SEED)
    y = synth_input(name + ".y", (B, cfg["adm_in_channels"]), SEED)
    t = torch.tensor(tvals, dtype=torch.float32)
    t0 = time.time()
    with fp32_forward():
        out = m(x, timesteps=t, context=ctx, y=y)
    dt = time.time() - t0
    import hashlib
    items = sorted(f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items())
    save(name, dict(cfg=cfg, B=B, L=L, m=mctx, seed=SEED, prefix="sgm_unet.", n_params=sum(p.numel() for p in m.parameters()),
                    n_keys=len(items), keys_sha1=hashlib.sha1("\n".join(items).encode()).hexdigest(), ref_cpu_seconds=dt,
                    threads=torch.get_num_threads()), t=t, y=out)
    return m


def g_sgm_unet_tiny():
    _sgm_unet_case("sgm_unet_tiny", TINY_SGM_UNET, 2, 16, 77, [10.0, 731.0])


def g_sgm_unet_small():
    _sgm_unet_case("sgm_unet_small_sdxl", SMALL_SDXL_UNET, 2, 16, 77, [3.0, 900.0])


def g_sgm_unet_full():
    _sgm_unet_case("sgm_unet_sdxl_full", SDXL_UNET, 2, 128, 77, [981.0, 981.0])


def g_sgm_trajectory():
    """5-step EulerEDMSampler through the reference's DiscreteDenoiser(EpsScaling, LegacyDDPMDiscretization) + VanillaCFG
    + OpenAIWrapper + sgm UNetModel (tiny), then AutoencoderKL decode with scale_factor 0.13025."""
    SU, DN, DZ, GD, SM, WR = _import_sgm()
    SM.denoising_status_queue = types.SimpleNamespace(put=lambda *a, **k: None)
    unet = SU.UNetModel(**TINY_SGM_UNET)
    synth_fill_(unet, SEED, prefix="sgm_unet.")
    model = WR.OpenAIWrapper(unet)
    den = DN.DiscreteDenoiser(scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
                              discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"})
    smp = SM.EulerEDMSampler(discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
                             num_steps=5, guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG",
                                                         "params": {"scale": 5.0}}, device="cpu")
    B, L = 2, 16
    c = {"crossattn": synth_input("sgmtraj.c", (B, 77, 128), SEED), "vector": synth_input("sgmtraj.cv", (B, 96), SEED)}
    uc = {"crossattn": synth_input("sgmtraj.uc", (B, 77, 128), SEED), "vector": synth_input("sgmtraj.ucv", (B, 96), SEED)}
    x0 = synth_input("sgmtraj.x0", (B, 4, L, L), SEED)
    ae = _make_ae(TINY_DD)
    with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
        x = smp(lambda inp, sigma, cc: den(model, inp, sigma, cc), x0.clone(), cond=c, uc=uc)  # sdxl_image_generator_utils.py:703-707
        img = ae.decode(x / 0.13025)
    save("traj_sdxl_euler_edm", dict(B=B, L=L, S=5, cfg=5.0, seed=SEED, unet=TINY_SGM_UNET, dd=TINY_DD, scale_factor=0.13025),
         sigmas=smp.discretization(5), table=den.sigmas, x=x, img=img)


def g_sgm_img2img():
    """SDXL img2img (the face-fix re-entry of config 5): the numeric core of do_img2img
    (sdxl_pipeline/sdxl_image_generator_utils.py:906-1025) through the reference's own pieces - AutoencoderKL encoder moments ->
    posterior sample x scale_factor (sgm/models/diffusion.py:139-151), Img2ImgDiscretizationWrapper pruning
    (scripts/demo/discretization.py:11-32), noising z + noise * sigma_0 then / sqrt(1 + sigma_0^2) (:1003-1009), EulerEDMSampler
    over the pruned sigmas with VanillaCFG, decode."""
    SU, DN, DZ, GD, SM, WR = _import_sgm()
    SM.denoising_status_queue = types.SimpleNamespace(put=lambda *a, **k: None)
    from scripts.demo.discretization import Img2ImgDiscretizationWrapper
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    unet = SU.UNetModel(**TINY_SGM_UNET)
    synth_fill_(unet, SEED, prefix="sgm_unet.")
    model = WR.OpenAIWrapper(unet)
    den = DN.DiscreteDenoiser(scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
                              discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"})
    S, strength = 10, 0.6
    smp = SM.EulerEDMSampler(discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
                             num_steps=S, guider_config={"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG",
                                                         "params": {"scale": 5.0}}, device="cpu")
    smp.discretization = Img2ImgDiscretizationWrapper(smp.discretization, strength=strength)  # sdxl_image_generator_utils.py:398-403
    B, L = 2, 16
    c = {"crossattn": synth_input("sgmi2i.c", (B, 77, 128), SEED), "vector": synth_input("sgmi2i.cv", (B, 96), SEED)}
    uc = {"crossattn": synth_input("sgmi2i.uc", (B, 77, 128), SEED), "vector": synth_input("sgmi2i.ucv", (B, 96), SEED)}
    img = synth_input("sgmi2i.img", (B, 3, 2 * L, 2 * L), SEED, 0.5).clamp(-1, 1)  # TINY_DD has one down level: 32x32 -> 16x16
    enc_noise = synth_input("sgmi2i.enc_noise", (B, 4, L, L), SEED)
    noise = synth_input("sgmi2i.noise", (B, 4, L, L), SEED)
    ae = _make_ae(TINY_DD)
    with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
        post = DiagonalGaussianDistribution(ae.quant_conv(ae.encoder(img)))
        z = 0.13025 * (post.mean + post.std * enc_noise)        # posterior.sample() with the noise made explicit
        sigmas = smp.discretization(smp.num_steps)
        noised_z = (z + noise * sigmas[0]) / torch.sqrt(1.0 + sigmas[0] ** 2.0)
        x = smp(lambda inp, sigma, cc: den(model, inp, sigma, cc), noised_z.clone(), cond=c, uc=uc)
        out = ae.decode(x / 0.13025)
    save("traj_sdxl_img2img", dict(B=B, L=L, S=S, strength=strength, cfg=5.0, seed=SEED, unet=TINY_SGM_UNET, dd=TINY_DD, scale_factor=0.13025),
         sigmas=sigmas, z=z, noised_z=noised_z, x=x, img=out)


def g_sgm_vae():
    """The SDXL first stage as the reference builds it (sd_xl_base.yaml:77-92 -> sgm/modules/diffusionmodules/model.py:
    Encoder :492-612, Decoder :614-760, AttnBlock :161-195, make_attn :277-309).  attn_type: the YAML asks for
    "vanilla-xformers" (MemoryEfficientAttnBlock, :198-266), which needs the absent xformers package; "vanilla" builds AttnBlock -
    the same parameters (norm, q, k, v, proj_out) and the same single-head softmax(q k^T / sqrt(C)) v through SDPA - so the
    fixtures are generated with it and the HIP path is checked with BOTH attn_type spellings against them."""
    _import_sgm()
    from sgm.modules.diffusionmodules import model as SV
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution as DG

    def build(dd):
        enc, dec = SV.Encoder(**dd), SV.Decoder(**dd)
        qc, pqc = torch.nn.Conv2d(2 * dd["z_channels"], 2 * 4, 1), torch.nn.Conv2d(4, dd["z_channels"], 1)
        box = torch.nn.Module()  # same key layout as AutoencoderKL: encoder.*, decoder.*, quant_conv.*, post_quant_conv.*
        box.encoder, box.decoder, box.quant_conv, box.post_quant_conv = enc, dec, qc, pqc
        synth_fill_(box, SEED, prefix="vae.")
        return box.eval()

    dd = dict(TINY_DD, attn_type="vanilla")
    ae = build(dd)
    z = synth_input("sgm_vae_tiny.z", (2, 4, 8, 8), SEED)
    img = synth_input("sgm_vae_tiny.img", (2, 3, 16, 16), SEED, 0.5).clamp(-1, 1)
    noise = synth_input("sgm_vae_tiny.noise", (2, 4, 8, 8), SEED)
    with fp32_forward():
        dec = ae.decoder(ae.post_quant_conv(z))
        moments = ae.quant_conv(ae.encoder(img))
        post = DG(moments)
        sample = post.mean + post.std * noise
    import hashlib
    items = sorted(f"{k}:{tuple(v.shape)}" for k, v in ae.state_dict().items())
    save("sgm_vae_tiny", dict(dd=dd, seed=SEED, prefix="vae.", n_keys=len(items), keys_sha1=hashlib.sha1("\n".join(items).encode()).hexdigest()),
         dec=dec, moments=moments, sample=sample)


def g_sgm_vae_full():
    """Full-size SDXL VAE (sd_xl_base.yaml ddconfig) decode of a 128x128 latent -> 1024x1024 (fp32, as the reference runs it:
    sd_xl_base.yaml:5 disable_first_stage_autocast), kept as an fp16 image of every 8th pixel row / column plus the exact
    fp32 values on a 16x coarser grid, and an encode at 256x256."""
    _import_sgm()
    from sgm.modules.diffusionmodules import model as SV
    dd = dict(SD15_DD, attn_type="vanilla")
    box = torch.nn.Module()
    box.encoder, box.decoder = SV.Encoder(**dd), SV.Decoder(**dd)
    box.quant_conv, box.post_quant_conv = torch.nn.Conv2d(8, 8, 1), torch.nn.Conv2d(4, 4, 1)
    synth_fill_(box, SEED, prefix="vae.")
    box.eval()
    z = synth_input("sgm_vae_full.z", (1, 4, 128, 128), SEED)
    t0 = time.time()
    with fp32_forward():
        dec = box.decoder(box.post_quant_conv(z / 0.13025))
    dt = time.time() - t0
    x = synth_input("sgm_vae_full.img", (1, 3, 256, 256), SEED, 0.5).clamp(-1, 1)
    with fp32_forward():
        moments = box.quant_conv(box.encoder(x))
    save("sgm_vae_full", dict(dd=dd, seed=SEED, prefix="vae.", ref_cpu_seconds=dt, threads=torch.get_num_threads(), scale_factor=0.13025, hw=256),
         dec_sub8_f16=dec[:, :, ::8, ::8].half(), dec_sub16=dec[:, :, ::16, ::16].contiguous(),
         dec_stats=torch.tensor([dec.mean().item(), dec.std().item(), dec.abs().max().item()]), moments=moments)


def g_c1_sd15_full_trajectory():
    """BASELINE.json configs[0] ("C1") at FULL size: SD1.5 txt2img 512x512, batch 1, 20-step Euler, CFG 7.5, fp32 on the CPU,
    through the reference's own stack (EulerSampler -> k_diffusion.sample_euler -> LDMWrapperForKDiffusion (CFG batch doubling)
    -> CompVisDenoiser -> LatentDiffusion.apply_model -> DiffusionWrapper -> UNetModel, 859.52 M parameters) and
    decode_first_stage with the full 49.49 M-parameter decoder.  The final latent, a few latents on the way (the drift of a
    bf16 run is a function of the step) and the image (every 8th pixel exact, all pixels as fp16) are the fixture the GPU
    path's accumulated 20-step error is measured against (north_star: "stated fp32 per-pixel tolerance")."""
    ldm = LatentDiffusion(first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                                              "params": dict(ddconfig=SD15_DD, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)},
                          cond_stage_config={"target": "torch.nn.Identity"},
                          unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": SD15_UNET},
                          linear_start=0.00085, linear_end=0.012, timesteps=1000, conditioning_key="crossattn",
                          scale_factor=0.18215, use_ema=False, cond_stage_trainable=False, first_stage_key="jpg",
                          cond_stage_key="txt", image_size=64, channels=4)
    synth_fill_(ldm.model.diffusion_model, SEED, prefix="unet.")
    synth_fill_(ldm.first_stage_model, SEED, prefix="vae.")
    ldm.eval()
    B, L, S = 1, 64, 20
    c = synth_input("c1.c", (B, 77, 768), SEED)
    uc = synth_input("c1.uc", (B, 77, 768), SEED)
    x0 = synth_input("c1.x0", (B, 4, L, L), SEED)
    R_ks.trange = R_ksamp.trange = lambda *a, **k: range(*a)
    s = R_ks.EulerSampler(ldm)
    mids = {}
    orig_model_call = R_kext.CompVisDenoiser.forward
    calls = [0]

    def spy(self, inp, sigma, **kw):  # records the latent that enters UNet call i (B = 2: CFG doubling) - no arithmetic changed
        i = calls[0]
        if i in (5, 10, 15):
            mids[i] = inp[:B].clone()
        calls[0] += 1
        return orig_model_call(self, inp, sigma, **kw)

    R_kext.CompVisDenoiser.forward = spy
    t0 = time.time()
    try:
        with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
            s._sample_common_prep(S=S, batch_size=B, shape=[4, L, L], conditioning=c, unconditional_guidance_scale=7.5,
                                  unconditional_conditioning=uc, x0=x0)
            x, _ = s.do_sample()
            t_unet = time.time() - t0
            img = ldm.decode_first_stage(x)
    finally:
        R_kext.CompVisDenoiser.forward = orig_model_call
    dt = time.time() - t0
    pix = torch.clamp((img + 1.0) / 2.0, 0.0, 1.0)
    save("traj_c1_sd15_full", dict(B=B, L=L, S=S, cfg=7.5, seed=SEED, unet=SD15_UNET, dd=SD15_DD, sampler="euler", ref_cpu_seconds=dt,
                                   ref_cpu_seconds_unet=t_unet, threads=torch.get_num_threads(), unet_calls=calls[0]),
         sigmas=s.sigmas, x=x, x5=mids[5], x10=mids[10], x15=mids[15], img_sub8=img[:, :, ::8, ::8].contiguous(), img_f16=img.half(),
         pix_stats=torch.tensor([pix.mean().item(), pix.std().item(), img.abs().max().item()]))


def g_c5_chain():
    """BASELINE config 5 as ONE chain at tiny size, every numeric piece the reference's own: SDXL txt2img (EulerEDMSampler +
    DiscreteDenoiser + VanillaCFG + sgm UNet, 5 steps) -> decode -> clamp((x+1)/2) -> crop a fixed box -> bilinear resize to
    the generation size -> img2img re-entry (Img2ImgDiscretizationWrapper strength 0.3 over 10 steps = 3 UNet steps; encoder
    moments -> posterior sample; Euler-EDM; decode) -> resize back -> paste.  The box replaces the face detector and the two
    resizes are torch bilinear instead of cv2 Lanczos (both outside the numeric path; see pipeline.txt2img_sdxl_facefix)."""
    import torch.nn.functional as F
    SU, DN, DZ, GD, SM, WR = _import_sgm()
    SM.denoising_status_queue = types.SimpleNamespace(put=lambda *a, **k: None)
    from scripts.demo.discretization import Img2ImgDiscretizationWrapper
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    unet = SU.UNetModel(**TINY_SGM_UNET)
    synth_fill_(unet, SEED, prefix="sgm_unet.")
    model = WR.OpenAIWrapper(unet)
    disc = {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}
    den = DN.DiscreteDenoiser(scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
                              discretization_config=disc)
    guider = {"target": "sgm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 5.0}}
    B, L, S1, S2, strength = 2, 16, 5, 10, 0.3
    boxes = [(4, 6, 16), (12, 8, 16)]  # (top, left, size) in the 32x32 first-pass image
    c = {"crossattn": synth_input("c5.c", (B, 77, 128), SEED), "vector": synth_input("c5.cv", (B, 96), SEED)}
    uc = {"crossattn": synth_input("c5.uc", (B, 77, 128), SEED), "vector": synth_input("c5.ucv", (B, 96), SEED)}
    x0 = synth_input("c5.x0", (B, 4, L, L), SEED)
    enc_noise = synth_input("c5.enc_noise", (B, 4, L, L), SEED)
    noise = synth_input("c5.noise", (B, 4, L, L), SEED)
    ae = _make_ae(TINY_DD)
    with fp32_forward(), contextlib.redirect_stdout(open(os.devnull, "w")):
        smp = SM.EulerEDMSampler(discretization_config=disc, num_steps=S1, guider_config=guider, device="cpu")
        x1 = smp(lambda inp, sigma, cc: den(model, inp, sigma, cc), x0.clone(), cond=c, uc=uc)
        first = torch.clamp((ae.decode(x1 / 0.13025) + 1.0) / 2.0, 0.0, 1.0)
        crops = torch.stack([F.interpolate(first[i:i + 1, :, t:t + sz, l:l + sz], size=(2 * L, 2 * L), mode="bilinear", align_corners=False)[0]
                             for i, (t, l, sz) in enumerate(boxes)])
        smp2 = SM.EulerEDMSampler(discretization_config=disc, num_steps=S2, guider_config=guider, device="cpu")
        smp2.discretization = Img2ImgDiscretizationWrapper(smp2.discretization, strength=strength)
        post = DiagonalGaussianDistribution(ae.quant_conv(ae.encoder(crops * 2.0 - 1.0)))
        z = 0.13025 * (post.mean + post.std * enc_noise)
        sigmas = smp2.discretization(smp2.num_steps)
        noised_z = (z + noise * sigmas[0]) / torch.sqrt(1.0 + sigmas[0] ** 2.0)
        x2 = smp2(lambda inp, sigma, cc: den(model, inp, sigma, cc), noised_z.clone(), cond=c, uc=uc)
        fixed = torch.clamp((ae.decode(x2 / 0.13025) + 1.0) / 2.0, 0.0, 1.0)
        out = first.clone()
        for i, (t, l, sz) in enumerate(boxes):
            out[i, :, t:t + sz, l:l + sz] = F.interpolate(fixed[i:i + 1], size=(sz, sz), mode="bilinear", align_corners=False)[0]
    save("traj_c5_chain", dict(B=B, L=L, S1=S1, S2=S2, strength=strength, cfg=5.0, seed=SEED, unet=TINY_SGM_UNET, dd=TINY_DD,
                               scale_factor=0.13025, boxes=boxes, second_pass_sigmas=len(sigmas)),
         x1=x1, first=first, z=z, x2=x2, fixed=fixed, out=out)


CASES = dict(alphas_doc=g_alphas_doc, param_contract=g_param_contract, groupnorm=g_groupnorm, timestep_embedding=g_timestep_embedding, resblock=g_resblock, updown=g_updown,
             attention=g_attention, transformer=g_transformer, unet_tiny=g_unet_tiny, unet_small_sd=g_unet_small_sd,
             vae_blocks=g_vae_blocks, vae_tiny=g_vae_tiny, schedules=g_schedules, trajectories=g_trajectories, hires_latent=g_hires_latent)
CASES.update(controlnet_hook=g_controlnet_hook, controlnet=g_controlnet)
CASES.update(sgm_unet_tiny=g_sgm_unet_tiny, sgm_unet_small=g_sgm_unet_small, sgm_trajectory=g_sgm_trajectory, sgm_img2img=g_sgm_img2img,
             sgm_vae=g_sgm_vae, c5_chain=g_c5_chain)
FULL = dict(controlnet_sd15_full=g_controlnet_sd15_full, unet_sd15_full=g_unet_sd15_full, vae_sd15_full=g_vae_sd15_full, sgm_unet_full=g_sgm_unet_full,
            sgm_vae_full=g_sgm_vae_full, c1_sd15_full_trajectory=g_c1_sd15_full_trajectory)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--full", action="store_true", help="also run the full-size SD1.5 UNet / VAE cases (minutes)")
    a = ap.parse_args()
    todo = dict(CASES)
    if a.full:
        todo.update(FULL)
    if a.only:
        allc = dict(CASES, **FULL)
        todo = {k: allc[k] for k in a.only}
    for k, fn in todo.items():
        t0 = time.time()
        fn()
        print(f"[golden] case {k} done in {time.time()-t0:.1f}s")
