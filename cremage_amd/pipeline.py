"""txt2img / img2img drivers over the HIP UNet + VAE: the numeric core of the reference's call sites
`sd.txt2img.generate` / `sd.img2img.generate` -> `image_generator.generate`
(modules/sd/image_generator.py:569-1268), without the app shell (prompt parsing, CLIP, PNG metadata, UI
queues - all out of scope).  Conditioning arrives as tensors ([b, 77*n, 768]), exactly what
`model.get_learned_conditioning` hands to the sampler at image_generator.py:785-808.

Step structure reproduced (file:line of the reference):
  txt2img : sampler.sample(S, conditioning=c, batch_size, shape=[4,H/8,W/8], cfg, uc)      :958-967
            decode_first_stage per image, clamp((x+1)/2, 0, 1)                             :1007-1015
  img2img : encode_first_stage -> get_first_stage_encoding (posterior sample * 0.18215)    :721
            t_enc = int(strength * steps); DDIM stochastic_encode + decode                 :727, :147-248
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import ops
from .ldm_hip.latent_diffusion import LatentDiffusion
from .ldm_hip.unet import UNetModel
from .ldm_hip.vae import AutoencoderKL
from .samplers import DDIMSampler, EulerAncestralSampler, EulerSampler
from .synth import synth_fill_

SD15_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                 num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
                 context_dim=768, use_checkpoint=True, legacy=False)  # v1-inference.yaml:29-44
SD15_VAE_DD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                   num_res_blocks=2, attn_resolutions=[], dropout=0.0)  # v1-inference.yaml:51-65

SAMPLERS = {"euler": EulerSampler, "euler_a": EulerAncestralSampler}  # sampler_utils.py:36-66 names "Euler", "Euler a"


def build_synthetic_ldm(unet_cfg=None, vae_dd=None, device="cuda", unet_dtype=torch.bfloat16, vae_dtype=torch.float32,
                        seed: int = 1234) -> LatentDiffusion:
    """LatentDiffusion with name-keyed synthetic weights (no checkpoint exists offline).
    Parameters are created on CPU in fp32, filled, then cast/moved - the same order as the reference's
    load_model_from_config (load_state_dict -> .half() -> .to(device), image_generator.py:345,489-493)."""
    unet = UNetModel(**(unet_cfg or SD15_UNET))
    vae = AutoencoderKL(vae_dd or SD15_VAE_DD, None, 4)
    synth_fill_(unet, seed, prefix="unet.")
    synth_fill_(vae, seed, prefix="vae.")
    ldm = LatentDiffusion(unet, vae)
    ldm.model.to(unet_dtype)
    ldm.first_stage_model.to(vae_dtype)
    return ldm.to(device).eval()


def build_ldm_sharded(rank: int, device, unet_cfg=None, vae_dd=None, unet_dtype=torch.bfloat16, vae_dtype=torch.float32, seed: int = 1234):
    """The model of one rank of a batch-sharded run (bench.py --gpus N; tests/test_dist_cpu.py runs this very function under gloo):
    rank 0 materialises the synthetic weights, every other rank builds the same module tree UNINITIALISED in the same dtypes and
    receives parameters + buffers through ONE flat broadcast per (dtype, device) bucket (dist.broadcast_module_, whose header check
    makes every rank raise if the trees differ).  Returns (ldm, bytes broadcast)."""
    from . import dist as D
    if rank == 0:
        ldm = build_synthetic_ldm(unet_cfg, vae_dd, device=device, unet_dtype=unet_dtype, vae_dtype=vae_dtype, seed=seed)
    else:
        ldm = LatentDiffusion(UNetModel(**(unet_cfg or SD15_UNET)), AutoencoderKL(vae_dd or SD15_VAE_DD, None, 4))
        ldm.model.to(unet_dtype)
        ldm.first_stage_model.to(vae_dtype)
        ldm = ldm.to(device).eval()
    return ldm, D.broadcast_module_(ldm, src=0)


def build_synthetic_control_ldm(unet_cfg=None, vae_dd=None, device="cuda", unet_dtype=torch.bfloat16, vae_dtype=torch.float32, seed: int = 1234):
    """ControlLDM (cldm.py:345-393, cldm_v15.yaml) with name-keyed synthetic weights: ControlledUnetModel + ControlNet + VAE."""
    from .cldm_hip import ControlLDM, ControlledUnetModel, ControlNet
    cfg = unet_cfg or SD15_UNET
    unet = ControlledUnetModel(**cfg)
    cnet = ControlNet(hint_channels=3, **{k: v for k, v in cfg.items() if k != "out_channels"})
    vae = AutoencoderKL(vae_dd or SD15_VAE_DD, None, 4)
    synth_fill_(unet, seed, prefix="unet.")
    synth_fill_(cnet, seed, prefix="cn.")
    synth_fill_(vae, seed, prefix="vae.")
    ldm = ControlLDM(cnet, "hint", False, unet, vae)
    ldm.model.to(unet_dtype)
    ldm.control_model.to(unet_dtype)
    ldm.first_stage_model.to(vae_dtype)
    return ldm.to(device).eval()


@torch.no_grad()
def decode_images(ldm: LatentDiffusion, samples: torch.Tensor, batch_decode: bool = True) -> torch.Tensor:
    """latents [b,4,L,L] -> images [b,3,8L,8L] fp32 in [0,1] (image_generator.py:1007-1015).  The reference
    decodes one image at a time to fit 8-24 GB cards (`save_memory`, options.py:268-273); with 288 GB the
    whole batch is decoded in one pass unless `batch_decode=False`."""
    if batch_decode:
        x = ldm.decode_first_stage(samples)
    else:
        x = torch.cat([ldm.decode_first_stage(s[None]) for s in samples])
    return ops.affine_cast(x, 0.5, 0.5, torch.float32, 0.0, 1.0)


@torch.no_grad()
def trajectory_noise_sampler(steps: int, shape, device, generators=None) -> Callable:
    """Ancestral noise of a whole sampling run drawn UP FRONT - one randn per image generator (or one for the batch from the default
    generator) instead of one per image and step - and handed out step by step: the same number of Gaussian samples as the per-step
    `torch.randn_like(x)` of k-diffusion's default noise sampler (sampling.py:147-163), in 1 / b launches instead of `steps` / b * steps.
    The last step of a schedule that ends at sigma = 0 draws nothing, as in the reference."""
    if generators is not None:
        noise = torch.stack([torch.randn((steps,) + tuple(shape[1:]), generator=g, device=device) for g in generators], dim=1)
    else:
        noise = torch.randn((steps,) + tuple(shape), device=device)
    it = iter(range(steps))
    return lambda sigma, sigma_next: noise[next(it)]


@torch.no_grad()
def txt2img(ldm: LatentDiffusion, c: torch.Tensor, uc: Optional[torch.Tensor], *, steps: int = 20, sampler: str = "euler_a",
            cfg_scale: float = 7.5, height: int = 512, width: int = 512, x0: Optional[torch.Tensor] = None,
            noise_sampler: Optional[Callable] = None, decode: bool = True, hint: Optional[torch.Tensor] = None, generators=None):
    """Returns (images or None, final latents).  `hint` ([b,3,H,W] in [0,1], ControlLDM only): the ControlNet control image;
    conditioning becomes {"c_crossattn": [c], "c_concat": [hint]} for both the positive and the negative prompt
    (image_generator.py:795-808).  `generators`: one torch.Generator per image (seed + global image index, cremage_amd.dist.image_seed):
    the initial latents (unless x0 is given) and the ancestral noise of image i come from generators[i], so that a batch sharded over
    GPUs draws what the unsharded batch draws.  Without an explicit `noise_sampler` the ancestral sampler's noise is drawn once per run
    (trajectory_noise_sampler)."""
    b = c.shape[0]
    dev = c.device
    if x0 is None and generators is not None:
        x0 = torch.stack([torch.randn((4, height // 8, width // 8), generator=g, device=dev) for g in generators])
    if noise_sampler is None and sampler == "euler_a" and dev.type == "cuda":
        noise_sampler = trajectory_noise_sampler(steps, (b, 4, height // 8, width // 8), dev, generators)
    if hint is not None:
        c = {"c_crossattn": [c], "c_concat": [hint]}
        uc = {"c_crossattn": [uc], "c_concat": [hint]} if uc is not None else None
    shape = [4, height // 8, width // 8]
    smp = SAMPLERS[sampler](ldm)
    smp.noise_sampler = noise_sampler
    samples, _ = smp.sample(S=steps, conditioning=c, batch_size=b, shape=shape, verbose=False,
                            unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uc, x0=x0)
    return (decode_images(ldm, samples) if decode else None), samples


@torch.no_grad()
def txt2img_hires(ldm: LatentDiffusion, c: torch.Tensor, uc: Optional[torch.Tensor], *, steps: int = 20, sampler: str = "euler_a",
                  cfg_scale: float = 7.5, height: int = 512, width: int = 512, factor: float = 2.0, strength: float = 0.5,
                  x0: Optional[torch.Tensor] = None, fwd_noise: Optional[torch.Tensor] = None, noise_sampler: Optional[Callable] = None,
                  decode: bool = True):
    """Hires-fix with the latent upscaler (image_generator.py:958-999): txt2img at (height, width); bilinear upscale of the final
    latents by `factor`; forward-diffuse to t_enc = int(strength * steps) (k_diffusion_samplers.py:255-296) and denoise the last
    t_enc + 1 sigmas at the larger size (img2img_sampling :227-246).  Returns (images, latents, base latents)."""
    import torch.nn.functional as F
    b = c.shape[0]
    smp = SAMPLERS[sampler](ldm)
    smp.noise_sampler = noise_sampler
    base, _ = smp.sample(S=steps, conditioning=c, batch_size=b, shape=[4, height // 8, width // 8], verbose=False,
                         unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uc, x0=x0)
    up = F.interpolate(base, scale_factor=factor, mode="bilinear", align_corners=False)
    t_enc = int(strength * steps)
    z_enc = smp.stochastic_encode(up, torch.tensor([t_enc] * b, device=up.device), sampling_steps=steps, noise=fwd_noise)
    samples, _ = smp.sample(S=steps, conditioning=c, batch_size=b, shape=list(up.shape[1:]), verbose=False,
                            unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uc, x0=z_enc, denoising_steps=t_enc)
    return (decode_images(ldm, samples) if decode else None), samples, base


@torch.no_grad()
def img2img(ldm: LatentDiffusion, init_image: torch.Tensor, c: torch.Tensor, uc: Optional[torch.Tensor], *, steps: int = 20,
            strength: float = 0.75, cfg_scale: float = 7.5, enc_noise: Optional[torch.Tensor] = None,
            fwd_noise: Optional[torch.Tensor] = None, decode: bool = True):
    """SD1.5 img2img as the reference drives it: sampler forced to DDIM (image_generator.py:679-681)."""
    b = init_image.shape[0]
    init_latent = ldm.get_first_stage_encoding(ldm.encode_first_stage(init_image), enc_noise)
    t_enc = int(strength * steps)
    smp = DDIMSampler(ldm)
    smp.make_schedule(ddim_num_steps=steps, ddim_eta=0.0, verbose=False)
    z_enc = smp.stochastic_encode(init_latent, torch.tensor([t_enc] * b, device=init_latent.device), noise=fwd_noise)
    samples = smp.decode(z_enc, c, t_enc, unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uc)
    return (decode_images(ldm, samples) if decode else None), samples


# ---------------------------------------------------------------------------------------------- SDXL (config 3)
SDXL_UNET = dict(adm_in_channels=2816, num_classes="sequential", use_checkpoint=True, in_channels=4, out_channels=4,
                 model_channels=320, attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64,
                 use_linear_in_transformer=True, transformer_depth=[1, 2, 10], context_dim=2048,
                 spatial_transformer_attn_type="softmax-xformers")  # sd_xl_base.yaml:17-33
SDXL_VAE_DD = dict(SD15_VAE_DD, attn_type="vanilla-xformers")       # sd_xl_base.yaml:80-92


def build_synthetic_sdxl(unet_cfg=None, vae_dd=None, device="cuda", unet_dtype=torch.bfloat16, vae_dtype=torch.float32, seed: int = 1234,
                         fill: bool = True):
    """DiffusionEngine-like container with name-keyed synthetic weights (UNet fp16/bf16, VAE fp32 as the reference runs
    them: vram_mode.py:24-28, sd_xl_base.yaml:5)."""
    from .sgm_hip.sampling import DiffusionEngine
    from .sgm_hip.unet import UNetModel as SgmUNet
    unet = SgmUNet(**(unet_cfg or SDXL_UNET))
    vae = AutoencoderKL(vae_dd or SDXL_VAE_DD, None, 4)
    if fill:
        synth_fill_(unet, seed, prefix="sgm_unet.")
        synth_fill_(vae, seed, prefix="vae.")
    eng = DiffusionEngine(unet, vae, 0.13025)
    eng.model.to(unet_dtype)
    eng.first_stage_model.to(vae_dtype)
    return eng.to(device).eval()


@torch.no_grad()
def txt2img_sdxl(eng, c: dict, uc: dict, *, steps: int = 30, cfg_scale: float = 5.0, height: int = 1024, width: int = 1024,
                 x0: Optional[torch.Tensor] = None, decode: bool = True):
    """run_txt2img -> do_sample (modules/sdxl/sdxl_pipeline/sdxl_image_generator_utils.py:559-772): randn [b,4,H/8,W/8]
    (:695), sampler(denoiser, randn, cond=c, uc=uc) (:707), decode_first_stage in fp32 (:727-734), clamp((x+1)/2, 0, 1).
    c / uc: {"crossattn": [b,77,2048], "vector": [b,2816]}."""
    b = c["crossattn"].shape[0]
    if x0 is None:
        x0 = torch.randn((b, 4, height // 8, width // 8), device=c["crossattn"].device)
    samples = eng.sample(x0, c, uc, steps, cfg_scale)
    if not decode:
        return None, samples
    x = eng.decode_first_stage(samples)
    return ops.affine_cast(x, 0.5, 0.5, torch.float32, 0.0, 1.0), samples


@torch.no_grad()
def img2img_sdxl(eng, init_image: torch.Tensor, c: dict, uc: dict, *, steps: int = 30, strength: float = 0.3, cfg_scale: float = 5.0,
                 enc_noise: Optional[torch.Tensor] = None, fwd_noise: Optional[torch.Tensor] = None, decode: bool = True):
    """run_img2img -> do_img2img (sdxl_image_generator_utils.py:775-1025); with strength 0.3 on a face crop this is the second
    pass of the auto-face-fix (modules/sdxl/face_img2img... -> the same do_img2img).  init_image [b,3,H,W] in [-1,1]."""
    samples = eng.img2img(init_image, c, uc, steps, strength, cfg_scale, enc_noise=enc_noise, fwd_noise=fwd_noise)
    if not decode:
        return None, samples
    x = eng.decode_first_stage(samples)
    return ops.affine_cast(x, 0.5, 0.5, torch.float32, 0.0, 1.0), samples


@torch.no_grad()
def txt2img_sdxl_facefix(eng, c: dict, uc: dict, boxes, *, steps: int = 30, cfg_scale: float = 5.0, height: int = 1024, width: int = 1024,
                         fix_size: Optional[int] = None, strength: float = 0.3, x0: Optional[torch.Tensor] = None,
                         enc_noise: Optional[torch.Tensor] = None, fwd_noise: Optional[torch.Tensor] = None, paste: bool = True):
    """BASELINE config 5: SDXL txt2img, then the auto-face-fix second pass on one region per image - the UNet RE-ENTRY on a crop.

    Reference flow (modules/sdxl/sdxl_pipeline/sdxl_image_generator_utils.py:559-772 txt2img, then per detected face
    modules/face_detection/face_img2img.py:57-235 -> do_img2img :906-1025): crop the face box, bring the crop to the generation
    size, img2img it with strength 0.3 (encode -> noise to sigma_0 of the pruned schedule -> Euler-EDM over the last
    int(0.3 * steps) sigmas -> decode), bring it back to the box size and paste it over the first-pass image.
    In scope here is the numeric path (both UNet passes, VAE encode / decode); the glue around it is deliberately plain PyTorch:
    `boxes` = one (top, left, size) per image instead of the face detector (out of scope), the two resizes are
    F.interpolate(bilinear, antialias off) instead of cv2 Lanczos (not built), the paste is a hard-edged copy.
    Returns (final images [b,3,H,W] in [0,1], first-pass images, second-pass crops at `fix_size`)."""
    import torch.nn.functional as F
    first, _ = txt2img_sdxl(eng, c, uc, steps=steps, cfg_scale=cfg_scale, height=height, width=width, x0=x0)
    fix = fix_size or height
    crops = torch.stack([F.interpolate(first[i:i + 1, :, t:t + sz, l:l + sz], size=(fix, fix), mode="bilinear", align_corners=False)[0]
                         for i, (t, l, sz) in enumerate(boxes)])
    fixed, _ = img2img_sdxl(eng, crops * 2.0 - 1.0, c, uc, steps=steps, strength=strength, cfg_scale=cfg_scale, enc_noise=enc_noise,
                            fwd_noise=fwd_noise)
    out = first.clone()
    if paste:
        for i, (t, l, sz) in enumerate(boxes):
            out[i, :, t:t + sz, l:l + sz] = F.interpolate(fixed[i:i + 1], size=(sz, sz), mode="bilinear", align_corners=False)[0]
    return out, first, fixed


@torch.no_grad()
def face_fix_sdxl(eng, images: torch.Tensor, faces, c: dict, uc: dict, *, steps: int = 30, strength: float = 0.3, cfg_scale: float = 5.0,
                  target_edge_len: int = 1024, enc_noise: Optional[torch.Tensor] = None, fwd_noise: Optional[torch.Tensor] = None):
    """The auto-face-fix second pass with the REFERENCE's host-side glue (cremage_amd.postprocess: buffer / clamp / aspect-preserving
    Lanczos resize / white padding / un-pad / resize back / paste, face_detector_engine.py:152-288) around the UNet re-entry
    (`img2img_sdxl`, strength 0.3).  images [b,3,H,W] in [0,1]; faces[i] = list of (x, y, w, h) boxes of image i (the detector is out
    of scope).  Differences from the reference that remain: plain paste instead of cv.seamlessClone (no OpenCV here), and one
    conditioning row per image instead of a gender-prefixed prompt.  Returns [b,3,H,W] in [0,1] on the images' device."""
    from . import postprocess as PP
    out = []
    for i in range(images.shape[0]):
        ci = {k: v[i:i + 1] for k, v in c.items()}
        uci = {k: v[i:i + 1] for k, v in uc.items()}

        def i2i(x):
            y, _ = img2img_sdxl(eng, x.to(images.device), ci, uci, steps=steps, strength=strength, cfg_scale=cfg_scale,
                                enc_noise=enc_noise[i:i + 1] if enc_noise is not None else None,
                                fwd_noise=fwd_noise[i:i + 1] if fwd_noise is not None else None)
            return y
        pil = PP.face_fix(PP.unit_tensor_to_pil(images[i]), faces[i], i2i, target_edge_len)
        out.append((PP.pil_to_unit_tensor(pil)[0] + 1.0) * 0.5)
    return torch.stack(out).to(images.device)
