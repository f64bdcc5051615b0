"""SDXL denoiser / guider / sampler glue - stays on PyTorch (north_star), behaviour of
modules/sdxl/sgm/modules/diffusionmodules/{denoiser.py:10-75, denoiser_scaling.py:29-37, discretizer.py:17-78,
guiders.py:24-65, sampling.py:29-219,309-318, wrappers.py:24-34} and DiffusionEngine.decode_first_stage
(sgm/models/diffusion.py:118-136).  Tensors here are [b, 4, L, L] latents and per-sample scalars; the per-step cost
is the UNet call."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from ..samplers import append_dims, append_zero, make_beta_schedule


class LegacyDDPMDiscretization:
    """discretizer.py:51-78 (+ Discretization.__call__ :17-24)."""

    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = make_beta_schedule("linear", num_timesteps, linear_start=linear_start, linear_end=linear_end)
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            timesteps = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
            alphas_cumprod = self.alphas_cumprod[timesteps]
        elif n == self.num_timesteps:
            alphas_cumprod = self.alphas_cumprod
        else:
            raise ValueError
        sigmas = torch.tensor((1 - alphas_cumprod) / alphas_cumprod, dtype=torch.float32, device=device) ** 0.5
        return torch.flip(sigmas, (0,))

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        sigmas = append_zero(sigmas) if do_append_zero else sigmas
        return sigmas if not flip else torch.flip(sigmas, (0,))


class EpsScaling:
    """denoiser_scaling.py:29-37."""

    def __call__(self, sigma):
        c_skip = torch.ones_like(sigma, device=sigma.device)
        c_out = -sigma
        c_in = 1 / (sigma ** 2 + 1.0) ** 0.5
        c_noise = sigma.clone()
        return c_skip, c_out, c_in, c_noise


class DiscreteDenoiser(nn.Module):
    """denoiser.py:10-75: sigma (and c_noise) quantised to the 1000-entry DDPM table; c_noise becomes the table index."""

    def __init__(self, num_idx: int = 1000, quantize_c_noise: bool = True):
        super().__init__()
        self.scaling = EpsScaling()
        self.discretization = LegacyDDPMDiscretization()
        self.register_buffer("sigmas", self.discretization(num_idx, do_append_zero=False, flip=True))
        self.quantize_c_noise = quantize_c_noise
        self.num_idx = num_idx

    def sigma_to_idx(self, sigma):
        dists = sigma - self.sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self.sigmas[idx]

    def forward(self, network, input, sigma, cond: Dict, **additional_model_inputs):
        sigma = self.idx_to_sigma(self.sigma_to_idx(sigma))
        sigma_shape = sigma.shape
        sigma = append_dims(sigma, input.ndim)
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = c_noise.reshape(sigma_shape)
        if self.quantize_c_noise:
            c_noise = self.sigma_to_idx(c_noise)
        return network(input * c_in, c_noise, cond, **additional_model_inputs) * c_out + input * c_skip


class VanillaCFG:
    """guiders.py:24-65.  The concatenated conditioning is built once per (c, uc) pair and reused every step
    (the reference re-concatenates identical tensors each step), which keeps the cross-attention K/V cache hot."""

    def __init__(self, scale: float):
        self.scale = scale
        self._cache = None

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return x_u + self.scale * (x_c - x_u)

    def prepare_inputs(self, x, s, c, uc):
        k = self._cache
        if k is None or k[0] is not c or k[1] is not uc:
            c_out = dict()
            for key in c:
                if key in ["vector", "crossattn", "concat"]:
                    c_out[key] = torch.cat((uc[key], c[key]), 0)
                else:
                    assert c[key] == uc[key]
                    c_out[key] = c[key]
            self._cache = k = (c, uc, c_out)
        return torch.cat([x] * 2), torch.cat([s] * 2), k[2]


class OpenAIWrapper(nn.Module):
    """wrappers.py:24-34."""

    def __init__(self, diffusion_model):
        super().__init__()
        self.diffusion_model = diffusion_model

    def forward(self, x, t, c: dict, **kwargs):
        if "concat" in c:
            x = torch.cat((x, c["concat"]), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)


class Img2ImgDiscretizationWrapper:
    """scripts/demo/discretization.py:11-32 (installed on the sampler by init_sampling, sdxl_image_generator_utils.py:398-403):
    keeps the last max(int(strength * len), 1) sigmas of the descending list, i.e. img2img starts part-way down the schedule."""

    def __init__(self, discretization, strength: float = 1.0):
        assert 0.0 <= strength <= 1.0
        self.discretization = discretization
        self.strength = strength

    def __call__(self, *args, **kwargs):
        sigmas = torch.flip(self.discretization(*args, **kwargs), (0,))
        sigmas = sigmas[: max(int(self.strength * len(sigmas)), 1)]
        return torch.flip(sigmas, (0,))


class EulerEDMSampler:
    """sampling.py:29-219,309-318 with s_churn = 0 (the reference's default): Euler steps on the EDM ODE."""

    def __init__(self, num_steps: int, guider: VanillaCFG, device="cuda", s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
        self.num_steps = num_steps
        self.discretization = LegacyDDPMDiscretization()
        self.guider = guider
        self.device = device
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise

    def denoise(self, x, denoiser, sigma, cond, uc):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma)

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, num_steps=None):
        n = self.num_steps if num_steps is None else num_steps
        sigmas = self.discretization(n, device=self.device)
        # CPU twin of the schedule for everything that steers the loop: a 0-dim device tensor in a Python comparison is a
        # stream synchronisation per step (cremage_amd.samplers._host_sigmas)
        sh = self.discretization(n, device="cpu").float()
        uc = cond if uc is None else uc
        x = x * float(torch.sqrt(1.0 + sh[0] ** 2.0))
        num_sigmas = len(sigmas)
        s_in = x.new_ones([x.shape[0]])
        for i in range(num_sigmas - 1):
            gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sh[i].item() <= self.s_tmax else 0.0
            sigma_hat_h = sh[i] * (gamma + 1.0)
            sigma_hat = s_in * (sigmas[i] * (gamma + 1.0))
            if gamma > 0:
                eps = torch.randn_like(x) * self.s_noise
                x = x + eps * float((sigma_hat_h ** 2 - sh[i] ** 2) ** 0.5)
            denoised = self.denoise(x, denoiser, sigma_hat, cond, uc)
            d = (x - denoised) / sigma_hat_h.item()
            x = x + (sh[i + 1] - sigma_hat_h).item() * d
        return x


class DiffusionEngine(nn.Module):
    """Minimal stand-in for sgm/models/diffusion.py:19-151: model wrapper + denoiser + first stage + scale factor."""

    def __init__(self, unet: nn.Module, first_stage_model: nn.Module, scale_factor: float = 0.13025):
        super().__init__()
        self.model = OpenAIWrapper(unet)
        self.denoiser = DiscreteDenoiser()
        self.first_stage_model = first_stage_model
        self.scale_factor = scale_factor

    @torch.no_grad()
    def decode_first_stage(self, z):
        return self.first_stage_model.decode(1.0 / self.scale_factor * z)

    @torch.no_grad()
    def encode_first_stage(self, x, noise=None):
        """sgm/models/diffusion.py:139-151: scale_factor * posterior.sample() (the SDXL first stage is
        AutoencoderKLInferenceWrapper, whose encode samples the posterior); `noise` makes the sample explicit."""
        return self.scale_factor * self.first_stage_model.encode(x).sample(noise)

    @torch.no_grad()
    def img2img(self, img, cond: Dict, uc: Dict, steps: int, strength: float, cfg_scale: float, enc_noise=None, fwd_noise=None):
        """do_img2img, sdxl_image_generator_utils.py:989-1016 (the face-fix re-entry of BASELINE config 5 is this call on a crop,
        strength 0.3): encode, noise to sigma_0 of the pruned schedule, Euler-EDM over the remaining sigmas."""
        z = self.encode_first_stage(img, enc_noise)
        smp = EulerEDMSampler(steps, VanillaCFG(cfg_scale), device=z.device)
        smp.discretization = Img2ImgDiscretizationWrapper(smp.discretization, strength=strength)
        sigmas = smp.discretization(steps, device=z.device)
        noise = torch.randn_like(z) if fwd_noise is None else fwd_noise
        noised_z = (z + noise * sigmas[0]) / torch.sqrt(1.0 + sigmas[0] ** 2.0)
        return smp(lambda inp, sigma, c: self.denoiser(self.model, inp, sigma, c), noised_z, cond=cond, uc=uc)

    @torch.no_grad()
    def sample(self, x, cond: Dict, uc: Dict, steps: int, cfg_scale: float):
        """do_sample, sdxl_image_generator_utils.py:695-707: sampler(denoiser, randn, cond=c, uc=uc)."""
        smp = EulerEDMSampler(steps, VanillaCFG(cfg_scale), device=x.device)
        return smp(lambda inp, sigma, c: self.denoiser(self.model, inp, sigma, c), x, cond=cond, uc=uc)
