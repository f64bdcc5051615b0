"""Sampler loop, CFG wrapper and schedules - these STAY ON PYTORCH by design (BASELINE.json
north_star: "The k_diffusion sampler loop ... stay on PyTorch-ROCm so Cremage's generation-info /
LoRA / ControlNet hooks still attach").  All tensors here are [b, 4, L, L] fp32 latents and scalars;
the per-step cost is the UNet call behind `eps_model`.

Mirrors (behaviour, names and argument meaning):
  DiscreteSchedule / DiscreteEpsDDPMDenoiser / CompVisDenoiser   modules/k_diffusion/external.py:41-147
  LDMWrapperForKDiffusion (CFG batch doubling)                    modules/ldm/models/diffusion/ldm_wrapper_for_k_diffusion.py:20-106
  sample_euler / sample_euler_ancestral / get_ancestral_step      modules/k_diffusion/sampling.py:51-58,118-163
  KDiffusionSamplerBase / EulerSampler / EulerAncestralSampler    modules/ldm/models/diffusion/k_diffusion_samplers.py:63-319
  DDIMSampler.make_schedule / stochastic_encode / decode          modules/ldm/models/diffusion/ddim.py:38-75,615-676
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch
from torch import nn

# dev knob: 0 = the fused sampler steps compute the CompVis wrapper's scalings and timestep per step (round 2) instead of once per run
STEP_TABLES = __import__("os").environ.get("CRG_SAMPLER_TABLES", "1") != "0"


def append_zero(x):
    """x followed by one 0 (the terminal sigma of every k-diffusion schedule)."""
    out = x.new_zeros(x.shape[0] + 1)
    out[:-1] = x
    return out


def append_dims(x, target_dims):
    """Per-sample vector -> broadcastable against a tensor of `target_dims` dims (trailing singleton axes)."""
    if x.ndim > target_dims:
        raise ValueError(f"cannot view a {x.ndim}-d tensor as {target_dims}-d")
    return x.reshape(tuple(x.shape) + (1,) * (target_dims - x.ndim))


def make_beta_schedule(schedule="linear", n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """ldm/modules/diffusionmodules/util.py:21-43 ('linear' is what SD uses)."""
    if schedule != "linear":
        raise ValueError(f"schedule '{schedule}' unknown.")
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def make_alphas_cumprod(n_timestep=1000, linear_start=0.00085, linear_end=0.012) -> torch.Tensor:
    """DDPM.register_schedule ddpm.py:134-186: np.cumprod(1 - betas) stored as an fp32 buffer."""
    betas = make_beta_schedule("linear", n_timestep, linear_start, linear_end)
    return torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)


class DiscreteSchedule(nn.Module):
    """The sigma table of a discrete-time model and the two maps between sigma and (fractional) timestep.

    Behaviour of k_diffusion `DiscreteSchedule` (external.py:41-84), pinned by tests/golden/schedules.npz; the code is this
    repo's own: one ascending fp32 table `log_sigmas` (a buffer, so `.to(device)` moves it) and linear interpolation in
    log-sigma between neighbouring integer timesteps."""

    def __init__(self, sigmas, quantize):
        super().__init__()
        table = sigmas.detach().clone()
        self.register_buffer("sigmas", table)
        self.register_buffer("log_sigmas", table.log())
        self.quantize = quantize

    sigma_min = property(lambda self: self.sigmas[0])
    sigma_max = property(lambda self: self.sigmas[-1])

    def _blend(self, idx_lo, idx_hi, frac):
        """(1 - frac) * log_sigmas[idx_lo] + frac * log_sigmas[idx_hi], the one interpolation rule both maps share."""
        lo, hi = self.log_sigmas[idx_lo], self.log_sigmas[idx_hi]
        return (1 - frac) * lo + frac * hi

    def t_to_sigma(self, t):
        t = t.float()
        below = t.floor()
        return self._blend(below.long(), t.ceil().long(), t - below).exp()

    def get_sigmas(self, n=None):
        """n sigmas from the table's largest to its smallest timestep (evenly spaced in t), then 0; the whole table reversed
        when n is None."""
        if n is None:
            return append_zero(torch.flip(self.sigmas, dims=(0,)))
        last = self.sigmas.shape[0] - 1
        return append_zero(self.t_to_sigma(torch.linspace(last, 0, n, device=self.sigmas.device)))

    def sigma_to_t(self, sigma, quantize=None):
        if quantize is None:
            quantize = self.quantize
        ls = sigma.log()
        if quantize:  # nearest table entry in log-sigma
            return (ls.reshape(1, -1) - self.log_sigmas.reshape(-1, 1)).abs().argmin(dim=0).reshape(sigma.shape)
        # the table is ascending: the bracketing pair is found by binary search (index of the last entry <= log sigma, clamped
        # so that there is always an upper neighbour); out-of-table sigmas clamp to the end timesteps
        n = self.log_sigmas.shape[0]
        lo = (torch.searchsorted(self.log_sigmas, ls.contiguous(), right=True) - 1).clamp(min=0, max=n - 2)
        a, b = self.log_sigmas[lo], self.log_sigmas[lo + 1]
        frac = ((a - ls) / (a - b)).clamp(0, 1)
        return ((1 - frac) * lo + frac * (lo + 1)).reshape(sigma.shape)


class CompVisDenoiser(DiscreteSchedule):
    """eps-prediction wrapper of a CompVis LatentDiffusion (k_diffusion external.py:87-147): sigma_t = sqrt((1 - a_t) / a_t),
    eps = model.apply_model(x / sqrt(sigma^2 + 1), sigma_to_t(sigma), ...), denoised = x - sigma * eps."""

    def __init__(self, model, quantize=False, device='cpu'):
        acp = model.alphas_cumprod
        super().__init__(((1 - acp) / acp) ** 0.5, quantize)
        self.inner_model = model
        self.sigma_data = 1.

    def get_scalings(self, sigma):
        """(c_out, c_in) of the eps parameterisation."""
        return -sigma, 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5

    def get_eps(self, *args, **kwargs):
        return self.inner_model.apply_model(*args, **kwargs)

    def forward(self, input, sigma, **kwargs):
        c_out, c_in = [append_dims(x, input.ndim) for x in self.get_scalings(sigma)]
        x_in = input * c_in
        if getattr(input, "_crg_cfg_dup", False):  # a batch-doubled input stays one under a per-sample scaling of equal sigmas
            x_in._crg_cfg_dup = True
        eps = self.get_eps(x_in, self.sigma_to_t(sigma), **kwargs)
        return input + eps * c_out


class LDMWrapperForKDiffusion(nn.Module):
    """ldm_wrapper_for_k_diffusion.py:20-106: classifier-free guidance by batch doubling, applied to the
    DENOISED outputs of the CompVis wrapper.  The concatenated conditioning is built once and reused
    for every step (the reference rebuilds the same `torch.cat` each step, :67-92), which lets the
    cross-attention K/V cache of cremage_amd.ldm_hip.transformer hit."""

    def __init__(self, compviz_wrapper_model, c, unconditional_conditioning, unconditional_guidance_scale: float):
        super().__init__()
        self.compviz_model = compviz_wrapper_model
        self.alphas_cumprod = compviz_wrapper_model.inner_model.alphas_cumprod
        self.ddpm_num_timesteps = compviz_wrapper_model.inner_model.num_timesteps
        self.c = c
        self.unconditional_conditioning = unconditional_conditioning
        self.unconditional_guidance_scale = unconditional_guidance_scale
        self._c_in = None

    _cat_memo = None  # (c, uc, c_in): one concatenated conditioning per (c, uc) pair across sampler instances, so that the
    #                   modules' K/V cache and a captured hipGraph (keyed on the tensor identity) survive from batch to batch

    def _cat_cond(self):
        m = LDMWrapperForKDiffusion._cat_memo
        if self._c_in is None and m is not None and m[0] is self.c and m[1] is self.unconditional_conditioning \
                and torch.is_tensor(self.c) and m[3] == (self.c._version, self.unconditional_conditioning._version):
            self._c_in = m[2]
        if self._c_in is None:
            c, uc = self.c, self.unconditional_conditioning
            if isinstance(c, dict):
                assert isinstance(uc, dict)
                c_in = dict()
                for k in c:
                    if isinstance(c[k], list):
                        c_in[k] = [torch.cat([uc[k][i], c[k][i]]) for i in range(len(c[k]))]
                    else:
                        c_in[k] = torch.cat([uc[k], c[k]])
            else:
                c_in = {"c_crossattn": [torch.cat([uc, c])]}
                LDMWrapperForKDiffusion._cat_memo = (c, uc, c_in, (c._version, uc._version))
            self._c_in = c_in
        return self._c_in

    def eps_pair(self, x, sigma):
        """The raw eps of the batch-doubled UNet call ([2b, ...], unconditional half first) for the fused sampler step:
        the part of apply_model / CompVisDenoiser.forward before the scalings that crg_cfg_euler_step folds in."""
        cv = self.compviz_model
        x_in = torch.cat([x] * 2)
        sigma_in = torch.cat([sigma] * 2)
        _, c_in = [append_dims(v, x_in.ndim) for v in cv.get_scalings(sigma_in)]
        return cv.get_eps(_mark_dup(x_in * c_in), cv.sigma_to_t(sigma_in), cond=self._cat_cond())

    def eps_tables(self, sigmas):
        """(c_in, t) of the CompVis wrapper for a whole vector of sigmas at once - the same elementwise arithmetic
        CompVisDenoiser.forward (external.py:111-114) performs per step on the batch-expanded sigma, so the values are identical;
        computed once per sampling run they replace ~25 tiny launches per step (scalings, log / searchsorted / blend of sigma_to_t)."""
        cv = self.compviz_model
        sigmas = sigmas.to(cv.log_sigmas.device)  # a CPU schedule worked with the per-step scalars (0-dim broadcast); keep it working
        _, c_in = cv.get_scalings(sigmas)
        return c_in, cv.sigma_to_t(sigmas)

    def time_rows(self, t_rep):
        """The wrapped UNet's timestep-only work for the whole table of timesteps `t_rep` [S, N] (`UNetModel.time_rows`) as (unet, rows),
        or None when the network does not offer it; step i attaches rows[i] to its timesteps (`ops.attach_time_rows`)."""
        dm = getattr(getattr(self.compviz_model.inner_model, "model", None), "diffusion_model", None)
        f = getattr(dm, "time_rows", None)
        from . import ops
        return (dm, f(t_rep)) if (f is not None and t_rep.is_cuda and ops.TIME_ROWS) else None

    def eps_pair_pre(self, x, c_in_i, t_row):
        """eps_pair with the step's scalars taken from eps_tables: `c_in_i` a 0-dim device tensor, `t_row` the step's timestep
        already expanded to the doubled batch.  ONE elementwise launch builds cat([x] * 2) * c_in."""
        xx = torch.empty((2,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        torch.mul(x.unsqueeze(0).expand_as(xx), c_in_i, out=xx)
        return self.compviz_model.get_eps(_mark_dup(xx.view((2 * x.shape[0],) + tuple(x.shape[1:]))), t_row, cond=self._cat_cond())

    def fused_step_ok(self, x) -> bool:
        return (self.unconditional_conditioning is not None and self.unconditional_guidance_scale != 1. and x.is_cuda
                and x.dtype == torch.float32 and isinstance(self.compviz_model, CompVisDenoiser))

    def apply_model(self, x, t, **kwargs):
        uc, scale = self.unconditional_conditioning, self.unconditional_guidance_scale
        if uc is None or scale == 1.:
            return self.compviz_model(x, t, self.c)
        x_in = _mark_dup(torch.cat([x] * 2))
        t_in = torch.cat([t] * 2)
        e_t_uncond, e_t = self.compviz_model(x_in, t_in, cond=self._cat_cond()).chunk(2)
        return e_t_uncond + scale * (e_t - e_t_uncond)

    def forward(self, *args, **kwargs):
        return self.apply_model(*args, **kwargs)


def _mark_dup(x_in):
    """The UNet input built by batch doubling: both halves hold the same latents and timesteps (only the conditioning differs) -
    said to the HIP UNet through a tensor attribute, which survives `apply_model` / `DiffusionWrapper.forward` of the reference's
    container as well (they pass x on untouched, ddpm.py:1034, :1517-1519)."""
    if x_in.is_cuda:
        from . import ops
        ops.mark_cfg_dup(x_in)
    return x_in


def to_d(x, sigma, denoised):
    return (x - denoised) / (append_dims(sigma, x.ndim) if torch.is_tensor(sigma) else sigma)


def get_ancestral_step(sigma_from, sigma_to, eta=1.):
    if not eta:
        return sigma_to, 0.
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


def _host_sigmas(sigmas, sigmas_host):
    """The schedule as a CPU fp32 tensor for everything that steers the loop (comparisons, the ancestral split, the scalar
    step sizes).  A 0-dim DEVICE tensor in a Python `if` / `min` is a device-to-host copy, i.e. a stream synchronisation per
    step: the host then never runs ahead of the GPU and every step starts with the GPU idle for the host's launch latency
    (measured: 40 syncs and ~15 ms of idle GPU per 300 ms batch).  The samplers pass the CPU copy they computed the schedule
    from; without one the device tensor is copied once, before the loop."""
    return sigmas.detach().float().cpu() if sigmas_host is None else sigmas_host


def _with_time_rows(model, t_rep):
    """The per-step rows of the timestep table [S, N] as a list, each carrying the UNet's hoisted timestep work when the wrapper
    offers it (LDMWrapperForKDiffusion.time_rows)."""
    rows = [t_rep[i] for i in range(t_rep.shape[0])]
    tr = model.time_rows(t_rep) if hasattr(model, "time_rows") else None
    if tr is not None:
        from . import ops
        for i, r in enumerate(rows):
            ops.attach_time_rows(r, tr[1][i], tr[0])
    return rows


@torch.no_grad()
def sample_euler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0., s_tmax=float('inf'),
                 s_noise=1., sigmas_host=None):
    """sampling.py:118-143 (Algorithm 2 of Karras et al. 2022), including the per-step randn_like draw
    that keeps the global RNG stream aligned with the reference (:128)."""
    extra_args = {} if extra_args is None else extra_args
    s_in = x.new_ones([x.shape[0]])
    sh = _host_sigmas(sigmas, sigmas_host)
    fused = callback is None and not extra_args and getattr(model, "fused_step_ok", lambda _x: False)(x)
    tables = None
    if fused:
        x = x.clone().contiguous()  # updated in place by the fused step
        if STEP_TABLES and s_churn == 0. and len(sigmas) > 1:  # sigma_hat == sigma on every step: the wrapper's per-step scalars from one vectorised pass
            c_in_all, t_all = model.eps_tables(sigmas[:-1])
            t_rep = t_all.reshape(-1, 1).expand(-1, 2 * x.shape[0]).contiguous()
            tables = (c_in_all, _with_time_rows(model, t_rep))
    for i in range(len(sigmas) - 1):
        gamma = min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sh[i].item() <= s_tmax else 0.
        eps = torch.randn_like(x) * s_noise
        sigma_hat = sh[i] * (gamma + 1)
        if gamma > 0:
            x = x + eps * ((sigma_hat ** 2 - sh[i] ** 2) ** 0.5).item()
        if fused:  # scalings + guidance + Euler update as one kernel (include/crg_hip.h: crg_cfg_euler_step)
            from . import ops
            e2 = model.eps_pair_pre(x, tables[0][i], tables[1][i]) if tables is not None else model.eps_pair(x, (sigmas[i] * (gamma + 1)) * s_in)
            ops.cfg_euler_step_(x, e2.contiguous(), None, sigma_hat.item(), (sh[i + 1] - sigma_hat).item(), model.unconditional_guidance_scale)
            continue
        denoised = model(x, (sigmas[i] * (gamma + 1)) * s_in, **extra_args)
        d = to_d(x, sigma_hat.item(), denoised)
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        dt = (sh[i + 1] - sigma_hat).item()
        x = x + d * dt
    return x


@torch.no_grad()
def sample_euler_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None,
                           sigmas_host=None):
    """sampling.py:147-163."""
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
    s_in = x.new_ones([x.shape[0]])
    sh = _host_sigmas(sigmas, sigmas_host)
    fused = callback is None and not extra_args and getattr(model, "fused_step_ok", lambda _x: False)(x)
    if fused:
        x = x.clone().contiguous()  # updated in place by the fused step
        tables = STEP_TABLES and len(sigmas) > 1
        if tables:
            c_in_all, t_all = model.eps_tables(sigmas[:-1])
            t_rep = _with_time_rows(model, t_all.reshape(-1, 1).expand(-1, 2 * x.shape[0]).contiguous())
    for i in range(len(sigmas) - 1):
        if fused:  # scalings + guidance + Euler update + ancestral noise as one kernel (crg_cfg_euler_step)
            from . import ops
            e2 = model.eps_pair_pre(x, c_in_all[i], t_rep[i]) if tables else model.eps_pair(x, sigmas[i] * s_in)
            sigma_down, sigma_up = get_ancestral_step(sh[i], sh[i + 1], eta=eta)
            noise = noise_sampler(sigmas[i], sigmas[i + 1]).contiguous() if sh[i + 1].item() > 0 else None
            ops.cfg_euler_step_(x, e2.contiguous(), noise, sh[i].item(), (sigma_down - sh[i]).item(), model.unconditional_guidance_scale,
                                s_noise * float(sigma_up))
            continue
        denoised = model(x, sigmas[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sh[i], sh[i + 1], eta=eta)   # CPU fp32 scalars: no device round trip
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        d = to_d(x, sh[i].item(), denoised)
        dt = (sigma_down - sh[i]).item()
        x = x + d * dt
        if sh[i + 1].item() > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * float(sigma_up)
    return x


class KDiffusionSamplerBase(object):
    """k_diffusion_samplers.py:63-296.  `model` is a LatentDiffusion-like object exposing
    `apply_model(x, t, cond)`, `alphas_cumprod`, `num_timesteps` and `device`."""

    def __init__(self, model, sigma_min=0.0316386, sigma_max=14.5521805, beta_d=19.9, beta_min=0.1, eps_s=1e-3):
        self.ldm_model = model
        self.ddpm_num_timesteps = model.num_timesteps
        assert model.alphas_cumprod.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        self.beta_d, self.beta_min, self.eps_s = beta_d, beta_min, eps_s
        self.device = model.device
        self.alphas_cumprod = model.alphas_cumprod.clone().detach().to(torch.float32).to(self.device)
        self.sqrt_alphas_cumprod = self.alphas_cumprod.sqrt()
        self.sqrt_one_minus_alphas_cumprod = (1. - self.alphas_cumprod).sqrt()
        self.noise_sampler = None
        self.callback = None

    def compute_sigmas(self, n: int):
        return None

    def compute_sigmas_host(self, n: int):
        """The same schedule computed on the CPU from a cached CPU copy of alphas_cumprod (fp32, the arithmetic of the
        reference run on a CPU device): steers the sampling loop without touching the device (see _host_sigmas)."""
        m = self.ldm_model
        acp = m.__dict__.get("_crg_acp_cpu")
        if acp is None or acp[0] != m.alphas_cumprod._version:
            acp = m.__dict__["_crg_acp_cpu"] = (m.alphas_cumprod._version, m.alphas_cumprod.detach().float().cpu())
        sched = DiscreteSchedule(((1 - acp[1]) / acp[1]) ** 0.5, False)
        return sched.get_sigmas(n)

    @torch.no_grad()
    def _sample_common_prep(self, S, batch_size, shape, conditioning=None, x0=None, unconditional_guidance_scale=1.,
                            unconditional_conditioning=None, **kwargs):
        C, H, W = shape
        size = (batch_size, C, H, W)
        # the reference ignores x_T for k-diffusion samplers and draws randn unless x0 is given (:165-171)
        self.x = torch.randn(size, device=self.device) if x0 is None else x0
        self.compviz_wrapper_model = CompVisDenoiser(self.ldm_model, False).to(self.device)
        self.ldm_wrapper_model = LDMWrapperForKDiffusion(self.compviz_wrapper_model, conditioning, unconditional_conditioning,
                                                         unconditional_guidance_scale)
        self.sigmas = self.compute_sigmas(S)
        self.sigmas_host = self.compute_sigmas_host(S)
        if "denoising_steps" in kwargs:  # partial denoising (img2img), :188-194
            t = kwargs["denoising_steps"]
            self.sigmas = self.sigmas[-(t + 1):]
            self.sigmas_host = self.sigmas_host[-(t + 1):] if self.sigmas_host is not None else None
            assert self.sigmas.shape[0] == t + 1

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, x0=None, x_T=None, eta=0., verbose=True,
               unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        self.callback = None  # the reference accepts `callback` but never forwards it to k-diffusion (:307)
        self._sample_common_prep(S=S, batch_size=batch_size, shape=shape, conditioning=conditioning, x0=x0,
                                 unconditional_guidance_scale=unconditional_guidance_scale,
                                 unconditional_conditioning=unconditional_conditioning, **kwargs)
        return self.do_sample()

    @torch.no_grad()
    def do_sample(self):
        return self.x, None

    @torch.no_grad()
    def stochastic_encode(self, x0, t, sampling_steps, noise=None):
        """k_diffusion_samplers.py:255-296: forward-diffuse x0 to DDPM step t*1000/sampling_steps."""
        if noise is None:
            noise = torch.randn_like(x0)
        t = (t * 1000.0 / sampling_steps).long()
        ex = lambda a: a.gather(-1, t).reshape(t.shape[0], *((1,) * (x0.ndim - 1)))
        return ex(self.sqrt_alphas_cumprod) * x0 + ex(self.sqrt_one_minus_alphas_cumprod) * noise


class EulerSampler(KDiffusionSamplerBase):
    @torch.no_grad()
    def compute_sigmas(self, n):
        return self.compviz_wrapper_model.get_sigmas(n).to(self.device)

    @torch.no_grad()
    def do_sample(self):
        return sample_euler(self.ldm_wrapper_model, self.x, self.sigmas, sigmas_host=self.sigmas_host), None


class EulerAncestralSampler(KDiffusionSamplerBase):
    @torch.no_grad()
    def compute_sigmas(self, n):
        return self.compviz_wrapper_model.get_sigmas(n).to(self.device)

    @torch.no_grad()
    def do_sample(self):
        return sample_euler_ancestral(self.ldm_wrapper_model, self.x, self.sigmas, noise_sampler=self.noise_sampler,
                                      sigmas_host=self.sigmas_host), None


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=False):
    """util.py:46-60 ('uniform')."""
    if ddim_discr_method != 'uniform':
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


class DDIMSampler(object):
    """ddim.py (eta = 0 as the img2img driver uses it): make_schedule :38-75, stochastic_encode :615-654,
    decode :657-676, p_sample_ddim :530-612."""

    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=False):
        assert ddim_eta == 0., "only the deterministic (eta = 0) DDIM of the img2img driver is restated"
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps)
        acp = self.model.alphas_cumprod.detach().float().cpu()
        assert acp.shape[0] == self.ddpm_num_timesteps
        dev = self.model.device
        self.ddim_alphas = acp[self.ddim_timesteps].to(dev)
        self.ddim_alphas_prev = torch.cat([acp[:1], acp[self.ddim_timesteps[:-1]]]).to(dev)
        self.ddim_sqrt_one_minus_alphas = (1. - self.ddim_alphas).sqrt()
        self.ddim_sigmas = torch.zeros_like(self.ddim_alphas)

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        assert not use_original_steps
        if noise is None:
            noise = torch.randn_like(x0)
        ex = lambda a: a.gather(-1, t).reshape(t.shape[0], *((1,) * (x0.ndim - 1)))
        return ex(self.ddim_alphas.sqrt()) * x0 + ex(self.ddim_sqrt_one_minus_alphas) * noise

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, unconditional_guidance_scale=1., unconditional_conditioning=None):
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            e_t = self.model.apply_model(x, t, c)
        else:
            x_in = _mark_dup(torch.cat([x] * 2))
            t_in = torch.cat([t] * 2)
            c_in = self._c_in if getattr(self, "_c_in", None) is not None else torch.cat([unconditional_conditioning, c])
            e_t_uncond, e_t = self.model.apply_model(x_in, t_in, c_in).chunk(2)
            e_t = e_t_uncond + unconditional_guidance_scale * (e_t - e_t_uncond)
        a_t, a_prev = self.ddim_alphas[index], self.ddim_alphas_prev[index]
        pred_x0 = (x - self.ddim_sqrt_one_minus_alphas[index] * e_t) / a_t.sqrt()
        dir_xt = (1. - a_prev).sqrt() * e_t
        return a_prev.sqrt() * pred_x0 + dir_xt, pred_x0

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        assert not use_original_steps
        timesteps = self.ddim_timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        # one concatenated conditioning for the whole decode (K/V cache friendly); same values as ddim.py:553
        self._c_in = None
        if unconditional_conditioning is not None and unconditional_guidance_scale != 1. and torch.is_tensor(cond):
            self._c_in = torch.cat([unconditional_conditioning, cond])
        x_dec = x_latent
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index, unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        self._c_in = None
        return x_dec
