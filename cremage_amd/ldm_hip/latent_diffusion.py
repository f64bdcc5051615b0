"""Minimal LatentDiffusion container: the glue between samplers and the HIP UNet / VAE.

The reference's container (modules/ldm/models/diffusion/ddpm.py: `DDPM` :60-186, `LatentDiffusion`
:453-1499, `DiffusionWrapper` :1502-1530) stays on PyTorch in a Cremage deployment and calls the
drop-in classes through its YAML `target:` lines.  This stand-alone container restates
only what the samplers and the decode/encode boundary use, so that the path can be driven (tests,
bench, multi-GPU sharding) without the pytorch_lightning / omegaconf stack:

  alphas_cumprod, num_timesteps        register_schedule            ddpm.py:134-186
  apply_model(x, t, cond)              crossattn branch             ddpm.py:926-1039 (:1034), :1517-1519
  decode_first_stage(z)                z / scale_factor -> decode   ddpm.py:741-798 (:748,798)
  encode_first_stage / get_first_stage_encoding                     ddpm.py:861-898, :575-582
  low_vram_shift                       no-op: 288 GB of HBM keeps UNet + VAE resident (ddpm.py:1460-1499)
"""
from __future__ import annotations

import importlib
from typing import Optional

import torch
import torch.nn as nn

from ..samplers import make_alphas_cumprod
from .vae import DiagonalGaussianDistribution


def get_obj_from_str(string: str):
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    """modules/ldm/util.py:81-96: `target` names the class, `params` its kwargs - the reference's plug point."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


class DiffusionWrapper(nn.Module):
    """ddpm.py:1502-1530, conditioning_key 'crossattn' (v1-inference.yaml:19)."""

    def __init__(self, diff_model_config, conditioning_key="crossattn"):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config) if isinstance(diff_model_config, dict) else diff_model_config
        self.conditioning_key = conditioning_key
        assert conditioning_key == "crossattn", "only the SD 'crossattn' conditioning is restated"

        self.graphed = None  # set by enable_hip_graph()

    def enable_hip_graph(self, on: bool = True):
        """Replay the UNet call from a captured hipGraph (cremage_amd.graphs) instead of ~390 eager launches."""
        if on:
            from ..graphs import GraphedModule
            self.graphed = GraphedModule(self.diffusion_model)
        else:
            self.graphed = None

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None):
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        if self.graphed is not None and x.is_cuda:
            return self.graphed(x, timesteps=t, context=cc)
        return self.diffusion_model(x, t, context=cc)


class LatentDiffusion(nn.Module):
    def __init__(self, unet_config, first_stage_config, linear_start=0.00085, linear_end=0.012, timesteps=1000,
                 scale_factor=0.18215, conditioning_key="crossattn", **ignored):
        super().__init__()
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.first_stage_model = instantiate_from_config(first_stage_config) if isinstance(first_stage_config, dict) else first_stage_config
        self.scale_factor = scale_factor
        self.num_timesteps = int(timesteps)
        self.parameterization = "eps"
        self.register_buffer("alphas_cumprod", make_alphas_cumprod(timesteps, linear_start, linear_end))

    @property
    def device(self):
        return self.alphas_cumprod.device

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        if isinstance(cond, dict):
            pass
        else:
            if not isinstance(cond, list):
                cond = [cond]
            cond = {"c_crossattn": cond}
        return self.model(x_noisy, t, **cond)

    @torch.no_grad()
    def decode_first_stage(self, z):
        return self.first_stage_model.decode(1. / self.scale_factor * z)

    @torch.no_grad()
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior, noise: Optional[torch.Tensor] = None):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample(noise)
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    def low_vram_shift(self, is_diffusing, deinit=False):
        return None
