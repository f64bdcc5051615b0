"""Leaf modules: parameter containers with the reference's names/shapes whose forward runs HIP kernels.

Mirrors modules/ldm/modules/diffusionmodules/util.py: `GroupNorm32` (:214-216), `normalization`
(:199-205), `conv_nd` (:218-228), `linear` (:231-235), `timestep_embedding` (:151-171),
`zero_module` (:174-180).  Parameters stay ordinary `nn.Parameter`s owned by torch so that
`load_state_dict`, LoRA `setattr` and `.half()/.to()` keep working.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


def compute_dtype_of(x: torch.Tensor) -> torch.dtype:
    """Activation dtype policy: bf16 stays bf16 (1-pass MFMA); everything else runs the fp32-class path."""
    return ops.HALF if x.dtype == ops.HALF else torch.float32


class GroupNorm32(nn.GroupNorm):
    """util.py:214-216 - statistics in fp32 whatever the storage dtype; optional fused SiLU."""

    def forward(self, x, silu: bool = False, x2=None, split: bool = False):
        return ops.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, silu=silu, x2=x2, split=split)


class GroupNorm(GroupNorm32):
    """`Normalize` of attention.py:189-190 / model.py:45-46 (eps 1e-6)."""


def normalization(channels: int) -> GroupNorm32:
    return GroupNorm32(32, channels)  # eps 1e-5 default, util.py:205


def Normalize(in_channels: int, num_groups: int = 32) -> GroupNorm:
    return GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class SiLU(nn.SiLU):
    """Stand-alone SiLU (only reached for non-fused uses, e.g. on the timestep embedding)."""

    def forward(self, x):
        return ops.silu(x)


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameter container; forward = implicit-GEMM MFMA conv (or GEMM for 1x1)."""

    def forward(self, x, **fused):
        if self.groups != 1 or self.dilation != (1, 1) or self.padding_mode != "zeros":
            raise NotImplementedError("cremage_amd Conv2d: groups/dilation/padding_mode unsupported")
        ks = self.kernel_size[0]
        pad = self.padding[0] if isinstance(self.padding, tuple) else self.padding
        if ks == 1 and self.stride == (1, 1) and pad == 0 and self.in_channels > 8 and self.out_channels > 8 \
                and not {"x2", "cvec", "upsample2x"} & set(k for k, v in fused.items() if v is not None and v is not False):
            return ops.conv1x1(x, self.weight, self.bias, residual=fused.get("residual"), gn_stats=bool(fused.get("gn_stats")))
        if "padding" in fused:
            pad = fused.pop("padding")
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride[0], padding=pad, **fused)


class Linear(nn.Linear):
    def forward(self, x, residual=None, act=None, out_dtype=None):
        return ops.linear(x, self.weight, self.bias, residual=residual, act=act, out_dtype=out_dtype)


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims}")  # util.py:228; only dims=2 is on the SD path
    return Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return Linear(*args, **kwargs)


def zero_module(module: nn.Module) -> nn.Module:
    for p in module.parameters():
        p.detach().zero_()
    return module


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False, dtype=torch.float32):
    """util.py:151-171 (max_period fixed at 10000 in the kernel, as every call site uses)."""
    if repeat_only or max_period != 10000:
        raise NotImplementedError("timestep_embedding: only the sinusoidal max_period=10000 form is on the SD path")
    return ops.timestep_embedding(timesteps, dim, dtype)
