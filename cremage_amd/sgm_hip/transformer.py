"""SDXL (sgm) transformer blocks on the same HIP kernels - drop-in for
modules/sdxl/sgm/modules/attention.py: `CrossAttention` (:358-534, SDPA) / `MemoryEfficientCrossAttention`
(:537-721), `BasicTransformerBlock` (:724-847), `SpatialTransformer` (:902-1133).

Differences from the SD1.5 twins that matter here: `use_linear` IS honoured (proj_in / proj_out are
nn.Linear [C, C], sd_xl_base.yaml:27), `depth` can be 2 or 10, heads = C / 64, the attention class is chosen
by the ctor argument `attn_mode` (<- YAML `spatial_transformer_attn_type`, sd_xl_base.yaml:33), and there is
no IP-Adapter branch.  The arithmetic is the same function, so the classes reuse the ldm_hip implementations;
every registry mode maps to the HIP attention.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..ldm_hip import transformer as T
from ..ldm_hip.nn import Conv2d, Linear, Normalize, zero_module


class CrossAttention(T.CrossAttention):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, backend=None, lora_ranks: List[int] = None,
                 lora_weights: List[float] = None, **kwargs):
        super().__init__(query_dim, context_dim=context_dim, heads=heads, dim_head=dim_head, dropout=dropout, lora_ranks=lora_ranks,
                         lora_weights=lora_weights)
        self.backend = backend

    def forward(self, x, context=None, mask=None, additional_tokens=None, n_times_crossframe_attn_in_self=0, residual=None, ln=None, **fused):
        if additional_tokens is not None or n_times_crossframe_attn_in_self:
            raise NotImplementedError("additional_tokens / cross-frame attention belong to the video models (out of scope)")
        return super().forward(x, context=context, mask=mask, residual=residual, ln=ln, **fused)


MemoryEfficientCrossAttention = CrossAttention


class BasicTransformerBlock(T.BasicTransformerBlock):
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": CrossAttention, "softmax-hip": CrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True, disable_self_attn=False,
                 attn_mode="softmax", sdp_backend=None, lora_ranks: List[int] = None, lora_weights: List[float] = None):
        nn.Module.__init__(self)
        assert attn_mode in self.ATTENTION_MODES
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.disable_self_attn = disable_self_attn
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              context_dim=context_dim if self.disable_self_attn else None, backend=sdp_backend, lora_ranks=lora_ranks,
                              lora_weights=lora_weights)
        self.ff = T.FeedForward(dim, dropout=dropout, glu=gated_ff, lora_ranks=lora_ranks, lora_weights=lora_weights)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              backend=sdp_backend, lora_ranks=lora_ranks, lora_weights=lora_weights)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None, additional_tokens=None, n_times_crossframe_attn_in_self=0):
        if additional_tokens is not None or n_times_crossframe_attn_in_self:
            raise NotImplementedError("additional_tokens / cross-frame attention belong to the video models (out of scope)")
        return self._forward(x, context)


class SpatialTransformer(T.SpatialTransformer):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False,
                 use_linear=False, attn_type="softmax", use_checkpoint=True, sdp_backend=None, lora_ranks: List[int] = None,
                 lora_weights: List[float] = None):
        nn.Module.__init__(self)
        if T.exists(context_dim) and type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)
        if T.exists(context_dim) and not isinstance(context_dim, list):
            context_dim = [context_dim]
        if T.exists(context_dim) and isinstance(context_dim, list):
            if depth != len(context_dim):  # attention.py:938-949
                assert all(map(lambda x: x == context_dim[0], context_dim)), "need homogenous context_dim to match depth automatically"
                context_dim = depth * [context_dim[0]]
        elif context_dim is None:
            context_dim = [None] * depth
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.lora_ranks = lora_ranks if lora_ranks is not None else []
        self.lora_weights = lora_weights if lora_weights is not None else [1.0] * len(self.lora_ranks)
        self.use_linear = use_linear
        if not use_linear:
            self.proj_in = Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        else:
            self.proj_in = Linear(in_channels, inner_dim)
        T._lora_lists(self, "proj_in", in_channels, inner_dim, self.lora_ranks, conv=not use_linear)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, attn_mode=attn_type, checkpoint=use_checkpoint,
                                  sdp_backend=sdp_backend, lora_ranks=self.lora_ranks, lora_weights=self.lora_weights)
            for d in range(depth)])
        if not use_linear:
            self.proj_out = zero_module(Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))
        else:
            self.proj_out = zero_module(Linear(inner_dim, in_channels))
        T._lora_lists(self, "proj_out", inner_dim, in_channels, self.lora_ranks, conv=not use_linear)

    def forward(self, x, context=None):
        if isinstance(context, list):  # attention.py:1070-1071,1103-1106: one context per block, or one shared
            if len(context) != 1:
                raise NotImplementedError("per-block context lists are not used by SDXL base")
            context = context[0]
        return super().forward(x, context)
