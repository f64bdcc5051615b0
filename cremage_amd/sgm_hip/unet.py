"""SDXL UNet on HIP kernels - drop-in for modules/sdxl/sgm/modules/diffusionmodules/openaimodel.py
`UNetModel` (:476-874; ctor :506-826, forward :828-874), configured by
modules/sdxl/configs/inference/sd_xl_base.yaml:17-33 (channel_mult [1,2,4], transformer_depth [1,2,10],
num_head_channels 64, context_dim 2048, use_linear_in_transformer, num_classes "sequential" with
adm_in_channels 2816 -> `label_emb`).  Same module tree / parameter names as the reference (2 567.46 M
parameters); reuses the SD1.5 ResBlock / Downsample / Upsample / TimestepEmbedSequential and the kernels.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import ops
from ..ldm_hip.nn import SiLU, conv_nd, linear, normalization, timestep_embedding, zero_module
from ..ldm_hip.unet import Downsample, ResBlock, TimestepBlock, Upsample
from ..ldm_hip.unet import UNetModel as _SD15UNet
from .transformer import SpatialTransformer, T


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """sgm openaimodel.py TimestepEmbedSequential: same dispatch rule as the SD1.5 one."""

    def forward(self, x, emb, context=None, **kw):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, T.SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    def __init__(self, in_channels: int, model_channels: int, out_channels: int, num_res_blocks: int, attention_resolutions: int,
                 dropout: float = 0.0, channel_mult: Union[List, Tuple] = (1, 2, 4, 8), conv_resample: bool = True, dims: int = 2,
                 num_classes: Optional[Union[int, str]] = None, use_checkpoint: bool = False, num_heads: int = -1,
                 num_head_channels: int = -1, num_heads_upsample: int = -1, use_scale_shift_norm: bool = False,
                 resblock_updown: bool = False, transformer_depth: int = 1, context_dim: Optional[int] = None,
                 disable_self_attentions: Optional[List[bool]] = None, num_attention_blocks: Optional[List[int]] = None,
                 disable_middle_self_attn: bool = False, disable_middle_transformer: bool = False,
                 use_linear_in_transformer: bool = False, spatial_transformer_attn_type: str = "softmax",
                 adm_in_channels: Optional[int] = None, lora_ranks: List[int] = None, lora_weights: List[float] = None):
        super().__init__()
        if not conv_resample or resblock_updown or dims != 2 or use_scale_shift_norm or disable_middle_transformer:
            raise NotImplementedError("sgm UNetModel: unsupported structural option for the SDXL path")
        if num_classes not in (None, "sequential"):
            raise NotImplementedError("sgm UNetModel: only num_classes None / 'sequential' (SDXL) are restated")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth = list(transformer_depth)
        transformer_depth_middle = transformer_depth[-1]
        if isinstance(num_res_blocks, int):
            self.num_res_blocks = len(channel_mult) * [num_res_blocks]
        else:
            if len(num_res_blocks) != len(channel_mult):
                raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                                 "as a list/tuple (per-level) with the same length as channel_mult")
            self.num_res_blocks = list(num_res_blocks)
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.compute_dtype: Optional[torch.dtype] = None
        self._ctx_cast = None
        self._resblocks = None

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def st(ch, depth, disabled_sa):
            nh, dh = heads_for(ch)
            return SpatialTransformer(ch, nh, dh, depth=depth, context_dim=context_dim, disable_self_attn=disabled_sa,
                                      use_linear=use_linear_in_transformer, attn_type=spatial_transformer_attn_type,
                                      use_checkpoint=use_checkpoint, lora_ranks=lora_ranks, lora_weights=lora_weights)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim))
        if self.num_classes == "sequential":
            assert adm_in_channels is not None
            self.label_emb = nn.Sequential(nn.Sequential(linear(adm_in_channels, time_embed_dim), SiLU(),
                                                         linear(time_embed_dim, time_embed_dim)))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        input_block_chans = [model_channels]
        ch = model_channels
        ds = 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims, use_checkpoint=use_checkpoint)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    disabled_sa = disable_self_attentions[level] if (context_dim is not None and T.exists(disable_self_attentions)) else False
                    if not T.exists(num_attention_blocks) or nr < num_attention_blocks[level]:
                        layers.append(st(ch, transformer_depth[level], disabled_sa))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, out_channels=ch, dims=dims, use_checkpoint=use_checkpoint),
            st(ch, transformer_depth_middle, disable_middle_self_attn),
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, time_embed_dim, dropout, out_channels=model_channels * mult, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    disabled_sa = disable_self_attentions[level] if T.exists(disable_self_attentions) else False
                    if not T.exists(num_attention_blocks) or i < num_attention_blocks[level]:
                        layers.append(st(ch, transformer_depth[level], disabled_sa))
                if level and i == self.num_res_blocks[level]:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))

    _emb_projections = _SD15UNet._emb_projections
    _emb_rows = _SD15UNet._emb_rows
    _emb_table = _SD15UNet._emb_table
    resolve_compute_dtype = _SD15UNet.resolve_compute_dtype

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """sgm openaimodel.py:828-874: emb = time_embed(t_emb) [+ label_emb(y)]; blocks as in SD1.5."""
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        cdt = self.resolve_compute_dtype()
        t_emb = timestep_embedding(timesteps, self.model_channels, dtype=cdt)
        emb = self.time_embed[2](self.time_embed[0](t_emb, act="silu"))
        if self.num_classes is not None:
            assert y.shape[0] == x.shape[0]
            le = self.label_emb[0]
            yv = y if y.dtype == cdt else y.to(cdt)
            emb = le[2](le[0](yv, act="silu"), residual=emb)  # emb + label_emb(y)
        emb._crg_emb_out = self._emb_projections(emb)
        if context is not None and context.dtype != cdt:
            c = self._ctx_cast
            if c is None or c[0] is not context or c[1] != context._version or c[2].dtype != cdt:
                self._ctx_cast = c = (context, context._version, context.to(cdt))
            context = c[2]
        hs = []
        h = ops.nchw_to_nhwc(x, cdt)
        for module in self.input_blocks:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for module in self.output_blocks:
            h = module((h, hs.pop()), emb, context)
        h = self.out[2](self.out[0](h, silu=True))
        return ops.nhwc_to_nchw(h, x.dtype if x.dtype in (torch.float32, ops.HALF) else torch.float32).to(x.dtype)
