"""ControlNet path - drop-in for modules/cldm/cldm.py.

  ControlledUnetModel  cldm.py:28-70    the SD1.5 UNet whose forward adds the control residuals after the middle block and
                                        to every skip connection (or only the former with only_mid_control)
  ControlNet           cldm.py:73-342   the UNet's encoder half + middle block with its own weights, a hint encoder
                                        (8 convs, `input_hint_block` :178-194) and one 1x1 "zero conv" per output (:176,
                                        :316-317, :314)
  ControlLDM           cldm.py:345-393  apply_model glue: control = control_model(x, hint, t, ctx) * control_scales -> UNet

Same constructor signatures, module tree and parameter names (`time_embed`, `input_blocks`, `zero_convs`,
`input_hint_block`, `middle_block`, `middle_block_out`), so a ControlNet checkpoint's `control_model.*` keys load as they do
in the reference.  What differs is how it runs on MI355X:
  * the whole walk is channels-last bf16 on the same MFMA conv / GEMM / flash-attention kernels as the UNet;
  * `guided_hint` depends only on the hint image, not on x or t: it is computed once per hint tensor (identity + version)
    instead of once per sampler step;
  * the residual adds of ControlledUnetModel (`h += control.pop()`, `hs.pop() + control.pop()`, :57-65) are in-place on
    tensors nobody else reads, and the skip concat is never materialised (the UNet's output blocks take the pair).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from ..ldm_hip.latent_diffusion import LatentDiffusion, instantiate_from_config
from ..ldm_hip.nn import SiLU, conv_nd, linear, timestep_embedding, zero_module
from ..ldm_hip.transformer import SpatialTransformer, exists
from ..ldm_hip.unet import Downsample, ResBlock, TimestepEmbedSequential, UNetModel


def _as_act(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """A control residual in the walk's layout/dtype (channels-last, compute dtype).  Tensors produced by the HIP
    ControlNet already are; NCHW fp32 tensors handed in by a caller (the reference's own calling convention) are converted."""
    if t.dtype == like.dtype and t.is_contiguous(memory_format=torch.channels_last):
        return t
    return ops.nchw_to_nhwc(t, like.dtype)


class ControlledUnetModel(UNetModel):
    """cldm.py:28-70.  With control=None it is the plain UNetModel (:57, :60-61)."""

    def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, cfg_dup: bool = False, **kwargs):
        cdt, emb, context = self._prologue(timesteps, context)
        control = list(control) if control is not None else None
        h, hs = self._input_blocks(x, cdt, emb, context, cfg_dup or getattr(x, "_crg_cfg_dup", False))  # the encoder runs no_grad / unchanged (:47-56)
        h = self.middle_block(h, emb, context)
        if control is not None:
            h = h.add_(_as_act(control.pop(), h))                      # Change 1 (:57-58)
        for module in self.output_blocks:
            skip = hs.pop()
            if control is not None and not only_mid_control:
                skip = skip.add_(_as_act(control.pop(), skip))          # Change 2 (:64): the skip has no other reader
            h = module((h, skip), emb, context)
        return self._epilogue(h, x)


class ControlNet(nn.Module):
    """cldm.py:73-342; only the structure cldm_v15.yaml uses is accepted (spatial transformer, conv down-sampling)."""

    def __init__(self, image_size, in_channels, model_channels, hint_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, use_checkpoint=False, use_fp16=False, num_heads=-1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None,
                 legacy=True, disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("ControlNet: the SD path always uses use_spatial_transformer=True with a context_dim")
        if not conv_resample or resblock_updown or dims != 2 or n_embed is not None or use_scale_shift_norm:
            raise NotImplementedError("ControlNet: unsupported structural option for the SD path")
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, 'Either num_heads or num_head_channels has to be set'
        if num_head_channels == -1:
            assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        self.dims = dims
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        if isinstance(num_res_blocks, int):
            self.num_res_blocks = len(channel_mult) * [num_res_blocks]
        else:
            if len(num_res_blocks) != len(channel_mult):
                raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                                 "as a list/tuple (per-level) with the same length as channel_mult")
            self.num_res_blocks = list(num_res_blocks)
        if disable_self_attentions is not None:
            assert len(disable_self_attentions) == len(channel_mult)
        if num_attention_blocks is not None:
            assert len(num_attention_blocks) == len(self.num_res_blocks)
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.compute_dtype: Optional[torch.dtype] = None
        self._ctx_cast = None
        self._resblocks = None
        self._hint_cache = None
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = False

        def st(ch, nh, **kw):
            # cldm.py:214-238: legacy=False => dim_head = ch // num_heads
            if num_head_channels == -1:
                dh = ch // nh
            else:
                nh = ch // num_head_channels
                dh = num_head_channels
            if legacy:
                dh = ch // nh
            return SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim, **kw)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        self.zero_convs = nn.ModuleList([self.make_zero_conv(model_channels)])
        self.input_hint_block = TimestepEmbedSequential(
            conv_nd(dims, hint_channels, 16, 3, padding=1), SiLU(),
            conv_nd(dims, 16, 16, 3, padding=1), SiLU(),
            conv_nd(dims, 16, 32, 3, padding=1, stride=2), SiLU(),
            conv_nd(dims, 32, 32, 3, padding=1), SiLU(),
            conv_nd(dims, 32, 96, 3, padding=1, stride=2), SiLU(),
            conv_nd(dims, 96, 96, 3, padding=1), SiLU(),
            conv_nd(dims, 96, 256, 3, padding=1, stride=2), SiLU(),
            zero_module(conv_nd(dims, 256, model_channels, 3, padding=1)))
        self._feature_size = model_channels
        ch = model_channels
        ds = 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims, use_checkpoint=use_checkpoint)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    disabled_sa = disable_self_attentions[level] if exists(disable_self_attentions) else False
                    if not exists(num_attention_blocks) or nr < num_attention_blocks[level]:
                        layers.append(st(ch, num_heads, disable_self_attn=disabled_sa, use_linear=use_linear_in_transformer,
                                         use_checkpoint=use_checkpoint))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self.zero_convs.append(self.make_zero_conv(ch))
                self._feature_size += ch
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                self.zero_convs.append(self.make_zero_conv(ch))
                ds *= 2
                self._feature_size += ch
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint),
            st(ch, num_heads, disable_self_attn=disable_middle_self_attn, use_linear=use_linear_in_transformer, use_checkpoint=use_checkpoint),
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint))
        self.middle_block_out = self.make_zero_conv(ch)
        self._feature_size += ch

    def make_zero_conv(self, channels):
        return TimestepEmbedSequential(zero_module(conv_nd(self.dims, channels, channels, 1, padding=0)))

    # shared with the UNet: batched timestep-embedding projections, dtype policy, context cast
    _emb_projections = UNetModel._emb_projections
    _emb_rows = UNetModel._emb_rows
    _emb_table = UNetModel._emb_table
    resolve_compute_dtype = UNetModel.resolve_compute_dtype
    _prologue = UNetModel._prologue

    def guided_hint(self, hint: torch.Tensor, cdt: torch.dtype) -> torch.Tensor:
        """input_hint_block(hint) (cldm.py:327): step-invariant, so cached per hint tensor (identity + version + dtype)
        and per state of the block's parameters."""
        key = (hint, hint._version, cdt, tuple(p._version for p in self.input_hint_block.parameters()))
        c = self._hint_cache
        if c is not None and c[0][0] is hint and c[0][1:] == key[1:]:
            return c[1]
        g = self.input_hint_block(ops.nchw_to_nhwc(hint, cdt), None, None)
        self._hint_cache = (key, g)
        return g

    def forward(self, x, hint, timesteps, context, **kwargs) -> List[torch.Tensor]:
        """-> [zero_conv_i(h_i) for every input block] + [middle_block_out(h_mid)] (cldm.py:319-342), channels-last tensors
        in the compute dtype (logical NCHW shapes as in the reference)."""
        cdt, emb, context = self._prologue(timesteps, context)
        g = self.guided_hint(hint, cdt)
        outs = []
        h = ops.nchw_to_nhwc(x, cdt)
        for i, (module, zero_conv) in enumerate(zip(self.input_blocks, self.zero_convs)):
            h = module(h, emb, context)
            if i == 0:
                h = h.add_(g)                  # `h += guided_hint` after the first input block only (:329-333)
            outs.append(zero_conv(h, emb, context))
        h = self.middle_block(h, emb, context)
        outs.append(self.middle_block_out(h, emb, context))
        return outs


class ControlLDM(LatentDiffusion):
    """cldm.py:345-393 over the stand-alone LatentDiffusion container: cond = {"c_crossattn": [c], "c_concat": [hint] | None}."""

    def __init__(self, control_stage_config, control_key="hint", only_mid_control=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.control_model = instantiate_from_config(control_stage_config) if isinstance(control_stage_config, dict) else control_stage_config
        self.control_key = control_key
        self.only_mid_control = only_mid_control
        self.control_scales = [1.0] * 13

    def apply_model(self, x_noisy, t, cond: Dict[str, list], *args, **kwargs):
        assert isinstance(cond, dict)
        diffusion_model = self.model.diffusion_model
        cond_txt = cond["c_crossattn"][0] if len(cond["c_crossattn"]) == 1 else torch.cat(cond["c_crossattn"], 1)
        if cond.get("c_concat") is None:
            return diffusion_model(x=x_noisy, timesteps=t, context=cond_txt, control=None, only_mid_control=self.only_mid_control)
        hint = cond["c_concat"][0] if len(cond["c_concat"]) == 1 else torch.cat(cond["c_concat"], 1)
        control = self.control_model(x=x_noisy, hint=hint, timesteps=t, context=cond_txt)
        control = [c if s == 1.0 else c * s for c, s in zip(control, self.control_scales)]
        return diffusion_model(x=x_noisy, timesteps=t, context=cond_txt, control=control, only_mid_control=self.only_mid_control)
