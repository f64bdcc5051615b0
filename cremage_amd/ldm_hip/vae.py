"""SD VAE (AutoencoderKL: Encoder / Decoder) on HIP kernels.

Drop-in for modules/ldm/modules/diffusionmodules/model.py (`Encoder` :375-466, `Decoder` :469-575,
`ResnetBlock` :89-148, `AttnBlock` :157-209, `Upsample` :49-64, `Downsample` :67-86) and
`AutoencoderKL` of modules/ldm/models/autoencoder.py:285-338 (same ctor `(ddconfig, lossconfig,
embed_dim, ...)`, same parameter names `encoder.* / decoder.* / quant_conv / post_quant_conv`).
It is a plain nn.Module (the reference's pytorch_lightning base only adds training plumbing).

Precision: the decoder's output must match the fp32 CPU path to 1e-3 per pixel (BASELINE.json), so by
default the VAE runs the fp32-class kernels (fp32 activations, split-bf16 x3 MFMA); bf16 parameters
or `compute_dtype=torch.bfloat16` select the 1-pass bf16 kernels.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .nn import Conv2d, Normalize


def _planes_ok(x, conv) -> bool:
    """fp32-class activations feeding a conv the MFMA path handles (not the thin-channel kernels): use split bf16 planes."""
    return x.dtype == torch.float32 and conv.in_channels % 8 == 0 and conv.in_channels >= 64 and conv.out_channels > 8


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("VAE Upsample without conv is not on the SD path")
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        # nearest 2x folded into the conv gather (model.py:60-64); fp32-class: the residual stream is split into bf16 planes
        # first (one elementwise pass) so that the conv runs on the LDS-DMA kernel
        if _planes_ok(x, self.conv):
            hi, lo = ops.split_bf16(x)
            return self.conv(hi, x_lo=lo, upsample2x=True, gn_stats=True)  # feeds the next ResnetBlock's norm1
        return self.conv(x, upsample2x=True)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if not with_conv:
            raise NotImplementedError("VAE Downsample without conv is not on the SD path")
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        # F.pad(x, (0,1,0,1)) + stride-2 pad-0 conv (model.py:79-83) == asymmetric zero padding in the gather
        if _planes_ok(x, self.conv):
            hi, lo = ops.split_bf16(x)
            return self.conv(hi, x_lo=lo, padding=(0, 0, 1, 1))
        return self.conv(x, padding=(0, 0, 1, 1))


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        if temb_channels > 0:
            raise NotImplementedError("VAE ResnetBlock: temb_channels is 0 on the SD path (model.py:383,477)")
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        if ops.mx_conv_ok(x, self.conv1.weight) and self.conv2.in_channels % 64 == 0 and self.conv2.out_channels % 4 == 0:
            # opt-in (CRG_VAE_MX=1): the same block on MX planes (CRG_PREC_F16MX: one fp16 pass + fp8 cross terms), see ops.VAE_MX
            a16, a8 = self.norm1(x, silu=True, split="mx")
            h = self.conv1(a16, x_mx=a8, gn_stats=True)
            a16, a8 = self.norm2(h, silu=True, split="mx")
            if self.in_channels != self.out_channels:
                x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
            return self.conv2(a16, x_mx=a8, residual=x, gn_stats=True)
        if _planes_ok(x, self.conv1) and _planes_ok(x, self.conv2):
            # fp32-class: GroupNorm+swish writes the two bf16 planes the conv's LDS-DMA kernel stages (no fp32 round trip)
            hi, lo = self.norm1(x, silu=True, split=True)
            h = self.conv1(hi, x_lo=lo, gn_stats=True)  # statistics for norm2 out of the conv's epilogue
            hi, lo = self.norm2(h, silu=True, split=True)
            if self.in_channels != self.out_channels:
                x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
            return self.conv2(hi, x_lo=lo, residual=x, gn_stats=True)  # ... for the next block's norm1 / norm_out
        h = self.conv1(self.norm1(x, silu=True))
        h = self.norm2(h, silu=True)
        if self.in_channels != self.out_channels:
            x = self.conv_shortcut(x) if self.use_conv_shortcut else self.nin_shortcut(x)
        return self.conv2(h, residual=x)


class AttnBlock(nn.Module):
    """model.py:157-209: single-head attention over HW tokens, q/k/v/proj_out 1x1 convs with bias."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        x = ops.to_channels_last(x)
        b, c, h, w = x.shape
        t = ops.tokens_of(self.norm(x))
        q = ops.linear(t, self.q.weight, self.q.bias)
        k = ops.linear(t, self.k.weight, self.k.bias)
        vt = ops.linear_transposed(t, self.v.weight, self.v.bias)
        o = ops.attention(q, k, vt, 1, h * w, int(c) ** (-0.5))
        y = ops.linear(o, self.proj_out.weight, self.proj_out.bias, residual=ops.tokens_of(x))
        return ops.image_of(y, h, w)


def make_attn(in_channels, attn_type="vanilla"):
    # sgm adds "vanilla-xformers" / "memory-efficient-cross-attn" (sgm/modules/diffusionmodules/model.py:277-309,
    # sd_xl_base.yaml:82): same parameters, same function -> the one HIP block
    assert attn_type in ["vanilla", "vanilla-xformers", "linear", "none"], f'attn_type {attn_type} unknown'
    if attn_type in ("vanilla", "vanilla-xformers"):
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError("linear attention is not on the SD path")


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, double_z=True, use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.conv_in = Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block = block
            down.attn = attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        """x: channels-last image in the compute dtype -> moments pre-quant_conv (model.py:441-466)."""
        h = self.conv_in(x)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        return self.conv_out(self.norm_out(h, silu=True))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False,
                 attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if give_pre_end or tanh_out:
            raise NotImplementedError("Decoder: give_pre_end / tanh_out are not on the SD path")
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block = block
            up.attn = attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z):
        """z: channels-last latent in the compute dtype -> image (model.py:542-575)."""
        self.last_z_shape = z.shape
        h = self.conv_in(z)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block](h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        return self.conv_out(self.norm_out(h, silu=True))


class DiagonalGaussianDistribution(object):
    """modules/ldm/modules/distributions/distributions.py:24-37 (tiny elementwise glue on [b,8,L,L]; the
    RNG stays torch's so seeds behave as in the reference)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, noise: Optional[torch.Tensor] = None):
        if noise is None:
            noise = torch.randn(self.mean.shape).to(device=self.parameters.device)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.loss = nn.Identity()
        assert ddconfig["double_z"]
        self.quant_conv = Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.compute_dtype: Optional[torch.dtype] = None
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)

    def resolve_compute_dtype(self) -> torch.dtype:
        if self.compute_dtype is not None:
            return self.compute_dtype
        return ops.HALF if self.post_quant_conv.weight.dtype in (torch.bfloat16, torch.float16) else torch.float32

    def encode(self, x):
        """image [b,3,H,W] NCHW in [-1,1] -> posterior over [b,4,H/8,W/8] (autoencoder.py:324-331)."""
        cdt = self.resolve_compute_dtype()
        h = self.encoder(ops.nchw_to_nhwc(x, cdt))
        moments = self.quant_conv(h)
        return DiagonalGaussianDistribution(ops.nhwc_to_nchw(moments, torch.float32))

    def decode(self, z):
        """latent [b,4,L,L] NCHW -> image [b,3,8L,8L] NCHW fp32 (autoencoder.py:333-338)."""
        cdt = self.resolve_compute_dtype()
        h = self.post_quant_conv(ops.nchw_to_nhwc(z, cdt))
        dec = self.decoder(h)
        return ops.nhwc_to_nchw(dec, torch.float32)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
