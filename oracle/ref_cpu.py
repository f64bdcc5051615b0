"""ORACLE — CPU restatement of the reference's Stable Diffusion denoising path.

TEST INFRASTRUCTURE ONLY.  Nothing under `cremage_amd/` may import this file; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do.

What it is: a functional (state-dict in, tensor out), fp32/fp64, CPU-only
restatement of the floating-point algorithm of HowToSD/cremage's hot path, written
from the reference source; every function cites the reference file:line it follows
(paths relative to the reference root).  Arithmetic primitives are plain PyTorch CPU
ops (the reference's own L7 backend) evaluated in the dtype of the
inputs - there is no fp16 cast anywhere (the reference's Apple-MPS `half()` casts,
openaimodel.py:85-90,794-795,814-815, are deliberately NOT reproduced: BASELINE.json
config 1 is the fp32 CPU path).

Pinning: `tests/test_oracle_golden.py` checks every function here against the
golden vectors in `tests/golden/`, which were produced by importing and running the
reference's own modules in the build container (`oracle/gen_golden.py`).
Parity status: PINNED by reference-generated fixtures (the reference's own tests hold
no numeric vectors for this path - SURVEY.md §4 - except the alphas_cumprod table,
which is also checked).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------
# leaf arithmetic
# ----------------------------------------------------------------------------
def group_norm(x, weight, bias, groups: int, eps: float):
    """GroupNorm32.forward util.py:214-216 (eps 1e-5) / Normalize attention.py:189-190,
    model.py:45-46 (eps 1e-6): per (sample, group) mean and *biased* variance over
    (C/groups, H, W), then per-channel affine."""
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    y = ((xg - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = [1, c] + [1] * (x.ndim - 2)
    return y * weight.reshape(shape) + bias.reshape(shape)


def silu(x):
    """nn.SiLU openaimodel.py:207,231,754; `nonlinearity` model.py:40-42."""
    return x * torch.sigmoid(x)


def layer_norm(x, weight, bias, eps: float = 1e-5):
    """nn.LayerNorm(dim) attention.py:900-902 (default eps 1e-5, biased variance)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def gelu_erf(x):
    """F.gelu default (exact erf form) attention.py:96."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x, sd: SD, p: str, bias: bool = True):
    w = sd[p + ".weight"].to(x.dtype)
    b = sd[p + ".bias"].to(x.dtype) if bias and (p + ".bias") in sd else None
    return F.linear(x, w, b)


def conv2d(x, sd: SD, p: str, stride: int = 1, padding: int = 0):
    w = sd[p + ".weight"].to(x.dtype)
    b = sd[p + ".bias"].to(x.dtype) if (p + ".bias") in sd else None
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def timestep_embedding(timesteps, dim: int, max_period: int = 10000):
    """util.py:151-171.  freqs in fp32 exactly as the reference builds them
    (exp(-ln(max_period) * arange(half)/half)); args = t[:,None].float() * freqs;
    cat([cos, sin]); zero-pad when dim is odd.  t may be fractional
    (k-diffusion sigma_to_t, external.py:66-78)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ----------------------------------------------------------------------------
# UNet blocks (modules/ldm/modules/diffusionmodules/openaimodel.py, modules/ldm/modules/attention.py)
# ----------------------------------------------------------------------------
def res_block(x, emb, sd: SD, p: str):
    """ResBlock._forward openaimodel.py:259-279 (non-updown, no scale-shift norm):
    h = conv3x3(SiLU(GN32(x))); h += Linear(SiLU(emb))[:, :, None, None];
    h = conv3x3(SiLU(GN32(h))) [dropout p=0]; return skip(x) + h,
    skip = Identity or 1x1 conv (openaimodel.py:238-245)."""
    h = group_norm(x, sd[p + ".in_layers.0.weight"].to(x.dtype), sd[p + ".in_layers.0.bias"].to(x.dtype), 32, 1e-5)
    h = conv2d(silu(h), sd, p + ".in_layers.2", padding=1)
    emb_out = linear(silu(emb), sd, p + ".emb_layers.1").to(h.dtype)
    h = h + emb_out[:, :, None, None]
    h = group_norm(h, sd[p + ".out_layers.0.weight"].to(x.dtype), sd[p + ".out_layers.0.bias"].to(x.dtype), 32, 1e-5)
    h = conv2d(silu(h), sd, p + ".out_layers.3", padding=1)
    if (p + ".skip_connection.weight") in sd:
        w = sd[p + ".skip_connection.weight"]
        x = conv2d(x, sd, p + ".skip_connection", padding=(w.shape[-1] - 1) // 2)
    return x + h


def downsample(x, sd: SD, p: str):
    """Downsample.forward openaimodel.py:162-164: 3x3 stride-2 pad-1 conv (`op`)."""
    return conv2d(x, sd, p + ".op", stride=2, padding=1)


def upsample(x, sd: SD, p: str):
    """Upsample.forward openaimodel.py:113-123: nearest 2x then 3x3 conv."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return conv2d(x, sd, p + ".conv", padding=1)


def _lora_delta(x, sd: SD, down_fmt: str, up_fmt: str, alpha_fmt: str, lora_ranks, lora_weights, conv: bool = False):
    """The additive LoRA branch the reference repeats at 8 sites
    (attention.py:88-96,157-168,616-641,685-692,1038-1056):
    sum_i up_i(down_i(x)) * lora_weights[i] * (alpha_i / rank_i)."""
    d_sum = 0
    for i, r in enumerate(lora_ranks or []):
        dn = sd[down_fmt.format(i)].to(x.dtype)
        up = sd[up_fmt.format(i)].to(x.dtype)
        d = F.conv2d(F.conv2d(x, dn), up) if conv else F.linear(F.linear(x, dn), up)
        d_sum = d_sum + d * lora_weights[i] * (sd[alpha_fmt.format(i)].to(x.dtype) / r)
    return d_sum


def cross_attention(x, context, sd: SD, p: str, heads: int, lora_ranks=None, lora_weights=None,
                    ipa_scale: float = 1.0, ipa_num_tokens: int = 0):
    """CrossAttentionOriginal.forward attention.py:611-693 (the CPU/oracle path):
    q = W_q x, k = W_k c, v = W_v c (no bias) [+ LoRA]; split heads
    'b n (h d) -> (b h) n d'; softmax(q k^T * d^-0.5) v; merge heads;
    optional IP-Adapter FaceID second attention on the last `ipa_num_tokens` context
    tokens (attention.py:623-627,660-683); to_out Linear + bias [+ LoRA]."""
    lora_weights = lora_weights if lora_weights is not None else [1.0] * len(lora_ranks or [])
    h = heads
    q = linear(x, sd, p + ".to_q", bias=False)
    q = q + _lora_delta(x, sd, p + ".q_lora_downs.{}.weight", p + ".q_lora_ups.{}.weight", p + ".q_lora_alphas.{}", lora_ranks, lora_weights)
    context = x if context is None else context
    ipa_context = None
    if ipa_num_tokens > 0:
        end = context.shape[1] - ipa_num_tokens
        context, ipa_context = context[:, :end], context[:, end:]
    k = linear(context, sd, p + ".to_k", bias=False)
    k = k + _lora_delta(context, sd, p + ".k_lora_downs.{}.weight", p + ".k_lora_ups.{}.weight", p + ".k_lora_alphas.{}", lora_ranks, lora_weights)
    v = linear(context, sd, p + ".to_v", bias=False)
    v = v + _lora_delta(context, sd, p + ".v_lora_downs.{}.weight", p + ".v_lora_ups.{}.weight", p + ".v_lora_alphas.{}", lora_ranks, lora_weights)

    def attend(q, k, v):
        b, n, c = q.shape
        d = c // h
        scale = d ** -0.5
        qh = q.reshape(b, n, h, d).permute(0, 2, 1, 3)
        kh = k.reshape(b, k.shape[1], h, d).permute(0, 2, 1, 3)
        vh = v.reshape(b, v.shape[1], h, d).permute(0, 2, 1, 3)
        sim = torch.einsum("bhid,bhjd->bhij", qh, kh) * scale
        attn = sim.softmax(dim=-1)
        out = torch.einsum("bhij,bhjd->bhid", attn, vh)
        return out.permute(0, 2, 1, 3).reshape(b, n, c)

    out = attend(q, k, v)
    if ipa_num_tokens > 0:
        k2 = linear(ipa_context, sd, p + ".to_k_ipa", bias=False)
        v2 = linear(ipa_context, sd, p + ".to_v_ipa", bias=False)
        out = out + ipa_scale * attend(q, k2, v2)
    res = linear(out, sd, p + ".to_out.0")
    res = res + _lora_delta(out, sd, p + ".out_lora_downs.{}.weight", p + ".out_lora_ups.{}.weight", p + ".out_lora_alphas.{}", lora_ranks, lora_weights)
    return res


def feed_forward(x, sd: SD, p: str, lora_ranks=None, lora_weights=None):
    """FeedForward.forward attention.py:157-168 with GEGLU_with_lora.forward :88-96:
    proj = Linear(C -> 8C) [+LoRA]; x, gate = chunk(2); x * gelu(gate); Linear(4C -> C) [+LoRA]."""
    lora_weights = lora_weights if lora_weights is not None else [1.0] * len(lora_ranks or [])
    out = linear(x, sd, p + ".net.0.proj")
    out = out + _lora_delta(x, sd, p + ".net.0.proj_lora_downs.{}.weight", p + ".net.0.proj_lora_ups.{}.weight",
                            p + ".net.0.proj_lora_alphas.{}", lora_ranks, lora_weights)
    a, gate = out.chunk(2, dim=-1)
    hmid = a * gelu_erf(gate)
    y = linear(hmid, sd, p + ".net.2")
    y = y + _lora_delta(hmid, sd, p + ".net_2_lora_downs.{}.weight", p + ".net_2_lora_ups.{}.weight",
                        p + ".net_2_lora_alphas.{}", lora_ranks, lora_weights)
    return y


def basic_transformer_block(x, context, sd: SD, p: str, heads: int, disable_self_attn: bool = False, **kw):
    """BasicTransformerBlock._forward attention.py:908-912."""
    ln = lambda t, q: layer_norm(t, sd[q + ".weight"].to(t.dtype), sd[q + ".bias"].to(t.dtype))
    lora = dict(lora_ranks=kw.get("lora_ranks"), lora_weights=kw.get("lora_weights"))
    x = cross_attention(ln(x, p + ".norm1"), context if disable_self_attn else None, sd, p + ".attn1", heads, **lora) + x
    x = cross_attention(ln(x, p + ".norm2"), context, sd, p + ".attn2", heads, ipa_scale=kw.get("ipa_scale", 1.0),
                        ipa_num_tokens=kw.get("ipa_num_tokens", 0), **lora) + x
    x = feed_forward(ln(x, p + ".norm3"), sd, p + ".ff", **lora) + x
    return x


def spatial_transformer(x, context, sd: SD, p: str, heads: int, depth: int = 1, **kw):
    """SpatialTransformer.forward attention.py:1031-1057: GN(eps 1e-6) -> 1x1 proj_in
    [+LoRA] -> 'b c h w -> b (h w) c' -> blocks -> back -> 1x1 proj_out [+LoRA] -> + x_in."""
    lora_ranks = kw.get("lora_ranks")
    lora_weights = kw.get("lora_weights")
    lora_weights = lora_weights if lora_weights is not None else [1.0] * len(lora_ranks or [])
    b, c, hh, ww = x.shape
    x_in = x
    xn = group_norm(x, sd[p + ".norm.weight"].to(x.dtype), sd[p + ".norm.bias"].to(x.dtype), 32, 1e-6)
    y = conv2d(xn, sd, p + ".proj_in")
    y = y + _lora_delta(xn, sd, p + ".proj_in_lora_downs.{}.weight", p + ".proj_in_lora_ups.{}.weight",
                        p + ".proj_in_lora_alphas.{}", lora_ranks, lora_weights, conv=True)
    y = y.reshape(b, y.shape[1], hh * ww).permute(0, 2, 1)
    for d in range(depth):
        y = basic_transformer_block(y, context, sd, f"{p}.transformer_blocks.{d}", heads, **kw)
    y = y.permute(0, 2, 1).reshape(b, -1, hh, ww)
    z = conv2d(y, sd, p + ".proj_out")
    z = z + _lora_delta(y, sd, p + ".proj_out_lora_downs.{}.weight", p + ".proj_out_lora_ups.{}.weight",
                        p + ".proj_out_lora_alphas.{}", lora_ranks, lora_weights, conv=True)
    return z + x_in


def unet_layout(cfg: dict):
    """Walk UNetModel.__init__ openaimodel.py:548-750 and return, per block, the list of
    (kind, heads) layers - the same tree the reference builds (SD1.5: 12 input blocks,
    middle, 12 output blocks).  Only the options the shipped YAMLs use are restated:
    use_spatial_transformer=True, conv_resample=True, resblock_updown=False, dims=2,
    num_heads given (legacy False => dim_head = ch // num_heads, openaimodel.py:575-582)."""
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
    att = set(cfg["attention_resolutions"])
    heads = cfg["num_heads"]
    inp: List[list] = [[("conv_in", None)]]
    ds = 1
    for level in range(len(mult)):
        for _ in range(nrb[level]):
            layers = [("res", None)]
            if ds in att:
                layers.append(("st", heads))
            inp.append(layers)
        if level != len(mult) - 1:
            inp.append([("down", None)])
            ds *= 2
    mid = [("res", None), ("st", heads), ("res", None)]
    out: List[list] = []
    for level in reversed(range(len(mult))):
        for i in range(nrb[level] + 1):
            layers = [("res", None)]
            if ds in att:
                layers.append(("st", heads))
            if level and i == nrb[level]:
                layers.append(("up", None))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def _run_block(h, emb, context, sd: SD, p: str, layers, depth: int, **kw):
    """TimestepEmbedSequential.forward openaimodel.py:80-92 (dispatch by layer kind)."""
    for j, (kind, heads) in enumerate(layers):
        q = f"{p}.{j}"
        if kind == "conv_in":
            h = conv2d(h, sd, q, padding=1)
        elif kind == "res":
            h = res_block(h, emb, sd, q)
        elif kind == "st":
            h = spatial_transformer(h, context, sd, q, heads, depth=depth, **kw)
        elif kind == "down":
            h = downsample(h, sd, q)
        elif kind == "up":
            h = upsample(h, sd, q)
        else:
            raise ValueError(kind)
    return h


def unet_forward(sd: SD, cfg: dict, x, timesteps, context, control: Optional[Sequence[torch.Tensor]] = None, **kw):
    """UNetModel.forward openaimodel.py:780-816: t_emb -> time_embed MLP (:793-796);
    12 input blocks pushing skips (:803-805); middle (:806); output blocks with
    cat([h, hs.pop()], dim=1) (:807-809); out = conv3x3(SiLU(GN32(h))) (:752-756,816).
    `control` (optional) reproduces ControlledUnetModel.forward cldm.py:57-65:
    middle += control.pop(); each skip += control.pop()."""
    inp, mid, out = unet_layout(cfg)
    depth = cfg.get("transformer_depth", 1)
    t_emb = timestep_embedding(timesteps, cfg["model_channels"]).to(x.dtype)
    emb = linear(silu(linear(t_emb, sd, "time_embed.0")), sd, "time_embed.2")
    hs = []
    h = x
    for i, layers in enumerate(inp):
        h = _run_block(h, emb, context, sd, f"input_blocks.{i}", layers, depth, **kw)
        hs.append(h)
    h = _run_block(h, emb, context, sd, "middle_block", mid, depth, **kw)
    control = list(control) if control is not None else None
    if control is not None:
        h = h + control.pop()
    for i, layers in enumerate(out):
        skip = hs.pop()
        if control is not None:
            skip = skip + control.pop()
        h = torch.cat([h, skip], dim=1)
        h = _run_block(h, emb, context, sd, f"output_blocks.{i}", layers, depth, **kw)
    h = group_norm(h, sd["out.0.weight"].to(h.dtype), sd["out.0.bias"].to(h.dtype), 32, 1e-5)
    return conv2d(silu(h), sd, "out.2", padding=1)


def controlnet_forward(sd: SD, cfg: dict, x, hint, timesteps, context, **kw):
    """ControlNet.forward modules/cldm/cldm.py:319-342 (ctor :80-317): the UNet's encoder half + middle block with
    its own weights.  guided_hint = input_hint_block(hint) (:178-194: 8 convs 3x3, SiLU between, strides
    1,1,2,1,2,1,2,1, hint_channels -> 16,16,32,32,96,96,256 -> model_channels) is added to the output of the first
    input block only (:329-333); every input block's output goes through its own 1x1 `zero_convs[i]` (:335,
    make_zero_conv :316-317) and the middle block's through `middle_block_out` (:337-338).  Returns the list
    [zero_conv_0(h_0), ..., zero_conv_11(h_11), middle_block_out(h_mid)] (13 tensors for the SD1.5 layout)."""
    inp, mid, _ = unet_layout(cfg)
    depth = cfg.get("transformer_depth", 1)
    t_emb = timestep_embedding(timesteps, cfg["model_channels"]).to(x.dtype)
    emb = linear(silu(linear(t_emb, sd, "time_embed.0")), sd, "time_embed.2")
    g = hint
    strides = [1, 1, 2, 1, 2, 1, 2, 1]
    for j, st in enumerate(strides):
        g = conv2d(g, sd, f"input_hint_block.{2 * j}", stride=st, padding=1)
        if j + 1 < len(strides):
            g = silu(g)
    outs = []
    h = x
    for i, layers in enumerate(inp):
        h = _run_block(h, emb, context, sd, f"input_blocks.{i}", layers, depth, **kw)
        if i == 0:
            h = h + g
        outs.append(conv2d(h, sd, f"zero_convs.{i}.0"))
    h = _run_block(h, emb, context, sd, "middle_block", mid, depth, **kw)
    outs.append(conv2d(h, sd, "middle_block_out.0"))
    return outs


def control_ldm_apply_model(unet_sd: SD, cn_sd: SD, cfg: dict, x, t, c_crossattn, c_concat, control_scales=None,
                            only_mid_control: bool = False, **kw):
    """ControlLDM.apply_model cldm.py:374-393: control = control_model(x, hint=cat(c_concat,1), t, ctx); each scaled by
    control_scales[i] (:388, default 1.0 x 13 :360); eps = ControlledUnetModel(x, t, ctx, control, only_mid_control)
    (:28-70: `h += control.pop()` after the middle block; `hs.pop() + control.pop()` for every skip unless
    only_mid_control).  With c_concat None the plain UNet runs (:384-385)."""
    ctx = torch.cat(list(c_crossattn), 1)
    if c_concat is None:
        return unet_forward(unet_sd, cfg, x, t, ctx, **kw)
    control = controlnet_forward(cn_sd, cfg, x, torch.cat(list(c_concat), 1), t, ctx, **kw)
    if control_scales is not None:
        control = [c * s for c, s in zip(control, control_scales)]
    if only_mid_control:
        control = control[-1:]
        return _unet_forward_mid_only(unet_sd, cfg, x, t, ctx, control[0], **kw)
    return unet_forward(unet_sd, cfg, x, t, ctx, control=control, **kw)


def _unet_forward_mid_only(sd: SD, cfg: dict, x, timesteps, context, mid_control, **kw):
    """ControlledUnetModel.forward with only_mid_control=True (cldm.py:57-63): only the middle residual is added."""
    inp, mid, out = unet_layout(cfg)
    depth = cfg.get("transformer_depth", 1)
    t_emb = timestep_embedding(timesteps, cfg["model_channels"]).to(x.dtype)
    emb = linear(silu(linear(t_emb, sd, "time_embed.0")), sd, "time_embed.2")
    hs = []
    h = x
    for i, layers in enumerate(inp):
        h = _run_block(h, emb, context, sd, f"input_blocks.{i}", layers, depth, **kw)
        hs.append(h)
    h = _run_block(h, emb, context, sd, "middle_block", mid, depth, **kw) + mid_control
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(h, emb, context, sd, f"output_blocks.{i}", layers, depth, **kw)
    h = group_norm(h, sd["out.0.weight"].to(h.dtype), sd["out.0.bias"].to(h.dtype), 32, 1e-5)
    return conv2d(silu(h), sd, "out.2", padding=1)


# ----------------------------------------------------------------------------
# VAE (modules/ldm/modules/diffusionmodules/model.py, modules/ldm/models/autoencoder.py)
# ----------------------------------------------------------------------------
def vae_resnet_block(x, sd: SD, p: str):
    """ResnetBlock.forward model.py:128-148 with temb=None (temb_ch = 0, model.py:383,477):
    h = conv1(swish(GN(x))); h = conv2(swish(GN(h))); shortcut = nin_shortcut 1x1 when
    in != out (model.py:114-126,142-146)."""
    h = group_norm(x, sd[p + ".norm1.weight"].to(x.dtype), sd[p + ".norm1.bias"].to(x.dtype), 32, 1e-6)
    h = conv2d(silu(h), sd, p + ".conv1", padding=1)
    h = group_norm(h, sd[p + ".norm2.weight"].to(x.dtype), sd[p + ".norm2.bias"].to(x.dtype), 32, 1e-6)
    h = conv2d(silu(h), sd, p + ".conv2", padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = conv2d(x, sd, p + ".nin_shortcut")
    elif (p + ".conv_shortcut.weight") in sd:
        x = conv2d(x, sd, p + ".conv_shortcut", padding=1)
    return x + h


def vae_attn_block(x, sd: SD, p: str):
    """AttnBlock.forward model.py:185-209: single-head attention over HW tokens with
    1x1-conv q,k,v (with bias), scale c^-0.5, softmax over keys, 1x1 proj_out, + x."""
    h_ = group_norm(x, sd[p + ".norm.weight"].to(x.dtype), sd[p + ".norm.bias"].to(x.dtype), 32, 1e-6)
    q = conv2d(h_, sd, p + ".q")
    k = conv2d(h_, sd, p + ".k")
    v = conv2d(h_, sd, p + ".v")
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + conv2d(h_, sd, p + ".proj_out")


def vae_decoder(sd: SD, dd: dict, z, p: str = "decoder"):
    """Decoder.forward model.py:542-575: conv_in; mid(block_1, attn_1, block_2);
    for levels reversed: (num_res_blocks+1) ResnetBlocks [+attn], Upsample (nearest 2x +
    conv, model.py:60-64) except at level 0; GN -> swish -> conv_out."""
    nres = len(dd["ch_mult"])
    h = conv2d(z, sd, p + ".conv_in", padding=1)
    h = vae_resnet_block(h, sd, p + ".mid.block_1")
    h = vae_attn_block(h, sd, p + ".mid.attn_1")
    h = vae_resnet_block(h, sd, p + ".mid.block_2")
    for lvl in reversed(range(nres)):
        for ib in range(dd["num_res_blocks"] + 1):
            h = vae_resnet_block(h, sd, f"{p}.up.{lvl}.block.{ib}")
            if f"{p}.up.{lvl}.attn.{ib}.norm.weight" in sd:
                h = vae_attn_block(h, sd, f"{p}.up.{lvl}.attn.{ib}")
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv2d(h, sd, f"{p}.up.{lvl}.upsample.conv", padding=1)
    h = group_norm(h, sd[p + ".norm_out.weight"].to(h.dtype), sd[p + ".norm_out.bias"].to(h.dtype), 32, 1e-6)
    return conv2d(silu(h), sd, p + ".conv_out", padding=1)


def vae_encoder(sd: SD, dd: dict, x, p: str = "encoder"):
    """Encoder.forward model.py:441-466; Downsample.forward model.py:79-86 pads (0,1,0,1)
    then 3x3 stride-2 pad-0 conv."""
    nres = len(dd["ch_mult"])
    h = conv2d(x, sd, p + ".conv_in", padding=1)
    for lvl in range(nres):
        for ib in range(dd["num_res_blocks"]):
            h = vae_resnet_block(h, sd, f"{p}.down.{lvl}.block.{ib}")
            if f"{p}.down.{lvl}.attn.{ib}.norm.weight" in sd:
                h = vae_attn_block(h, sd, f"{p}.down.{lvl}.attn.{ib}")
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = conv2d(h, sd, f"{p}.down.{lvl}.downsample.conv", stride=2, padding=0)
    h = vae_resnet_block(h, sd, p + ".mid.block_1")
    h = vae_attn_block(h, sd, p + ".mid.attn_1")
    h = vae_resnet_block(h, sd, p + ".mid.block_2")
    h = group_norm(h, sd[p + ".norm_out.weight"].to(h.dtype), sd[p + ".norm_out.bias"].to(h.dtype), 32, 1e-6)
    return conv2d(silu(h), sd, p + ".conv_out", padding=1)


def autoencoder_decode(sd: SD, dd: dict, z):
    """AutoencoderKL.decode autoencoder.py:333-338: post_quant_conv (1x1) then Decoder."""
    return vae_decoder(sd, dd, conv2d(z, sd, "post_quant_conv"))


def autoencoder_encode_moments(sd: SD, dd: dict, x):
    """AutoencoderKL.encode autoencoder.py:324-331 up to the moments tensor
    (Encoder then quant_conv 1x1); the unconditional `x.half()` at :327 is NOT reproduced."""
    return conv2d(vae_encoder(sd, dd, x), sd, "quant_conv")


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution distributions.py:24-37: mean, logvar = chunk(2, dim=1);
    logvar clamped to [-30, 20]; sample = mean + exp(0.5*logvar) * noise."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def decode_first_stage(sd: SD, dd: dict, z, scale_factor: float = 0.18215):
    """LatentDiffusion.decode_first_stage ddpm.py:748,798: z = z / scale_factor; decode."""
    return autoencoder_decode(sd, dd, z / scale_factor)


def get_first_stage_encoding(sd: SD, dd: dict, x, noise, scale_factor: float = 0.18215):
    """encode_first_stage ddpm.py:861-898 + get_first_stage_encoding ddpm.py:575-582:
    scale_factor * posterior.sample()."""
    return scale_factor * gaussian_sample(autoencoder_encode_moments(sd, dd, x), noise)


# ----------------------------------------------------------------------------
# schedules and samplers (stay on PyTorch in the product too; restated here so the
# trajectory fixtures pin the whole loop)
# ----------------------------------------------------------------------------
def make_beta_schedule_linear(n_timestep: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012):
    """make_beta_schedule('linear') util.py:21-25: linspace(sqrt(s), sqrt(e), n, float64)**2."""
    return torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2


def alphas_cumprod(n_timestep: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012):
    """DDPM.register_schedule ddpm.py:134-186: cumprod(1 - betas) in float64, stored fp32."""
    betas = make_beta_schedule_linear(n_timestep, linear_start, linear_end)
    return torch.cumprod(1.0 - betas, dim=0)


def sigmas_table(acp32: torch.Tensor):
    """DiscreteEpsDDPMDenoiser.__init__ external.py:93-95: sigma = sqrt((1-a)/a) (fp32 buffer)."""
    return ((1 - acp32) / acp32) ** 0.5


def t_to_sigma(log_sigmas, t):
    """DiscreteSchedule.t_to_sigma external.py:80-84."""
    t = t.float()
    low, high, w = t.floor().long(), t.ceil().long(), t.frac()
    return ((1 - w) * log_sigmas[low] + w * log_sigmas[high]).exp()


def get_sigmas(sigmas, n: int):
    """DiscreteSchedule.get_sigmas external.py:59-64: t = linspace(T-1, 0, n); append 0."""
    t = torch.linspace(len(sigmas) - 1, 0, n)
    s = t_to_sigma(sigmas.log(), t)
    return torch.cat([s, s.new_zeros([1])])


def sigma_to_t(sigmas, sigma):
    """DiscreteSchedule.sigma_to_t external.py:66-78 (quantize=False): interpolate t in log-sigma."""
    log_sigmas = sigmas.log()
    log_sigma = sigma.log()
    dists = log_sigma - log_sigmas[:, None]
    low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=log_sigmas.shape[0] - 2)
    high_idx = low_idx + 1
    low, high = log_sigmas[low_idx], log_sigmas[high_idx]
    w = ((low - log_sigma) / (low - high)).clamp(0, 1)
    t = (1 - w) * low_idx + w * high_idx
    return t.view(sigma.shape)


def cfg_denoise(eps_fn, sigmas_tab, x, sigma, cond, uncond, cfg_scale: float):
    """One denoiser call as the sampler sees it:
    LDMWrapperForKDiffusion.apply_model ldm_wrapper_for_k_diffusion.py:48-101 (batch-doubling
    x_in = cat([x]*2), c_in = cat([uc, c]); e_uncond + s*(e - e_uncond)) around
    DiscreteEpsDDPMDenoiser.forward external.py:111-114 (c_in = 1/sqrt(sigma^2+1),
    t = sigma_to_t(sigma), denoised = input + eps * (-sigma)).
    NOTE the order: the CFG wrapper is OUTSIDE the denoiser, so CFG mixes *denoised*
    images, exactly as the reference nests them (k_diffusion_samplers.py:176-182)."""
    x_in = torch.cat([x] * 2)
    s_in = torch.cat([sigma] * 2)
    c_in = torch.cat([uncond, cond])
    c_scale = 1.0 / (s_in ** 2 + 1.0) ** 0.5
    t = sigma_to_t(sigmas_tab, s_in)
    eps = eps_fn(x_in * c_scale[:, None, None, None], t, c_in)
    den = x_in + eps * (-s_in)[:, None, None, None]
    d_u, d_c = den.chunk(2)
    return d_u + cfg_scale * (d_c - d_u)


def sample_euler(denoise, x, sigmas):
    """sample_euler k_diffusion/sampling.py:118-143 with s_churn = 0 (gamma = 0):
    d = (x - denoised)/sigma; x += d * (sigma_next - sigma).  (The reference also draws an
    unused randn_like per step, :128 - RNG state only, not restated.)"""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        den = denoise(x, sigmas[i] * s_in)
        d = (x - den) / sigmas[i]
        x = x + d * (sigmas[i + 1] - sigmas[i])
    return x


def kdiff_stochastic_encode(x0, t_enc: int, sampling_steps: int, noise, acp=None):
    """KDiffusionSamplerBase.stochastic_encode k_diffusion_samplers.py:255-296: DDPM index t = int(t_enc * 1000 / S);
    x_t = sqrt(acp[t]) * x0 + sqrt(1 - acp[t]) * noise."""
    acp = alphas_cumprod().float() if acp is None else acp
    t = int(t_enc * 1000.0 / sampling_steps)
    return acp[t].sqrt() * x0 + (1.0 - acp[t]).sqrt() * noise


def hires_latent_upscale(samples, factor):
    """image_generator.py:975: F.interpolate(samples, scale_factor, mode='bilinear', align_corners=False) in latent space."""
    return F.interpolate(samples, scale_factor=factor, mode="bilinear", align_corners=False)


def get_ancestral_step(sigma_from, sigma_to, eta: float = 1.0):
    """k_diffusion/sampling.py:51-58."""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def sample_euler_ancestral(denoise, x, sigmas, noises):
    """sample_euler_ancestral k_diffusion/sampling.py:147-163 with eta = 1, s_noise = 1;
    `noises[i]` replaces noise_sampler(sigma_i, sigma_{i+1}) = randn_like(x) (:61-62)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        den = denoise(x, sigmas[i] * s_in)
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1])
        d = (x - den) / sigmas[i]
        x = x + d * (sigma_down - sigmas[i])
        if sigmas[i + 1] > 0:
            x = x + noises[i] * sigma_up
    return x


def make_ddim_schedule(acp32: torch.Tensor, S: int, ddpm_steps: int = 1000):
    """DDIMSampler.make_schedule ddim.py:38-75 with eta = 0, 'uniform' discretisation
    (util.py:46-60): timesteps = arange(0, T, T//S) + 1; a_t = acp[timesteps];
    a_prev = [acp[0]] + acp[timesteps[:-1]]."""
    c = ddpm_steps // S
    ts = torch.arange(0, ddpm_steps, c) + 1
    a = acp32[ts]
    a_prev = torch.cat([acp32[:1], acp32[ts[:-1]]])
    return ts, a, a_prev


def ddim_stochastic_encode(x0, t_enc: int, a, noise):
    """DDIMSampler.stochastic_encode ddim.py:615-654: sqrt(a[t])*x0 + sqrt(1-a[t])*noise
    with t = t_enc - 1... the reference indexes the ddim tables with t (ddim.py:640-654)."""
    return a[t_enc].sqrt() * x0 + (1 - a[t_enc]).sqrt() * noise


def ddim_decode(eps_fn, x_latent, cond, uncond, cfg_scale: float, t_start: int, ts, a, a_prev):
    """DDIMSampler.decode ddim.py:657-676 -> p_sample_ddim :530-612, eta = 0:
    for index = t_start-1 .. 0: e = e_u + s (e_c - e_u) (:538-561);
    pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); x = sqrt(a_prev) pred_x0 + sqrt(1-a_prev) e."""
    x = x_latent
    for index in reversed(range(t_start)):
        step = ts[index]
        tt = torch.full((x.shape[0],), int(step), dtype=torch.long)
        e = eps_fn(torch.cat([x] * 2), torch.cat([tt] * 2), torch.cat([uncond, cond]))
        e_u, e_c = e.chunk(2)
        e = e_u + cfg_scale * (e_c - e_u)
        pred_x0 = (x - (1 - a[index]).sqrt() * e) / a[index].sqrt()
        x = a_prev[index].sqrt() * pred_x0 + (1 - a_prev[index]).sqrt() * e
    return x


# ----------------------------------------------------------------------------
# SDXL (sgm) twins: modules/sdxl/sgm/modules/{attention.py, diffusionmodules/openaimodel.py, denoiser*.py,
# discretizer.py, guiders.py, sampling.py}
# ----------------------------------------------------------------------------
def sgm_unet_layout(cfg: dict):
    """sgm UNetModel.__init__ openaimodel.py:629-826: per-level transformer depth, heads = ch // num_head_channels
    (:654-659), no attention where ds is not in attention_resolutions (SDXL: none at level 0)."""
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
    att = set(cfg["attention_resolutions"])
    depth = cfg.get("transformer_depth", 1)
    depth = [depth] * len(mult) if isinstance(depth, int) else list(depth)
    nhc = cfg.get("num_head_channels", -1)
    nh = cfg.get("num_heads", -1)
    heads_of = lambda ch: (ch // nhc) if nhc != -1 else nh
    inp = [[("conv_in", None, 0)]]
    ch, ds = mc, 1
    for level in range(len(mult)):
        for _ in range(nrb[level]):
            ch = mult[level] * mc
            layers = [("res", None, 0)]
            if ds in att:
                layers.append(("st", heads_of(ch), depth[level]))
            inp.append(layers)
        if level != len(mult) - 1:
            inp.append([("down", None, 0)])
            ds *= 2
    mid = [("res", None, 0), ("st", heads_of(ch), depth[-1]), ("res", None, 0)]
    out = []
    for level in reversed(range(len(mult))):
        for i in range(nrb[level] + 1):
            ch = mc * mult[level]
            layers = [("res", None, 0)]
            if ds in att:
                layers.append(("st", heads_of(ch), depth[level]))
            if level and i == nrb[level]:
                layers.append(("up", None, 0))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def sgm_spatial_transformer(x, context, sd: SD, p: str, heads: int, depth: int, use_linear: bool):
    """sgm SpatialTransformer.forward attention.py:1068-1133: with use_linear the 1x1 convs become nn.Linear applied
    after / before the rearrange - the same per-pixel affine map, so only the weight's shape differs."""
    b, c, hh, ww = x.shape
    x_in = x
    xn = group_norm(x, sd[p + ".norm.weight"].to(x.dtype), sd[p + ".norm.bias"].to(x.dtype), 32, 1e-6)
    y = xn.reshape(b, c, hh * ww).permute(0, 2, 1)
    w_in = sd[p + ".proj_in.weight"].to(x.dtype).reshape(-1, c)
    y = F.linear(y, w_in, sd[p + ".proj_in.bias"].to(x.dtype))
    for d in range(depth):
        y = basic_transformer_block(y, context, sd, f"{p}.transformer_blocks.{d}", heads)
    w_out = sd[p + ".proj_out.weight"].to(x.dtype)
    w_out = w_out.reshape(w_out.shape[0], -1)
    y = F.linear(y, w_out, sd[p + ".proj_out.bias"].to(x.dtype))
    return y.permute(0, 2, 1).reshape(b, -1, hh, ww) + x_in


def sgm_unet_forward(sd: SD, cfg: dict, x, timesteps, context, y=None):
    """sgm UNetModel.forward openaimodel.py:828-874: emb = time_embed(timestep_embedding(t)) + label_emb(y)
    (label_emb = Sequential(Sequential(Linear, SiLU, Linear)), :617-625,853-859); then the SD1.5 block walk."""
    inp, mid, out = sgm_unet_layout(cfg)
    use_linear = cfg.get("use_linear_in_transformer", False)
    t_emb = timestep_embedding(timesteps, cfg["model_channels"]).to(x.dtype)
    emb = linear(silu(linear(t_emb, sd, "time_embed.0")), sd, "time_embed.2")
    if cfg.get("num_classes") is not None:
        emb = emb + linear(silu(linear(y.to(x.dtype), sd, "label_emb.0.0")), sd, "label_emb.0.2")

    def run(h, p, layers):
        for j, (kind, heads, depth) in enumerate(layers):
            q = f"{p}.{j}"
            if kind == "conv_in":
                h = conv2d(h, sd, q, padding=1)
            elif kind == "res":
                h = res_block(h, emb, sd, q)
            elif kind == "st":
                h = sgm_spatial_transformer(h, context, sd, q, heads, depth, use_linear)
            elif kind == "down":
                h = downsample(h, sd, q)
            elif kind == "up":
                h = upsample(h, sd, q)
        return h

    hs = []
    h = x
    for i, layers in enumerate(inp):
        h = run(h, f"input_blocks.{i}", layers)
        hs.append(h)
    h = run(h, "middle_block", mid)
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(h, f"output_blocks.{i}", layers)
    h = group_norm(h, sd["out.0.weight"].to(h.dtype), sd["out.0.bias"].to(h.dtype), 32, 1e-5)
    return conv2d(silu(h), sd, "out.2", padding=1)


def legacy_ddpm_sigmas(n: int, num_timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012):
    """LegacyDDPMDiscretization.get_sigmas discretizer.py:51-78 (+ Discretization.__call__ :20-24 appending a zero):
    timesteps = linspace(T-1, 0, n, endpoint=False).astype(int)[::-1]; sigma = sqrt((1-a)/a) flipped to descending."""
    import numpy as np
    betas = make_beta_schedule_linear(num_timesteps, linear_start, linear_end).numpy()
    acp = np.cumprod(1.0 - betas, axis=0)
    if n < num_timesteps:
        ts = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        acp = acp[ts]
    sig = torch.tensor((1 - acp) / acp, dtype=torch.float32) ** 0.5
    return torch.cat([torch.flip(sig, (0,)), torch.zeros(1)])


def discrete_denoiser_table(num_idx: int = 1000):
    """DiscreteDenoiser.__init__ denoiser.py:42-59: sigmas = discretization(num_idx, do_append_zero=False, flip=True)
    -> the 1000 DDPM sigmas in ASCENDING order."""
    s = legacy_ddpm_sigmas(num_idx)[:-1]
    return torch.flip(s, (0,))


def sdxl_denoise(net_fn, table, x, sigma, cond: dict, uc: dict, cfg_scale: float):
    """EDMSampler.denoise sampling.py:97-122 = VanillaCFG.prepare_inputs guiders.py:38-65 (cat uc|c for crossattn and
    vector, x and sigma doubled) -> DiscreteDenoiser.forward denoiser.py:23-39,61-75 (sigma quantised to the nearest
    table entry; EpsScaling denoiser_scaling.py:29-37: c_skip 1, c_out -sigma, c_in 1/sqrt(sigma^2+1), c_noise = sigma
    -> its table INDEX) -> network(x*c_in, idx, cond) * c_out + x -> VanillaCFG.__call__ guiders.py:28-36."""
    x2 = torch.cat([x] * 2)
    s2 = torch.cat([sigma] * 2)
    c = {k: torch.cat((uc[k], cond[k]), 0) for k in cond}
    idx = (s2 - table[:, None]).abs().argmin(dim=0)
    sq = table[idx]
    c_in = 1 / (sq ** 2 + 1.0) ** 0.5
    c_noise = (sq - table[:, None]).abs().argmin(dim=0)
    out = net_fn(x2 * c_in[:, None, None, None], c_noise, c["crossattn"], c["vector"]) * (-sq)[:, None, None, None] + x2
    x_u, x_c = out.chunk(2)
    return x_u + cfg_scale * (x_c - x_u)


def img2img_prune_sigmas(sigmas, strength: float):
    """Img2ImgDiscretizationWrapper.__call__ scripts/demo/discretization.py:23-32: keep the LAST max(int(strength * len), 1)
    entries of the descending sigma list (which ends with the appended 0)."""
    s = torch.flip(sigmas, (0,))
    s = s[: max(int(strength * len(s)), 1)]
    return torch.flip(s, (0,))


def sdxl_img2img_latents(vae_sd: SD, dd: dict, img, enc_noise, fwd_noise, sigmas, scale_factor: float = 0.13025):
    """do_img2img sdxl_image_generator_utils.py:989-1009: z = scale_factor * posterior.sample() (sgm/models/diffusion.py:139-151);
    noised_z = (z + noise * sigma_0) / sqrt(1 + sigma_0^2) (the sampler multiplies the same factor back, sampling.py:83)."""
    z = scale_factor * gaussian_sample(autoencoder_encode_moments(vae_sd, dd, img), enc_noise)
    return z, (z + fwd_noise * sigmas[0]) / torch.sqrt(1.0 + sigmas[0] ** 2.0)


def sdxl_sample_euler_edm(net_fn, x, cond, uc, num_steps: int, cfg_scale: float, sigmas=None):
    """EulerEDMSampler (sampling.py:147-219,309-318) with s_churn = 0: x *= sqrt(1 + sigma_0^2) (:83); per step
    d = (x - denoised)/sigma, x += d * (sigma_next - sigma).  `sigmas`: an already pruned list (img2img)."""
    sigmas = legacy_ddpm_sigmas(num_steps) if sigmas is None else sigmas
    table = discrete_denoiser_table()
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        den = sdxl_denoise(net_fn, table, x, s_in * sigmas[i], cond, uc, cfg_scale)
        d = (x - den) / sigmas[i]
        x = x + d * (sigmas[i + 1] - sigmas[i])
    return x
