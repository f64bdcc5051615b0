"""SD1.5 UNet on HIP kernels - drop-in for modules/ldm/modules/diffusionmodules/openaimodel.py.

Same constructor signature (openaimodel.py:447-481, incl. the LoRA / IP-Adapter arguments injected
at sd/image_generator.py:314-320), same module tree and parameter names (`time_embed`,
`input_blocks`, `middle_block`, `output_blocks`, `out`; pinned by
test/ldm/ldm_instantiation_test.py:21-25), same `TimestepBlock` / `SpatialTransformer` dispatch in
`TimestepEmbedSequential` (openaimodel.py:80-92) so ControlNet's subclass (cldm.py:28-70) and
`DiffusionWrapper.forward` (ddpm.py:1517-1519) call it unchanged.

MI355X-first differences in HOW it computes (results are the same function):
  * activations are channels-last bf16 (or fp32 for the fp32-class path) from the first conv to the
    last; NCHW fp32 only at the two boundaries;
  * GroupNorm+SiLU is one fused HBM-bound pass feeding an implicit-GEMM MFMA conv whose epilogue adds
    bias, the timestep-embedding vector and the residual;
  * `th.cat([h, hs.pop()], dim=1)` (openaimodel.py:808) is never materialised: the output blocks get
    the pair and GroupNorm / the 1x1 skip conv read both halves through two pointers;
  * nearest-2x Upsample is folded into the following conv's gather (openaimodel.py:120-122).
"""
from __future__ import annotations

from abc import abstractmethod
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .nn import Conv2d, Linear, SiLU, conv_nd, linear, normalization, timestep_embedding, zero_module
from .transformer import SpatialTransformer, exists


_embcat_cache = ops.TensorKeyedCache()


class TimestepBlock(nn.Module):
    """Any module where forward() takes timestep embeddings as a second argument (openaimodel.py:62-71)."""

    @abstractmethod
    def forward(self, x, emb):
        ...


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """openaimodel.py:74-92.  `x` may be the pair (h, skip) standing for cat([h, skip], dim=1)."""

    def forward(self, x, emb, context=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    """openaimodel.py:95-123: nearest 2x then (optional) 3x3 conv - fused into one conv launch."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if dims != 2 or not use_conv:
            raise NotImplementedError("Upsample: only dims=2 with use_conv=True is on the SD path")
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.conv(x, upsample2x=True, gn_stats=True)  # feeds the next ResBlock's GroupNorm


class Downsample(nn.Module):
    """openaimodel.py:138-164: 3x3 stride-2 pad-1 conv (`op`)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if dims != 2 or not use_conv:
            raise NotImplementedError("Downsample: only dims=2 with use_conv=True is on the SD path")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.op(x, gn_stats=True)  # feeds the next ResBlock's GroupNorm (and, as a skip, an output block's)


class ResBlock(TimestepBlock):
    """openaimodel.py:167-279 (the SD configuration: no up/down, no scale-shift norm, dropout 0)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False):
        super().__init__()
        if up or down or use_scale_shift_norm or dims != 2:
            raise NotImplementedError("ResBlock: resblock_updown / use_scale_shift_norm / dims != 2 are not on the SD path")
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.updown = False
        self.in_layers = nn.Sequential(normalization(channels), SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def forward(self, x, emb):
        return self._forward(x, emb)

    def _forward(self, x, emb):
        x2 = None
        if isinstance(x, tuple):
            x, x2 = x
        # emb_layers = SiLU -> Linear (openaimodel.py:222-228).  UNetModel.forward precomputes all 22 blocks'
        # projections with ONE GEMM per step (the embedding is step-constant across blocks) and hands each block
        # its [N, Cout] fp32 slice; a ResBlock used on its own computes it here.
        pre = getattr(emb, "_crg_emb_out", None)
        emb_out = pre.get(id(self)) if pre is not None else None
        if emb_out is None:
            emb_out = self.emb_layers[1](ops.silu(emb), out_dtype=torch.float32)  # [N, Cout] fp32
        h = self.in_layers[0](x, silu=True, x2=x2)                              # GN32 + SiLU (one tensor even for a pair)
        h = self.in_layers[2](h, cvec=emb_out, gn_stats=True)                   # conv + bias + emb_out[:, :, None, None]; statistics for out_layers[0]
        h = self.out_layers[0](h, silu=True)
        if isinstance(self.skip_connection, nn.Identity):
            skip = x if x2 is None else torch.cat([x, x2], dim=1)
        else:
            skip = self.skip_connection(x, x2=x2) if x2 is not None else self.skip_connection(x)
        return self.out_layers[3](h, residual=skip, gn_stats=True)              # conv + bias + skip(x); statistics for the next GroupNorm


class UNetModel(nn.Module):
    """openaimodel.py:417-816.  Only the structure the shipped configs use is accepted:
    use_spatial_transformer=True, conv_resample=True, resblock_updown=False, dims=2, num_classes=None
    (v1-inference.yaml:29-44, cldm_v15.yaml); anything else raises instead of silently diverging."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                 n_embed=None, legacy=True, disable_self_attentions=None, num_attention_blocks=None,
                 disable_middle_self_attn=False, use_linear_in_transformer=False, lora_ranks: List[int] = None,
                 lora_weights: List[float] = None, ipa_scale=1.0, ipa_num_tokens=0):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("UNetModel: the SD path always uses use_spatial_transformer=True with a context_dim")
        if not conv_resample or resblock_updown or dims != 2 or num_classes is not None or n_embed is not None or use_scale_shift_norm:
            raise NotImplementedError("UNetModel: unsupported structural option for the SD path")
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, 'Either num_heads or num_head_channels has to be set'
        if num_head_channels == -1:
            assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
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
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32  # kept for ControlNet's `x.type(self.dtype)` (cldm.py:52)
        self.compute_dtype: Optional[torch.dtype] = None           # None = follow the parameters / autocast
        self._ctx_cast = None
        self._resblocks = None
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = False

        def heads_for(ch, nh):
            # openaimodel.py:575-582: legacy=False => dim_head = ch // num_heads (SD1.5: 40/80/160)
            if num_head_channels == -1:
                dh = ch // nh
            else:
                nh = ch // num_head_channels
                dh = num_head_channels
            if legacy:
                dh = ch // nh
            return nh, dh

        def st(ch, nh, **kw):
            n, dh = heads_for(ch, nh)
            return SpatialTransformer(ch, n, dh, depth=transformer_depth, context_dim=context_dim, lora_ranks=lora_ranks,
                                      lora_weights=lora_weights, ipa_scale=ipa_scale, ipa_num_tokens=ipa_num_tokens, **kw)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        self._feature_size = model_channels
        input_block_chans = [model_channels]
        ch = model_channels
        ds = 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    if not exists(num_attention_blocks) or nr < num_attention_blocks[level]:
                        # NB the reference does not forward disable_self_attn / use_linear here (openaimodel.py:597-604)
                        layers.append(st(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                out_ch = ch
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=out_ch)))
                ch = out_ch
                input_block_chans.append(ch)
                ds *= 2
                self._feature_size += ch

        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint),
            st(ch, num_heads, disable_self_attn=disable_middle_self_attn, use_linear=use_linear_in_transformer,
               use_checkpoint=use_checkpoint),
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint))
        self._feature_size += ch

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, time_embed_dim, dropout, out_channels=model_channels * mult, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    disabled_sa = disable_self_attentions[level] if exists(disable_self_attentions) else False
                    if not exists(num_attention_blocks) or i < num_attention_blocks[level]:
                        layers.append(st(ch, num_heads, disable_self_attn=disabled_sa, use_linear=use_linear_in_transformer,
                                         use_checkpoint=use_checkpoint))
                if level and i == self.num_res_blocks[level]:
                    out_ch = ch
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=out_ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch

        self.out = nn.Sequential(normalization(ch), SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))

    # -- per-step batched timestep-embedding projections ------------------------------
    def _emb_projections(self, emb: torch.Tensor):
        return self._emb_table(self._emb_rows(emb))

    def _emb_rows(self, emb: torch.Tensor):
        """All ResBlocks' `emb_layers` Linear(1280 -> Cout) as one [N, 1280] x [sum Cout, 1280]^T GEMM -> [N, sum Cout] fp32."""
        blocks = self._resblocks
        if blocks is None:
            blocks = self._resblocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        ws = tuple(b.emb_layers[1].weight for b in blocks)
        bs = tuple(b.emb_layers[1].bias for b in blocks)
        hit = _embcat_cache.get(ws + bs)
        if hit is None:
            with torch.no_grad():
                hit = _embcat_cache.put(ws + bs, (), (torch.cat([w.detach() for w in ws], 0).contiguous(),
                                                      torch.cat([b.detach().float() for b in bs], 0).contiguous()))
        wcat, bcat = hit
        return ops.linear(ops.silu(emb), wcat, bcat, out_dtype=torch.float32)

    def _emb_table(self, rows: torch.Tensor):
        """{id(ResBlock): its [N, Cout] slice} of one [N, sum Cout] fp32 projection tensor (_emb_projections / time_rows)."""
        blocks = self._resblocks
        if blocks is None:
            blocks = self._resblocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        table, off = {}, 0
        for b in blocks:
            n = b.out_channels
            table[id(b)] = rows[:, off:off + n]
            off += n
        assert off == rows.shape[1], "time rows: width does not match this UNet's ResBlocks"
        return table

    @torch.no_grad()
    def time_rows(self, t_table: torch.Tensor) -> torch.Tensor:
        """Everything of a denoising call that depends on the timestep alone, for a whole schedule at once: `t_table` [S, N] (the
        timesteps of S sampler steps, already expanded to the UNet batch) -> [S, N, sum Cout] fp32, the 22 ResBlocks' `emb_layers`
        outputs (openaimodel.py:793-796 + :222-228 per block) from ONE pass of the three GEMMs at S * N rows.  A sampler that knows
        its schedule hands row s to step s with `ops.attach_time_rows(timesteps, rows[s], unet)`; forward then skips the embedding MLP
        (5 launches and 55 MB of weights per call at batch 8)."""
        s, n = t_table.shape
        cdt = self.resolve_compute_dtype()
        t_emb = timestep_embedding(t_table.reshape(-1), self.model_channels, dtype=cdt)
        emb = self.time_embed[2](self.time_embed[0](t_emb, act="silu"))
        return self._emb_rows(emb).view(s, n, -1)

    # -- dtype policy ---------------------------------------------------------------------------
    def resolve_compute_dtype(self) -> torch.dtype:
        if self.compute_dtype is not None:
            return self.compute_dtype
        p = self.time_embed[0].weight.dtype
        if p in (torch.bfloat16, torch.float16) or torch.is_autocast_enabled():
            return ops.HALF  # the library's half type: bfloat16, or fp16 under CRG_HALF=f16
        return torch.float32

    def _prologue(self, timesteps, context):
        """timestep embedding MLP (openaimodel.py:793-796) + the per-step batched ResBlock projections + one cast of the
        context per context tensor -> (compute dtype, emb, context)."""
        cdt = self.resolve_compute_dtype()
        rows = ops.time_rows_of(timesteps, self)
        if rows is not None:
            # hoisted by the sampler (time_rows): `emb` itself has no other reader in this network, the rows stand in for it
            assert rows.dim() == 2 and rows.shape[0] == timesteps.shape[0] and rows.dtype == torch.float32
            emb = rows
            emb._crg_emb_out = self._emb_table(rows)
        else:
            t_emb = timestep_embedding(timesteps, self.model_channels, dtype=cdt)
            emb = self.time_embed[0](t_emb, act="silu")
            emb = self.time_embed[2](emb)
            emb._crg_emb_out = self._emb_projections(emb)
        if context is not None and context.dtype != cdt:
            # cast once per context tensor (identity + version), so that the cross-attention K/V cache, which is
            # keyed on the tensor it receives, keeps hitting across sampler steps
            c = self._ctx_cast
            if c is None or c[0] is not context or c[1] != context._version or c[2].dtype != cdt:
                self._ctx_cast = c = (context, context._version, context.to(cdt))
            context = c[2]
        return cdt, emb, context

    def _epilogue(self, h, x):
        """out = conv3x3(SiLU(GN32(h))) (openaimodel.py:752-756,816), back to NCHW in x.dtype (:810)."""
        h = self.out[0](h, silu=True)
        h = self.out[2](h)
        return ops.nhwc_to_nchw(h, x.dtype if x.dtype in (torch.float32, ops.HALF) else torch.float32).to(x.dtype)

    # -- CFG-shared prefix (ops.mark_cfg_dup) ---------------------------------------------------------
    def _cfg_split_index(self):
        """Index of the first input block that reads the conditioning, if it can take a half batch and hand back both halves
        (a ResBlock followed by a SpatialTransformer whose first attention is a self-attention), else None."""
        r = self.__dict__.get("_crg_cfg_split", False)
        if r is False:
            r = None
            for i, m in enumerate(self.input_blocks):
                sts = [l for l in m if isinstance(l, SpatialTransformer)]
                if sts:
                    layers = list(m)
                    if len(sts) == 1 and layers[-1] is sts[0] and sts[0].cfg_dup_ok() and all(isinstance(l, TimestepBlock) for l in layers[:-1]):
                        r = i
                    break
            self.__dict__["_crg_cfg_split"] = r
        return r

    def _input_blocks(self, x, cdt, emb, context, cfg_dup: bool):
        """The encoder half (openaimodel.py:804-806): returns (h, hs).  With `cfg_dup` - x is cat([x'] * 2) and the timesteps are
        doubled likewise, the caller's promise - everything before the first cross-attention runs on ONE half: both halves would
        compute the same values there (conv_in, the first ResBlock, GroupNorm + proj_in, LayerNorm + Q | K | V, the 64x64 self-attention,
        its out-projection, LayerNorm + to_q)."""
        split = self._cfg_split_index() if (cfg_dup and ops.CFG_SHARE and x.shape[0] % 2 == 0 and context is not None) else None
        hs = []
        if split is None:
            h = ops.nchw_to_nhwc(x, cdt)
            for module in self.input_blocks:
                h = module(h, emb, context)
                hs.append(h)
            return h, hs
        half = x.shape[0] // 2
        emb_h = emb[:half]
        emb_h._crg_emb_out = {k: v[:half] for k, v in emb._crg_emb_out.items()}
        h = ops.nchw_to_nhwc(x[:half], cdt)
        for i, module in enumerate(self.input_blocks):
            if i < split:
                h = module(h, emb_h, context)
                hs.append(ops.dup_batch(h))  # the skip connection feeds an output block, which runs the whole batch
            elif i == split:
                layers = list(module)
                for layer in layers[:-1]:
                    h = layer(h, emb_h)
                h = layers[-1](h, context, cfg_dup=True)
                hs.append(h)
            else:
                h = module(h, emb, context)
                hs.append(h)
        return h, hs

    def forward(self, x, timesteps=None, context=None, y=None, cfg_dup: bool = False, **kwargs):
        """x [N, C, H, W] (any float dtype, NCHW) , timesteps [N] (may be fractional), context [N, T, D]
        -> eps [N, C_out, H, W] in x.dtype (openaimodel.py:780-816).  `cfg_dup` (or ops.mark_cfg_dup(x)): see _input_blocks."""
        assert y is None, "must specify y if and only if the model is class-conditional"
        cdt, emb, context = self._prologue(timesteps, context)
        h, hs = self._input_blocks(x, cdt, emb, context, cfg_dup or getattr(x, "_crg_cfg_dup", False))
        h = self.middle_block(h, emb, context)
        for module in self.output_blocks:
            h = module((h, hs.pop()), emb, context)
        return self._epilogue(h, x)
