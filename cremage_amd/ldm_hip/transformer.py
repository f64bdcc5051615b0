"""SpatialTransformer / BasicTransformerBlock / CrossAttention / FeedForward on HIP kernels.

Drop-in for modules/ldm/modules/attention.py: same constructor signatures (incl. `lora_ranks`,
`lora_weights`, `ipa_scale`, `ipa_num_tokens`, attention.py:266-269,538-541,870-875,923-929), same
parameter names (so SD1.5 checkpoints, the 792 LoRA keys of
cremage/utils/sd15_weight_list_with_lora.py and `to_k_ipa/to_v_ipa` load unchanged), same
`ATTENTION_MODES` registry (attention.py:865-869) with every mode mapped to the one HIP class.

What differs is only HOW forward computes:
  * tokens stay [B, HW, C] = the channels-last image itself: no `b c h w -> b (hw) c` copies
    (attention.py:1045,1048);
  * q/k/v/out, proj_in/out, GEGLU and net.2 are MFMA GEMMs with bias / residual / GEGLU fused in the
    epilogue; in bf16 the projections that share an input (q|k|v of self-attention, k|v of the context) are ONE
    launch over row-stacked weights and the flash kernel reads V row-major (transposing LDS read);
  * softmax(QK^T)V is one flash-style kernel, never materialising the N x N scores;
  * LoRA branches (attention.py:88-96,157-168,616-641,685-692,1038-1056) are folded into an
    effective weight W + sum_i w_i (alpha_i / r_i) Up_i Down_i when the packed weight is (re)built;
  * cross-attention K / V^T are cached while the very same context tensor object is passed again
    (the context is step-invariant).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .nn import Conv2d, Linear, Normalize, zero_module


import os

_SELF_QKV = os.environ.get("CRG_SELF_QKV", "1") != "0"  # dev knob: 0 = self-attention as fused Q|K GEMM + transposed-V GEMM (round-1 form)


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def zero_init_module(module):
    for p in module.parameters():
        p.data.zero_()
    return module


# ---------------------------------------------------------------------------------------------- LoRA folding
_merge_cache = ops.TensorKeyedCache()
_qk_cache = ops.TensorKeyedCache()


def _tkey(t: torch.Tensor):
    return (id(t), t.data_ptr(), t._version, t.dtype, t.device, tuple(t.shape))


def effective_weight(base: torch.Tensor, downs, ups, alphas, ranks, weights) -> torch.Tensor:
    """W + sum_i weights[i] * (alpha_i / rank_i) * Up_i @ Down_i  (fp32, cached on the identity and
    version of every operand).  Equals the reference's additive branch
    `d = up(down(x)); out += d * w * (alpha / rank)` (attention.py:616-619 and its 7 siblings)."""
    if not ranks:
        return base
    srcs = (base,) + tuple(d.weight for d in downs) + tuple(u.weight for u in ups) + tuple(alphas)
    extra = tuple(float(w) for w in weights)
    hit = _merge_cache.get(srcs, extra)
    if hit is not None:
        return hit
    with torch.no_grad():
        w = base.detach().float().reshape(base.shape[0], -1).clone()
        for d, u, a, lw, r in zip(downs, ups, alphas, weights, ranks):
            dn = d.weight.detach().float().reshape(d.weight.shape[0], -1)
            up = u.weight.detach().float().reshape(u.weight.shape[0], -1)
            w += (up @ dn) * (float(lw) * float(a.detach().float()) / float(r))
        w = w.reshape(base.shape).contiguous()
    return _merge_cache.put(srcs, extra, w)


def _lora_lists(obj, prefix: str, in_dim: int, out_dim: int, ranks, conv: bool = False):
    downs, ups, alphas = nn.ModuleList(), nn.ModuleList(), nn.ParameterList()
    for rank in ranks:
        if conv:
            downs.append(zero_init_module(nn.Conv2d(in_dim, rank, kernel_size=1, stride=1, padding=0, bias=False)))
            ups.append(zero_init_module(nn.Conv2d(rank, out_dim, kernel_size=1, stride=1, padding=0, bias=False)))
        else:
            downs.append(zero_init_module(nn.Linear(in_dim, rank, bias=False)))
            ups.append(zero_init_module(nn.Linear(rank, out_dim, bias=False)))
        alphas.append(nn.Parameter(torch.tensor(float(rank))))
    setattr(obj, prefix + "_lora_downs", downs)
    setattr(obj, prefix + "_lora_ups", ups)
    setattr(obj, prefix + "_lora_alphas", alphas)


def _eff(obj, base: torch.Tensor, prefix: str) -> torch.Tensor:
    return effective_weight(base, getattr(obj, prefix + "_lora_downs"), getattr(obj, prefix + "_lora_ups"),
                            getattr(obj, prefix + "_lora_alphas"), obj.lora_ranks, obj.lora_weights)


# ---------------------------------------------------------------------------------------------- feed-forward
class GEGLU_with_lora(nn.Module):
    """attention.py:66-96: proj Linear(dim_in -> 2*dim_out) [+LoRA], x * gelu(gate) (erf form)."""

    def __init__(self, dim_in, dim_out, lora_ranks: List[int] = None, lora_weights: List[float] = None):
        super().__init__()
        self.lora_ranks = lora_ranks if lora_ranks is not None else []
        self.lora_weights = lora_weights if lora_weights is not None else [1.0] * len(self.lora_ranks)
        self.proj = Linear(dim_in, dim_out * 2)
        _lora_lists(self, "proj", dim_in, dim_out * 2, self.lora_ranks)

    def forward(self, x, ln: Optional[nn.LayerNorm] = None):
        """`ln`: the LayerNorm in front of this projection, to be applied to x first - fused into the GEMM launch where the
        row-resident kernel takes the shape (ops.ln_linear_ok), as a separate pass otherwise."""
        return ops.ln_linear_auto(x, ln, _eff(self, self.proj.weight, "proj"), self.proj.bias, act="geglu")


class FeedForward(nn.Module):
    """attention.py:119-168 (glu=True is what SD uses, BasicTransformerBlock passes gated_ff=True)."""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0., lora_ranks: List[int] = None,
                 lora_weights: List[float] = None):
        super().__init__()
        self.lora_ranks = lora_ranks if lora_ranks is not None else []
        self.lora_weights = lora_weights if lora_weights is not None else [1.0] * len(self.lora_ranks)
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        if not glu:
            raise NotImplementedError("FeedForward(glu=False) is not on the SD path (attention.py:135-137)")
        if dropout != 0.:
            raise NotImplementedError("inference-only: dropout must be 0")
        self.net = nn.ModuleList([GEGLU_with_lora(dim, inner_dim, lora_ranks=lora_ranks, lora_weights=lora_weights),
                                  nn.Dropout(dropout), Linear(inner_dim, dim_out)])
        _lora_lists(self, "net_2", inner_dim, dim_out, self.lora_ranks)

    def forward(self, x, residual=None, ln: Optional[nn.LayerNorm] = None, out_stats: bool = False):
        """`out_stats`: the result feeds the next block's LayerNorm - its row statistics come out of this launch (ops.linear, row_stats)."""
        h = self.net[0](x, ln=ln)
        return ops.linear(h, _eff(self, self.net[2].weight, "net_2"), self.net[2].bias, residual=residual, row_stats=out_stats)


# ---------------------------------------------------------------------------------------------- attention
class CrossAttention(nn.Module):
    """HIP counterpart of CrossAttentionOriginal / CrossAttention / MemoryEfficientCrossAttention
    (attention.py:537-693, 265-534, 696-861): identical parameters, one flash-attention kernel."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0., lora_ranks: List[int] = None,
                 lora_weights: List[float] = None, ipa_scale=1.0, ipa_num_tokens=0):
        super().__init__()
        self.lora_ranks = lora_ranks if lora_ranks is not None else []
        self.lora_weights = lora_weights if lora_weights is not None else [1.0] * len(self.lora_ranks)
        self.ipa_scale = ipa_scale
        self.ipa_num_tokens = ipa_num_tokens
        inner_dim = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = Linear(query_dim, inner_dim, bias=False)
        self.to_k = Linear(context_dim, inner_dim, bias=False)
        self.to_v = Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(Linear(inner_dim, query_dim), nn.Dropout(dropout))
        _lora_lists(self, "q", query_dim, inner_dim, self.lora_ranks)
        _lora_lists(self, "k", context_dim, inner_dim, self.lora_ranks)
        _lora_lists(self, "v", context_dim, inner_dim, self.lora_ranks)
        _lora_lists(self, "out", inner_dim, query_dim, self.lora_ranks)
        if self.ipa_num_tokens > 0:  # IP-Adapter FaceID, attention.py:606-609
            self.to_k_ipa = Linear(context_dim, inner_dim, bias=False)
            self.to_v_ipa = Linear(context_dim, inner_dim, bias=False)
        self._kv = None  # (context tensor object, version, weight keys, k, vt, [k_ipa, vt_ipa])

    def _fused(self, dtype: torch.dtype) -> bool:
        """bf16 with a head dim the flash kernel takes: projections that share an input are ONE GEMM over row-stacked weights
        and V stays row-major (ops.attention_rows_v); otherwise (fp32-class) the separate / transposed-V projections."""
        return dtype == ops.HALF and (self.to_q.weight.shape[0] // self.heads) <= 160

    @staticmethod
    def _stack(*ws):
        hit = _qk_cache.get(ws)
        if hit is None:
            with torch.no_grad():
                hit = _qk_cache.put(ws, (), torch.cat([w.detach() for w in ws], 0).contiguous())
        return hit

    def _project_kv(self, context: torch.Tensor, dtype: torch.dtype):
        wk, wv = _eff(self, self.to_k.weight, "k"), _eff(self, self.to_v.weight, "v")
        fused = self._fused(dtype)
        wkeys = (_tkey(wk), _tkey(wv), dtype, fused)
        c = self._kv
        if c is not None and c[0] is context and c[1] == context._version and c[2] == wkeys:
            return c[3]
        ctx = context if context.dtype == dtype else context.to(dtype)
        ipa = None
        C_ = wk.shape[0]
        if self.ipa_num_tokens > 0:  # attention.py:623-627: last tokens go to the FaceID K/V projections
            end = ctx.shape[1] - self.ipa_num_tokens
            ctx, ipa_ctx = ctx[:, :end].contiguous(), ctx[:, end:].contiguous()
            if fused:
                kv = ops.linear(ipa_ctx, self._stack(self.to_k_ipa.weight, self.to_v_ipa.weight))
                ipa = (kv[..., :C_], kv[..., C_:], ipa_ctx.shape[1])
            else:
                ipa = (ops.linear(ipa_ctx, self.to_k_ipa.weight), ops.linear_transposed(ipa_ctx, self.to_v_ipa.weight), ipa_ctx.shape[1])
        if fused:
            kv = ops.linear(ctx, self._stack(wk, wv))   # [b, m, 2C]: K | V, both row-major
            out = (kv[..., :C_], kv[..., C_:], ctx.shape[1], ipa)
        else:
            out = (ops.linear(ctx, wk), ops.linear_transposed(ctx, wv), ctx.shape[1], ipa)
        self._kv = (context, context._version, wkeys, out)
        return out

    def forward(self, x, context=None, mask=None, residual=None, ln: Optional[nn.LayerNorm] = None, dup: bool = False, out_stats: bool = False):
        """`ln` (not in the reference's signature; passed by BasicTransformerBlock): the LayerNorm in front of this attention.
        Its application is this module's job then - fused into the projection launch where ops.ln_linear_ok allows.
        `dup` (cross-attention only): x and residual are ONE half of a CFG-doubled batch whose halves are identical up to here
        (ops.mark_cfg_dup); the context holds both halves.  The query projection runs once, q and the residual are duplicated.
        `out_stats`: the result (to_out + residual) feeds a LayerNorm - its row statistics come out of the to_out launch."""
        if exists(mask):
            raise NotImplementedError("attention masks are never passed on the SD path (attention.py:648-652)")
        fused = self._fused(x.dtype)

        def project(w, transposed_from=None):  # LayerNorm (if any) + the projection of x: one launch wherever a fused route exists
            return ops.ln_linear_auto(x, ln, w, transposed_from=transposed_from)

        def out_proj(o):
            return ops.linear(o, _eff(self, self.to_out[0].weight, "out"), self.to_out[0].bias, residual=residual, row_stats=out_stats)
        if context is None:
            wq, wk, wv = _eff(self, self.to_q.weight, "q"), _eff(self, self.to_k.weight, "k"), _eff(self, self.to_v.weight, "v")
            c = wq.shape[0]
            if self.ipa_num_tokens > 0:
                raise NotImplementedError("ipa_num_tokens > 0 needs a context (attention.py:623-627)")
            wqkv = self._stack(wq, wk, wv) if fused else None
            if fused and _SELF_QKV and ln is not None and not ops.ln_epi_ok(x, wqkv) and ops.ln_linear_ok(x, wqkv, transposed_from=2 * c):
                # 64x64 level: LayerNorm + Q | K | V in ONE launch whose V third is written transposed, so that the 4096-token
                # self-attention runs on the transposed-V flash kernel (the row-major-V variant is 10-20 % slower there)
                qk, vt = ops.ln_linear(x, ln.weight, ln.bias, ln.eps, wqkv, transposed_from=2 * c)
                return out_proj(ops.attention(qk[..., :c], qk[..., c:], vt, self.heads, x.shape[1], self.scale))
            if fused and _SELF_QKV and x.dim() == 3 and x.shape[1] % 64 == 0 and (c // self.heads) in (40, 64, 80) \
                    and ops.linear_transposed_ok(x, wqkv, 2 * c):
                # head dims with an LDS-DMA / pipelined attention kernel (they stage V^T): the V third of the fused projection comes
                # out transposed from the GEMM's own epilogue (LayerNorm: as an epilogue correction of the same launch when x carries
                # its producer's row statistics).  Device time in a graph, 8 x 1024 tokens, d 80: attention 43.6 -> 30.2 us;
                # SDXL 4 x 4096, d 64: 256 -> 205 us (tools/attn_vt_probe.py)
                qk, vt = project(wqkv, transposed_from=2 * c)
                return out_proj(ops.attention(qk[..., :c], qk[..., c:], vt, self.heads, x.shape[1], self.scale))
            if fused and _SELF_QKV:
                # self-attention: the three projections share their input -> ONE GEMM, q / k / v are column slices of its output
                qkv = project(wqkv)
                return out_proj(ops.attention_rows_v(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.heads, self.scale))
            if ln is not None:
                x, ln = ops.layer_norm(x, ln.weight, ln.bias, ln.eps), None
            qk = ops.linear(x, self._stack(wq, wk))
            q, k = qk[..., :c], qk[..., c:]
            vt = ops.linear_transposed(x, wv)
            return out_proj(ops.attention(q, k, vt, self.heads, x.shape[1], self.scale))
        else:
            q = project(_eff(self, self.to_q.weight, "q"))
            if dup:
                q = ops.dup_batch(q)
                residual = ops.dup_batch(residual) if residual is not None else None
            k, vt, nk, ipa = self._project_kv(context, x.dtype)
        att = (lambda kk, vv, n: ops.attention_rows_v(q, kk, vv, self.heads, self.scale)) if fused else \
              (lambda kk, vv, n: ops.attention(q, kk, vv, self.heads, n, self.scale))
        out = att(k, vt, nk)
        if ipa is not None:  # attention.py:660-681
            out_ipa = att(ipa[0], ipa[1], ipa[2])
            ops.axpby_(out, out_ipa, float(self.ipa_scale), 1.0)
        return out_proj(out)


CrossAttentionOriginal = CrossAttention
MemoryEfficientCrossAttention = CrossAttention


class BasicTransformerBlock(nn.Module):
    """attention.py:864-912.  All registry modes resolve to the HIP class; the reference's selection
    logic (:877-883) only ever chose among numerically equivalent implementations."""
    ATTENTION_MODES = {
        "softmax": CrossAttention,
        "softmax-xformers": CrossAttention,
        "softmax-original": CrossAttention,
        "softmax-hip": CrossAttention,
    }

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, lora_ranks: List[int] = None, lora_weights: List[float] = None, ipa_scale=1.0,
                 ipa_num_tokens=0):
        super().__init__()
        attn_cls = self.ATTENTION_MODES["softmax-hip"]
        self.disable_self_attn = disable_self_attn
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              context_dim=context_dim if self.disable_self_attn else None, lora_ranks=lora_ranks,
                              lora_weights=lora_weights)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff, lora_ranks=lora_ranks, lora_weights=lora_weights)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              lora_ranks=lora_ranks, lora_weights=lora_weights, ipa_scale=ipa_scale, ipa_num_tokens=ipa_num_tokens)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint  # gradient checkpointing is a no-op at inference (util.py:102-116)

    @staticmethod
    def _ln(norm: nn.LayerNorm, x):
        return ops.layer_norm(x, norm.weight, norm.bias, norm.eps)

    def forward(self, x, context=None, cfg_dup: bool = False):
        return self._forward(x, context, cfg_dup)

    def cfg_dup_ok(self) -> bool:
        """Can this block take ONE half of a CFG-doubled batch and hand back both (see CrossAttention.forward, `dup`)?  Its first
        attention must not read the conditioning."""
        return not self.disable_self_attn

    def _forward(self, x, context=None, cfg_dup: bool = False):
        # x = attn1(norm1(x)) + x; x = attn2(norm2(x), ctx) + x; x = ff(norm3(x)) + x (attention.py:908-912).  The LayerNorms are
        # handed to the consumers, which fuse them into their first GEMM launch where the kernel takes the shape
        # Each producer of a LayerNorm input (to_out + residual twice, net[2] + residual when another block follows) hands the consumer
        # its row statistics, so that the LayerNorm is an epilogue correction of the consuming GEMM (ops.linear, ln=)
        st = ops.ln_epi_wanted(x.shape[-1])
        x = self.attn1(x, context=context if self.disable_self_attn else None, residual=x, ln=self.norm1, out_stats=st)
        x = self.attn2(x, context=context, residual=x, ln=self.norm2, dup=cfg_dup and context is not None,
                       out_stats=ops.ln_epi_wanted(x.shape[-1], geglu=True))
        if cfg_dup and context is None:
            x = ops.dup_batch(x)
        x = self.ff(x, residual=x, ln=self.norm3, out_stats=st and getattr(self, "_crg_next_is_block", False))
        return x


class SpatialTransformer(nn.Module):
    """attention.py:915-1057.  `use_linear` is accepted and, as in the reference (:930-945), ignored:
    proj_in/proj_out stay 1x1 Conv2d parameters so LoRA weights keep their shapes."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True, lora_ranks: List[int] = None, lora_weights: List[float] = None,
                 ipa_scale=1.0, ipa_num_tokens=0):
        super().__init__()
        if exists(context_dim) and not isinstance(context_dim, list):
            context_dim = [context_dim]
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.lora_ranks = lora_ranks if lora_ranks is not None else []
        self.lora_weights = lora_weights if lora_weights is not None else [1.0] * len(self.lora_ranks)
        self.proj_in = Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        _lora_lists(self, "proj_in", in_channels, inner_dim, self.lora_ranks, conv=True)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, checkpoint=use_checkpoint, lora_ranks=self.lora_ranks,
                                  lora_weights=self.lora_weights, ipa_scale=ipa_scale, ipa_num_tokens=ipa_num_tokens)
            for d in range(depth)])
        self.proj_out = zero_module(Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))
        _lora_lists(self, "proj_out", inner_dim, in_channels, self.lora_ranks, conv=True)

    def cfg_dup_ok(self) -> bool:
        return len(self.transformer_blocks) > 0 and self.transformer_blocks[0].cfg_dup_ok()

    def forward(self, x, context=None, cfg_dup: bool = False):
        """`cfg_dup` (test cfg_dup_ok first): x is one half of a CFG-doubled batch with identical halves (ops.mark_cfg_dup), the
        context holds both; GroupNorm, proj_in and the first block's self-attention run once, the result has the full batch."""
        x = ops.to_channels_last(x)
        b, c, h, w = x.shape
        x_in = ops.tokens_of(x)
        xn = self.norm(x)
        t = ops.linear(ops.tokens_of(xn), _eff(self, self.proj_in.weight, "proj_in"), self.proj_in.bias,
                       row_stats=ops.ln_epi_wanted(self.proj_in.weight.shape[0]))  # feeds the first block's norm1
        nb = len(self.transformer_blocks)
        for i, block in enumerate(self.transformer_blocks):
            block._crg_next_is_block = i + 1 < nb  # its feed-forward output then feeds another LayerNorm
            if cfg_dup and i == 0:
                t = block(t, context=context, cfg_dup=True)
                x_in = ops.dup_batch(x_in)
                continue
            t = block(t, context=context)
        # the block's output feeds the next ResBlock's GroupNorm: its statistics come out of this launch's epilogue
        y = ops.linear(t, _eff(self, self.proj_out.weight, "proj_out"), self.proj_out.bias, residual=x_in, gn_hw=h * w)
        return ops.image_of_stats(y, h, w)
