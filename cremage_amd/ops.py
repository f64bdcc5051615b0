"""Torch-facing wrappers over the C-ABI (libcrg_hip.so).

PyTorch is used here only as the allocator / stream owner (`torch.empty`, `data_ptr()`,
`torch.cuda.current_stream()`); every arithmetic op on the hot path is a hand-written HIP kernel
reached through `cremage_amd._lib`.  There is deliberately no CPU or eager fallback: a CPU tensor,
an unsupported shape or a missing library raises.

Layout contract: image activations are 4-D tensors with logical shape [N, C, H, W] and
`torch.channels_last` strides (physically NHWC), so the reference's hooks that index channels on
dim 1 (ControlNet residual adds cldm.py:57-65, `th.cat(..., dim=1)` openaimodel.py:808) keep
working on them; token activations are contiguous [B, T, C].

Precision contract: bf16 activations -> CRG_PREC_BF16 kernels; fp32 activations -> CRG_PREC_BF16X3
(split-bf16, fp32-class) kernels.  Weights stay torch Parameters (any float dtype); their packed
bf16 images are cached keyed on (data_ptr, _version, dtype, device, kind) so that
`load_state_dict`, LoRA `setattr` (image_generator.py:408-453) and `.to()/.half()` invalidate them.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L

# The library's 16-bit type: bfloat16 (default) or, with CRG_HALF=f16, IEEE fp16 (the fp16-operand build of the same kernels).
# Code L.BF16 in a dtype argument means "the library's half type" in both builds.
HALF = torch.float16 if L.HALF_F16 else torch.bfloat16
_DT = {HALF: L.BF16, torch.float32: L.F32}
if not L.HALF_F16:
    _DT[torch.float16] = L.F16  # weight sources only (crg_pack_weight converts)


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise L.CrgError(f"unsupported dtype {t.dtype}")


def _act_dt(t: torch.Tensor) -> int:
    if t.dtype not in (HALF, torch.float32):
        raise L.CrgError(f"activations must be {HALF} or float32, got {t.dtype}")
    return _DT[t.dtype]


def _prec(t: torch.Tensor) -> int:
    return L.PREC_BF16 if t.dtype == HALF else L.PREC_BF16X3


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.CrgError("cremage_amd.ops: tensors must live on a HIP device (no CPU fallback exists)")


def _h(t: torch.Tensor):
    """crg context of the tensor's device.  Launches go to torch's CURRENT stream (`_st`), which belongs to the current device: a
    tensor on another device would have its pointers handed to the wrong GPU's queue, so that is an error, not a silent launch
    (multi-GPU runs are one process per GPU; a process driving several sets `torch.cuda.device(t.device)` around its calls)."""
    cur = torch.cuda.current_device()
    idx = t.device.index if t.device.index is not None else cur
    if idx != cur:
        raise L.CrgError(f"tensor lives on cuda:{idx} but the current device is cuda:{cur}: wrap the call in torch.cuda.device(...)")
    return L.ctx(idx)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---------------------------------------------------------------------------------- weight cache
class TensorKeyedCache:
    """Cache of values derived from torch tensors (packed weights, fp32 copies, merged LoRA weights).

    An entry is keyed on the IDENTITY of its source tensors (id + weakref) and stamped with their
    (data_ptr, _version, dtype, device): it is hit only while the very same tensor objects still hold
    the very same, unmodified storage.  Keying on data_ptr alone would be wrong - the caching allocator
    hands the address of a freed parameter to the next model's parameter of the same shape.  Entries
    die with their sources (weakref callbacks), so `load_state_dict` (in-place copy -> _version bump),
    LoRA `setattr` of fresh Parameters (image_generator.py:408-453) and `.to()/.half()` (new storage)
    all invalidate correctly."""

    def __init__(self, max_entries: int = 8192):
        self._d = {}
        self._max = max_entries

    @staticmethod
    def _stamp(t: torch.Tensor):
        return (t.data_ptr(), t._version, t.dtype, t.device, tuple(t.shape))

    def get(self, tensors, extra=()):
        key = (tuple(id(t) for t in tensors), extra)
        e = self._d.get(key)
        if e is None:
            return None
        refs, stamps, value = e
        for t, r, st in zip(tensors, refs, stamps):
            if r() is not t or self._stamp(t) != st:
                self._d.pop(key, None)
                return None
        return value

    def put(self, tensors, extra, value):
        import weakref
        if len(self._d) >= self._max:
            self.clear()
        key = (tuple(id(t) for t in tensors), extra)
        d = self._d

        def _drop(_ref, key=key, d=d):
            d.pop(key, None)

        self._d[key] = (tuple(weakref.ref(t, _drop) for t in tensors), tuple(self._stamp(t) for t in tensors), value)
        TensorKeyedCache.epoch += 1
        return value

    # bumped whenever ANY derived weight image is (re)built or a Parameter is registered: cremage_amd.graphs re-lists the module's
    # parameters when it moves (the weight stamp itself is the parameters' (data_ptr, _version))
    epoch = 0
    # bumped whenever ANY cache DROPS its values (clear(), or the overflow clear in put()): a captured hipGraph bakes in the
    # addresses of the packed / stacked / merged images it was warmed up with, and those are freed here while the parameters -
    # and with them the weight stamp - stay what they were.  The generation is part of the stamp, so such a graph is re-captured.
    generation = 0

    def clear(self):
        self._d.clear()
        TensorKeyedCache.generation += 1


_pack_cache = TensorKeyedCache()
_f32_cache = TensorKeyedCache()


def clear_weight_cache():
    _pack_cache.clear()
    _f32_cache.clear()


# ---------------------------------------------------------------------------------- GroupNorm statistics side channel
# A conv / linear whose output feeds a GroupNorm can write, next to its output, the per-(32-row block, channel) sum and sum of
# squares of what it stored (crg_conv_args.gn_stats / crg_gemm_args.gn_stats).  The buffer rides on the returned tensor object as
# `_crg_gn = (stats, version, rows per sample)`; group_norm() uses it when the tensor has not been written since (`_version`:
# ControlNet's in-place residual adds, cldm.py:57-65, invalidate it) and then skips its statistics pass over the tensor.
GN_STATS = __import__("os").environ.get("CRG_GN_STATS", "1") != "0"  # dev knob: 0 = every GroupNorm computes its own statistics
GN_STATS_MIN_HW = 512  # smaller images take the single-launch GroupNorm kernel, which reads the tensor once anyway
VAE_GN_STATS = __import__("os").environ.get("CRG_VAE_GN_STATS", "1") != "0"  # dev knob: 0 = the fp32-class convs emit no statistics
# Round 4: a producer may report a coarser granularity than 32 rows (crg_conv_args.gn_stats_rows: one partial per 256-pixel tile and channel
# from the staggered 3x3 conv); the tuple carries it as a fourth element and crg_groupnorm_pre, given tile partials on every input, folds
# them inside the normalising launch - the finalise launch disappears.  CRG_GN_TILE=0 (dev knob, A/B): 32-row partials everywhere.
GN_TILE = __import__("os").environ.get("CRG_GN_TILE", "1") != "0"


def _gn_stats_buffer(rows: int, cols: int, hw: Optional[int], in_dtype, out_dtype, device) -> Optional[torch.Tensor]:
    if not GN_STATS or not hw or hw % 32 or hw < GN_STATS_MIN_HW or rows % hw or cols % 8:
        return None
    if in_dtype != HALF or out_dtype not in (HALF, torch.float32):  # fp32 outputs: the fp32-class conv on split planes (VAE)
        return None
    return torch.empty((2, (rows + 31) // 32, cols), dtype=torch.float32, device=device)


def _gn_stats_of(t: torch.Tensor, hw: int) -> Optional[torch.Tensor]:
    g = getattr(t, "_crg_gn", None)
    if g is None or g[1] != t._version or g[2] != hw or g[0].shape[2] != t.shape[1]:
        return None
    return g[0]


def _gn_rows_of(t: Optional[torch.Tensor]) -> int:
    """Rows per partial of the statistics riding on t (32 unless the producer said otherwise)."""
    g = getattr(t, "_crg_gn", None) if t is not None else None
    return int(g[3]) if (g is not None and len(g) > 3) else 32


# ---------------------------------------------------------------------------------- LayerNorm as a GEMM epilogue
# nn.LayerNorm in front of a Linear (attention.py:900-912) never runs as a launch of its own when the LayerNorm's INPUT was written by a
# GEMM of this library: that producer hands over per-row (sum, sum of squares) partials (crg_gemm_args.row_stats, riding on the
# tensor as `_crg_ln = (stats, version)` like the GroupNorm channel above), and the consuming GEMM runs on the raw rows with W o gamma
# and corrects its accumulators in the epilogue (crg_gemm_args.ln_stats).  Any width K; the K = 320 row-resident kernel (crg_ln_gemm)
# stays the route of the 64x64 level unless CRG_LN_EPI_320 says otherwise.
_env = __import__("os").environ
LN_EPI = _env.get("CRG_LN_EPI", "1") != "0"          # dev knob (A/B): 0 = stand-alone layernorm launches as in round 3
# K = 320 (the 64x64 level, where crg_ln_gemm exists): 0 = row-resident kernel for every consumer (round 3), 1 (default) = the plain
# projections on the epilogue route (device time in a graph, tools/lnepi_probe.py: LN + Q | K | V 43.4 -> 40.6 us, LN + to_q 19.1 -> 16.6),
# the GEGLU projection stays row-resident (76.8 us against 95.7 on the epilogue route), 2 = the GEGLU projection as well
LN_EPI_320 = int(_env.get("CRG_LN_EPI_320", "1"))
LN_EPI_MASK = int(_env.get("CRG_LN_EPI_MASK", "3"))   # dev knob (A/B): bit 0 = plain consumers (Q | K | V, to_q) on the epilogue route, bit 1 = GEGLU consumers


def row_stats_parts(n: int) -> int:
    """Column partials per row the producer writes for an N-wide output (include/crg_hip.h: 2 * ceil(N / tile), tile 160 | 128)."""
    bn = 160 if n % 160 == 0 else 128
    return 2 * ((n + bn - 1) // bn)


def ln_epi_wanted(width: int, geglu: bool = False) -> bool:
    """Should the producer of a LayerNorm input of this width emit row statistics - would its consumer (the GEGLU projection if `geglu`,
    else a plain projection) use them?"""
    return LN_EPI and bool(LN_EPI_MASK & (2 if geglu else 1)) and row_stats_parts(width) <= 16 and (width != 320 or LN_EPI_320 >= (2 if geglu else 1))


def _ln_stats_of(x: torch.Tensor) -> Optional[torch.Tensor]:
    g = getattr(x, "_crg_ln", None)
    K = x.shape[-1]
    if g is None or g[1] != x._version or g[0].shape[0] != x.numel() // K or g[0].shape[1] != row_stats_parts(K) or g[0].shape[1] > 16:
        return None
    return g[0]


def ln_epi_ok(x: torch.Tensor, weight: torch.Tensor, act: Optional[str] = None) -> bool:
    """Can `linear(x, weight, ..., ln=...)` run - does x carry valid row statistics and does the GEMM take the shape?"""
    if not (LN_EPI and x.is_cuda and x.dtype == HALF and x.is_contiguous() and act in (None, "geglu")) or not (LN_EPI_MASK & (2 if act == "geglu" else 1)):
        return False
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    if weight[0].numel() != K or K % 8 or N % (32 if act == "geglu" else 8):
        return False
    if K == 320 and LN_EPI_320 < (2 if act == "geglu" else 1):
        return False
    return _ln_stats_of(x) is not None


def packed_ln_weight(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor], geglu: bool):
    """(W o gamma in the library's half type [N, K], column sums of that rounded image fp32 [N], folded bias W beta + b fp32 [N]) -
    crg_pack_ln_weight; cached on the identity + version of all four sources (LoRA-merged weights arrive as their own tensors)."""
    _need_cuda(w, gamma, beta, bias)
    srcs = (w, gamma, beta) + ((bias,) if bias is not None else ())
    r = _pack_cache.get(srcs, ("ln", geglu, bias is not None))
    if r is not None:
        return r
    src = w.detach().reshape(w.shape[0], -1).contiguous()
    if src.dtype not in _DT:
        src = src.float()
    n_out, n_in = src.shape
    dw = torch.empty((n_out, n_in), dtype=HALF, device=w.device)
    ds = torch.empty((n_out,), dtype=torch.float32, device=w.device)
    db = torch.empty((n_out,), dtype=torch.float32, device=w.device)
    g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    bb = bias.detach().float().contiguous() if bias is not None else None
    h = _h(w)
    L.check(L.load().crg_pack_ln_weight(h, _st(), _p(src), _dt(src), _p(g32), _p(b32), _p(bb), L.PACK_GEGLU if geglu else L.PACK_LINEAR, n_out, n_in,
                                        _p(dw), _p(ds), _p(db)), h, "crg_pack_ln_weight")
    return _pack_cache.put(srcs, ("ln", geglu, bias is not None), (dw, ds, db))


def _written(t: torch.Tensor) -> torch.Tensor:
    """A kernel of this library just wrote `t` in place through its raw pointer: tell PyTorch (bump `_version`), so that everything keyed
    on the version - the GroupNorm statistics riding on the tensor (`_crg_gn`), the cross-attention K / V cache, the weight caches -
    sees the write as it sees a torch in-place op.  A raw write through a VIEW bumps the shared counter of the base as well."""
    torch.autograd.graph.increment_version(t)
    for a in ("_crg_gn", "_crg_ln"):
        if hasattr(t, a):
            delattr(t, a)
    return t


def f32_vec(v: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """fp32 contiguous view/copy of a bias / norm gain (kernels take fp32 vectors)."""
    if v is None:
        return None
    if v.dtype == torch.float32 and v.is_contiguous():
        return v.detach()
    r = _f32_cache.get((v,))
    if r is None:
        r = _f32_cache.put((v,), (), v.detach().float().contiguous())
    return r


def packed_weight(w: torch.Tensor, kind: int, split: bool) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """bf16 [N, K] image of a Linear / conv weight (+ bf16 residual plane when `split`)."""
    _need_cuda(w)
    if kind == L.PACK_LINEAR and not split and w.dtype == HALF and w.is_contiguous():
        return w.detach().reshape(w.shape[0], -1), None  # zero-copy: already the packed image
    r = _pack_cache.get((w,), (kind, split))
    if r is not None:
        return r
    src = w.detach().contiguous()
    if src.dtype not in _DT:
        src = src.float()  # e.g. bfloat16 parameters handed to the fp16-operand build
    n_out = src.shape[0]
    if kind == L.PACK_CONV:
        n_in, ks = src.shape[1], src.shape[2]
        cols = n_in * ks * ks
    else:
        n_in, ks = src[0].numel(), 1
        cols = n_in
    hi = torch.empty((n_out, cols), dtype=HALF, device=w.device)
    lo = torch.empty_like(hi) if split else None
    h = _h(w)
    L.check(L.load().crg_pack_weight(h, _st(), _p(src), _dt(src), kind, n_out, n_in, ks, _p(hi), _p(lo)), h, "crg_pack_weight")
    return _pack_cache.put((w,), (kind, split), (hi, lo))


# ---------------------------------------------------------------------------------- MX planes (CRG_PREC_F16MX)
# The fp32-class 3x3 conv in two matrix passes' worth of cycles instead of three: operands as a fp16 plane plus a plane of e4m3 pairs
# (include/crg_hip.h, crg_split_mx).  Activations behind GroupNorm + SiLU are O(1): fixed power-of-two scales (hi8: x * 2^4, saturating at
# |x| = 28 - the cross terms lose precision there, nothing else; lo8: (x - half(x)) * 2^15).  Weights: scales from the tensor's maximum at
# pack time.  Correct and validated (tests/test_hip_ops.py::test_conv_mx) but measured 7-18 % slower than bf16 x 3 in round 4 (gemm_conv.hip,
# conv3_rowhalo_kernel<.., MX>: status note), so the VAE uses it only under CRG_VAE_MX=1.
VAE_MX = __import__("os").environ.get("CRG_VAE_MX", "0") != "0"
MX_X_LOG2 = (4, 15)


def packed_weight_mx(w: torch.Tensor):
    """(w16 fp16 [Cout, 9 Cin], w8 uint8 [Cout, 9 Cin / 64, 128], hi_log2, lo_log2) of a 3x3 conv weight (crg_pack_weight_mx)."""
    _need_cuda(w)
    r = _pack_cache.get((w,), ("mx",))
    if r is not None:
        return r
    src = w.detach().contiguous()
    if src.dtype not in _DT:
        src = src.float()
    n_out, n_in = src.shape[0], src.shape[1]
    amax = float(src.detach().abs().max().item())  # pack time only (outside graph capture: the warm-up calls pack)
    import math
    hi = int(math.floor(math.log2(448.0 / max(amax, 1e-30)))) if amax > 0 else 0
    hi = max(-60, min(60, hi))
    lo = hi + 11
    w16 = torch.empty((n_out, 9 * n_in), dtype=torch.float16, device=w.device)
    w8 = torch.empty((n_out, 9 * n_in // 64, 128), dtype=torch.uint8, device=w.device)
    h = _h(w)
    L.check(L.load().crg_pack_weight_mx(h, _st(), _p(src), _dt(src), n_out, n_in, _p(w16), _p(w8), hi, lo), h, "crg_pack_weight_mx")
    return _pack_cache.put((w,), ("mx",), (w16, w8, hi, lo))


def _empty_mx(n, c, hh, ww, device):
    x16 = torch.empty((n, hh, ww, c), dtype=torch.float16, device=device).permute(0, 3, 1, 2)
    x8 = torch.empty((n, hh, ww, 2 * c), dtype=torch.uint8, device=device)
    return x16, x8


def split_mx(x: torch.Tensor):
    """fp32 channels-last image -> MX planes (x16 fp16 image, x8 uint8 [N, H, W, 2 C]) for conv2d(x16, ..., x_mx=x8); C % 64 == 0."""
    _need_cuda(x)
    x = to_channels_last(x)
    n, c, hh, ww = x.shape
    if x.dtype != torch.float32 or c % 64:
        raise L.CrgError("split_mx: fp32 image with C % 64 == 0 expected")
    x16, x8 = _empty_mx(n, c, hh, ww, x.device)
    h = _h(x)
    L.check(L.load().crg_split_mx(h, _st(), _p(x), _p(x16), _p(x8), n * hh * ww, c, MX_X_LOG2[0], MX_X_LOG2[1]), h, "crg_split_mx")
    return x16, x8


def packed_geglu_bias(b: torch.Tensor) -> torch.Tensor:
    r = _pack_cache.get((b,), ("geglu_bias",))
    if r is None:
        src = b.detach().float().contiguous()
        r = torch.empty_like(src)
        h = _h(b)
        L.check(L.load().crg_pack_geglu_bias(h, _st(), _p(src), src.numel(), _p(r)), h, "crg_pack_geglu_bias")
        _pack_cache.put((b,), ("geglu_bias",), r)
    return r


# ---------------------------------------------------------------------------------- layout helpers
def to_channels_last(x: torch.Tensor) -> torch.Tensor:
    """No-op for tensors that already have NHWC strides; otherwise one boundary transpose kernel."""
    if x.dim() != 4:
        raise L.CrgError(f"expected a 4-D image tensor, got shape {tuple(x.shape)}")
    if x.permute(0, 2, 3, 1).is_contiguous():
        return x
    return nchw_to_nhwc(x.contiguous(), x.dtype)


def empty_image(n, c, h, w, dtype, device) -> torch.Tensor:
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def tokens_of(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] channels-last image -> [N, H*W, C] contiguous view (no copy)."""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n, h * w, c)


def image_of(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """[N, H*W, C] tokens -> [N, C, H, W] channels-last view (no copy)."""
    n, hw, c = t.shape
    return t.reshape(n, h, w, c).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------- CFG-shared prefix
# Classifier-free guidance by batch doubling (ldm_wrapper_for_k_diffusion.py:67-93) hands the UNet cat([x] * 2) with cat([t] * 2): until
# the first cross-attention reads the (different) conditioning of the two halves, both halves compute the same values.  The sampler
# wrapper of this package marks such an input (never inferred from data); the UNet then runs that prefix on one half and duplicates
# the activations where the halves start to differ.  CRG_CFG_SHARE=0 (dev knob, A/B) runs the whole batch as before.
CFG_SHARE = __import__("os").environ.get("CRG_CFG_SHARE", "1") != "0"


def mark_cfg_dup(x: torch.Tensor) -> torch.Tensor:
    """Caller's promise: x == cat([h, h]) along the batch dim and so are the timesteps that go with it."""
    x._crg_cfg_dup = True
    return x


def dup_batch(t: torch.Tensor) -> torch.Tensor:
    """cat([t, t], dim 0) of a token tensor [B, T, C] or of a channels-last image (strides kept); one copy kernel.  The GroupNorm
    statistics riding on an image (32-row blocks, sample-major) are duplicated with it."""
    if t.dim() == 4:
        v = t.permute(0, 2, 3, 1)
        if not v.is_contiguous():
            v = to_channels_last(t).permute(0, 2, 3, 1)
        out = torch.cat([v, v], 0).permute(0, 3, 1, 2)
        g = getattr(t, "_crg_gn", None)
        if g is not None and g[1] == t._version:
            rows = g[3] if len(g) > 3 else 32
            if rows == 32:
                st = torch.cat([g[0], g[0]], 1)
            else:  # tile partials: the first M / rows partials of each plane, sample-major; the plane stride stays that of the 32-row layout
                used = g[0].shape[1] * 32 // rows
                st = torch.empty((2, 2 * g[0].shape[1], g[0].shape[2]), dtype=g[0].dtype, device=g[0].device)
                st[:, :used] = g[0][:, :used]
                st[:, used:2 * used] = g[0][:, :used]
            out._crg_gn = (st, out._version, g[2], rows)
        return out
    t = t.contiguous()
    return torch.cat([t, t], 0)


# ---------------------------------------------------------------------------------- hoisted timestep work
# What a UNet call computes from the timestep alone (sinusoid -> time_embed MLP -> SiLU -> the ResBlocks' emb_layers, openaimodel.py:793-796,
# :222-228) is known for the whole schedule before the first step.  A sampler of this package computes it once (`UNetModel.time_rows`) and
# hands each step its rows on the timestep tensor, which the reference's containers pass on untouched (ddpm.py:1034, :1517-1519) like the
# CFG mark above.  The rows name the network they were computed with: a ControlNet that receives the same timesteps ignores them.
# CRG_TIME_ROWS=0 (dev knob, A/B) computes the embedding inside every call as before.
TIME_ROWS = __import__("os").environ.get("CRG_TIME_ROWS", "1") != "0"


def attach_time_rows(timesteps: torch.Tensor, rows: torch.Tensor, owner) -> torch.Tensor:
    """Caller's promise: rows == owner.time_rows(timesteps[None])[0] ([N, sum Cout] fp32 of `owner`, a UNetModel)."""
    timesteps._crg_time_rows = (owner, rows)
    return timesteps


def time_rows_of(timesteps, owner):
    """The rows attached for `owner`, or None."""
    tr = getattr(timesteps, "_crg_time_rows", None)
    return tr[1] if (TIME_ROWS and tr is not None and tr[0] is owner) else None


def nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Contiguous NCHW tensor -> channels-last tensor of `dtype` (one transpose+cast kernel)."""
    _need_cuda(x)
    n, c, hh, ww = x.shape
    if x.dtype not in (torch.float32, HALF):
        x = x.float()
    x = x.contiguous()
    y = empty_image(n, c, hh, ww, dtype, x.device)
    h = _h(x)
    L.check(L.load().crg_nchw_to_nhwc(h, _st(), _p(x), _p(y), n, c, hh * ww, _act_dt(x), _DT[dtype]), h, "crg_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Channels-last tensor -> contiguous NCHW tensor of `dtype`."""
    _need_cuda(x)
    x = to_channels_last(x)
    n, c, hh, ww = x.shape
    y = torch.empty((n, c, hh, ww), dtype=dtype, device=x.device)
    h = _h(x)
    L.check(L.load().crg_nhwc_to_nchw(h, _st(), _p(x), _p(y), n, c, hh * ww, _act_dt(x), _DT[dtype]), h, "crg_nhwc_to_nchw")
    return y


def affine_cast(x: torch.Tensor, a: float, b: float, dtype: torch.dtype, lo: float = float("-inf"), hi: float = float("inf")):
    """y = clamp(a*x + b, lo, hi) cast to `dtype`, preserving strides (dense tensors only)."""
    _need_cuda(x)
    y = torch.empty_strided(x.shape, x.stride(), dtype=dtype, device=x.device)
    h = _h(x)
    L.check(L.load().crg_affine_cast(h, _st(), _p(x), _p(y), x.numel(), a, b, lo, hi, _act_dt(x), _DT[dtype]), h, "crg_affine_cast")
    return y


# ---------------------------------------------------------------------------------- norms
def group_norm(x: torch.Tensor, weight, bias, groups: int, eps: float, silu: bool = False, x2: Optional[torch.Tensor] = None,
               split: bool = False):
    """GroupNorm(+SiLU) over a channels-last image; `x2` = second half of a virtual channel concat.  With `split` (fp32
    inputs only) the result is returned as the pair of bf16 planes (hi, lo) that conv2d(hi, ..., x_lo=lo) consumes."""
    _need_cuda(x, weight, bias, x2)
    x = to_channels_last(x)
    n, c1, hh, ww = x.shape
    c = c1
    if x2 is not None:
        x2 = to_channels_last(x2)
        if x2.shape[0] != n or x2.shape[2:] != x.shape[2:] or x2.dtype != x.dtype:
            raise L.CrgError("group_norm: concat halves disagree in shape/dtype")
        c = c1 + x2.shape[1]
    if weight.numel() != c:
        raise L.CrgError(f"group_norm: {weight.numel()} gains for {c} channels")
    h = _h(x)
    if split == "mx":  # MX planes for the fp32-class 3x3 conv (CRG_PREC_F16MX)
        if x.dtype != torch.float32 or x2 is not None or c % 64:
            raise L.CrgError("group_norm(split='mx'): one fp32 input with C % 64 == 0 expected")
        y16, y8 = _empty_mx(n, c, hh, ww, x.device)
        st1 = _gn_stats_of(x, hh * ww)
        L.check(L.load().crg_groupnorm_mx(h, _st(), _p(x), _p(st1), _p(f32_vec(weight)), _p(f32_vec(bias)), _p(y16), _p(y8), n, hh * ww, c, groups, eps,
                                          int(silu), MX_X_LOG2[0], MX_X_LOG2[1]), h, "crg_groupnorm_mx")
        return y16, y8
    if split:
        if x.dtype != torch.float32:
            raise L.CrgError("group_norm(split=True) is the fp32-class path: fp32 input expected")
        hi, lo = empty_image(n, c, hh, ww, HALF, x.device), empty_image(n, c, hh, ww, HALF, x.device)
        st1 = _gn_stats_of(x, hh * ww) if x2 is None else None
        if st1 is not None:  # statistics handed over by the fp32-class conv that produced x: no statistics pass over the fp32 tensor
            L.check(L.load().crg_groupnorm_pre_split(h, _st(), _p(x), _p(st1), _p(f32_vec(weight)), _p(f32_vec(bias)), _p(hi), _p(lo), n, hh * ww, c,
                                                     groups, eps, int(silu)), h, "crg_groupnorm_pre_split")
            return hi, lo
        L.check(L.load().crg_groupnorm_split(h, _st(), _p(x), _p(x2), c1, _p(f32_vec(weight)), _p(f32_vec(bias)), _p(hi), _p(lo), n, hh * ww, c,
                                             groups, eps, int(silu)), h, "crg_groupnorm_split")
        return hi, lo
    y = empty_image(n, c, hh, ww, x.dtype, x.device)
    if x.dtype == HALF:
        # statistics handed over by the producer(s) of x (and x2): no statistics pass over the tensor
        st1 = _gn_stats_of(x, hh * ww)
        st2 = _gn_stats_of(x2, hh * ww) if x2 is not None else None
        if st1 is not None and (x2 is None or st2 is not None):
            L.check(L.load().crg_groupnorm_pre(h, _st(), _p(x), _p(x2), c1, _p(st1), _p(st2), _p(f32_vec(weight)), _p(f32_vec(bias)), _p(y),
                                               n, hh * ww, c, groups, eps, int(silu), L.BF16, _gn_rows_of(x), _gn_rows_of(x2)), h, "crg_groupnorm_pre")
            return y
    L.check(L.load().crg_groupnorm(h, _st(), _p(x), _p(x2), c1, _p(f32_vec(weight)), _p(f32_vec(bias)), _p(y), n, hh * ww, c,
                                   groups, eps, int(silu), _act_dt(x)), h, "crg_groupnorm")
    return y


def layer_norm(x: torch.Tensor, weight, bias, eps: float = 1e-5):
    _need_cuda(x, weight, bias)
    x = x.contiguous()
    dim = x.shape[-1]
    y = torch.empty_like(x)
    h = _h(x)
    L.check(L.load().crg_layernorm(h, _st(), _p(x), _p(f32_vec(weight)), _p(f32_vec(bias)), _p(y), x.numel() // dim, dim, eps,
                                   _act_dt(x)), h, "crg_layernorm")
    return y


def softmax_rows_(x: torch.Tensor, cols: int, scale: float):
    """In-place softmax(x[:, :cols] * scale) over the last dim of a 2-D view with row stride x.shape[-1]."""
    _need_cuda(x)
    ld = x.shape[-1]
    rows = x.numel() // ld
    h = _h(x)
    L.check(L.load().crg_softmax_rows(h, _st(), _p(x), _p(x), rows, cols, ld, scale, _act_dt(x)), h, "crg_softmax_rows")
    return _written(x)


# ---------------------------------------------------------------------------------- GEMM family
def _gemm(h, **kw):
    a = L.GemmArgs()
    for k, v in kw.items():
        setattr(a, k, v)
    L.check(L.load().crg_gemm(h, _st(), C.byref(a)), h, "crg_gemm")


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           act: Optional[str] = None, out_dtype: Optional[torch.dtype] = None, gn_hw: Optional[int] = None,
           transposed_from: Optional[int] = None, row_stats: bool = False, ln=None):
    """y = act(x @ weight^T + bias) + residual over the last dim of x.
    weight: [N, K] (nn.Linear) or [N, K, 1, 1] (1x1 conv).  act: None | 'silu' | 'geglu'.
    gn_hw: y (as an image of gn_hw tokens per sample) feeds a GroupNorm - emit its statistics side channel (bf16 only, see above).
    transposed_from = n0 (test linear_transposed_ok first; x must be [B, T, K]): output columns >= n0 are returned as a SECOND tensor
    [B, N - n0, ld], ld = roundup(T, 8), transposed per sample - the V^T operand of `attention` out of the same launch as Q | K; the
    first tensor then has n0 columns.
    row_stats: y feeds an nn.LayerNorm - emit its row-statistics side channel where the path supports it (bf16, plain epilogue).
    ln = (gamma, beta, eps) (test ln_epi_ok first): y = act(LayerNorm(x) @ weight^T + bias) with the LayerNorm carried as an epilogue
    correction on the statistics x brought along; x itself is read raw, the normalised tensor never exists."""
    _need_cuda(x, weight, bias, residual)
    ln_stats = None
    if ln is not None:
        if not ln_epi_ok(x, weight, act) or residual is not None or gn_hw is not None or out_dtype not in (None, HALF):
            raise L.CrgError("linear: ln= outside its domain (test ln_epi_ok; no residual / GroupNorm statistics / fp32 output)")
        ln_stats = _ln_stats_of(x)
    x = x.contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    if weight[0].numel() != K:
        raise L.CrgError(f"linear: weight {tuple(weight.shape)} does not match input width {K}")
    split = x.dtype == torch.float32
    out_dtype = out_dtype or x.dtype
    if split and out_dtype != torch.float32:
        raise L.CrgError("linear: fp32 (BF16X3) inputs produce fp32 outputs")
    geglu = act == "geglu"
    ln_s = None
    if ln is not None:
        hi, ln_s, ln_b = packed_ln_weight(weight, ln[0], ln[1], bias, geglu)
        lo = None
    else:
        hi, lo = packed_weight(weight, L.PACK_GEGLU if geglu else L.PACK_LINEAR, split)
    n_out = N // 2 if geglu else N
    vt, vt_tokens, vt_ld = None, 0, 0
    if transposed_from is not None:
        if not linear_transposed_ok(x, weight, transposed_from) or act is not None or residual is not None or gn_hw is not None:
            raise L.CrgError("linear: transposed_from outside its domain (test linear_transposed_ok; no activation / residual / statistics)")
        n_out, vt_tokens = transposed_from, x.shape[1]
        vt_ld = (vt_tokens + 7) // 8 * 8
        vt = torch.empty((x.shape[0], N - transposed_from, vt_ld), dtype=x.dtype, device=x.device)
        if vt_ld != vt_tokens:
            vt[:, :, vt_tokens:].zero_()
    y = torch.empty(x.shape[:-1] + (n_out,), dtype=out_dtype, device=x.device)
    b = None
    if ln is not None:
        b = ln_b  # W beta + bias, already in the packed row order
    elif bias is not None:
        b = packed_geglu_bias(bias) if geglu else f32_vec(bias)
    if residual is not None:
        residual = residual.contiguous()
        if residual.shape != y.shape or residual.dtype != y.dtype:
            raise L.CrgError("linear: residual must match the output in shape and dtype")
    h = _h(x)
    stats = _gn_stats_buffer(M, N, gn_hw, x.dtype, out_dtype, x.device) if not geglu else None
    rs = None
    if row_stats and LN_EPI and stats is None and ln is None and act is None and vt is None and x.dtype == HALF and out_dtype == HALF and N % 8 == 0 \
            and row_stats_parts(N) <= 16:
        rs = torch.empty((M, row_stats_parts(N), 2), dtype=torch.float32, device=x.device)
    _gemm(h, a=x.data_ptr(), lda=K, a_bstride=0, w=hi.data_ptr(), ldw=K, w_bstride=0, w_lo=lo.data_ptr() if lo is not None else None,
          bias=b.data_ptr() if b is not None else None, bias_mode=L.BIAS_COL if b is not None else L.BIAS_NONE,
          residual=residual.data_ptr() if residual is not None else None, ldr=n_out, r_bstride=0,
          y=y.data_ptr(), ldy=n_out, y_bstride=0, M=M, N=N, K=K, batch=1,
          epilogue={None: L.EPI_NONE, "silu": L.EPI_SILU, "geglu": L.EPI_GEGLU}[act],
          a_dtype=_act_dt(x), y_dtype=_DT[out_dtype], prec=_prec(x), a_is_weight=0, a_lo=None,
          gn_stats=stats.data_ptr() if stats is not None else None,
          vt=vt.data_ptr() if vt is not None else None, vt_n0=transposed_from or 0, vt_tokens=vt_tokens, vt_ld=vt_ld,
          row_stats=rs.data_ptr() if rs is not None else None, row_stats_parts=rs.shape[1] if rs is not None else 0,
          ln_stats=ln_stats.data_ptr() if ln_stats is not None else None, ln_parts=ln_stats.shape[1] if ln_stats is not None else 0,
          ln_colsum=ln_s.data_ptr() if ln_s is not None else None, ln_eps=float(ln[2]) if ln is not None else 0.0)
    if stats is not None:
        y._crg_gn_pending = stats  # the caller that shapes y into an image attaches it (image_of_stats)
    if rs is not None:
        y._crg_ln = (rs, y._version)
    return (y, vt) if vt is not None else y


def ln_linear_auto(x: torch.Tensor, ln, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
                   transposed_from: Optional[int] = None):
    """act(LayerNorm(x) @ weight^T + bias) by the cheapest route the shapes allow (`ln`: an nn.LayerNorm-like module with weight / bias /
    eps, or None for no norm): the epilogue correction when x carries its producer's row statistics, the row-resident kernel for
    K = 320, else layer_norm + linear.  `transposed_from`: the caller has tested linear_transposed_ok (or ln_linear_ok) for it."""
    if ln is None:
        return linear(x, weight, bias, act=act, transposed_from=transposed_from)
    if ln_epi_ok(x, weight, act) and (transposed_from is None or linear_transposed_ok(x, weight, transposed_from)):
        return linear(x, weight, bias, act=act, transposed_from=transposed_from, ln=(ln.weight, ln.bias, ln.eps))
    if ln_linear_ok(x, weight, act, transposed_from):
        return ln_linear(x, ln.weight, ln.bias, ln.eps, weight, bias, act=act, transposed_from=transposed_from)
    return linear(layer_norm(x, ln.weight, ln.bias, ln.eps), weight, bias, act=act, transposed_from=transposed_from)


def linear_transposed_ok(x: torch.Tensor, weight: torch.Tensor, n0: int) -> bool:
    """Shapes `linear(..., transposed_from=n0)` takes: [B, T, K] tokens in the library's half type, N and n0 on tile boundaries of the
    paired epilogue (tile = 160 columns when N % 160 == 0, else 128), K below crg_gemm's split-K rule (the transposed range comes out of
    the GEMM's own epilogue, not out of a reduce pass)."""
    if not (x.is_cuda and x.dim() == 3 and x.dtype == HALF):
        return False
    N, K = weight.shape[0], weight[0].numel()
    bn = 160 if N % 160 == 0 else 128
    return K == x.shape[-1] and K % 8 == 0 and K < 24 * 64 and N % 8 == 0 and 0 < n0 < N and n0 % bn == 0 and N > 32


def ln_linear_ok(x: torch.Tensor, weight: torch.Tensor, act: Optional[str] = None, transposed_from: Optional[int] = None) -> bool:
    """Every precondition crg_ln_gemm enforces (the row-resident LayerNorm + GEMM kernel), so that callers can route on this
    predicate ALONE and fall back to layer_norm + linear otherwise: bf16 tokens of width 320; N a multiple of 8 (GEGLU: of 32,
    packed value / gate groups); a transposed column range starts on a tile boundary (tile = 160 columns when N % 160 == 0, else
    128) of a [B, T, K] input without activation; x, w and y (and the V^T output) each below 2 GiB."""
    if not (x.is_cuda and x.dtype == HALF and x.shape[-1] == 320 and weight[0].numel() == 320):
        return False
    N = weight.shape[0]
    M = x.numel() // 320
    geglu = act == "geglu"
    if act not in (None, "geglu") or N % (32 if geglu else 8):
        return False
    n_out = N // 2 if geglu else N
    if transposed_from is not None:
        bn = 160 if N % 160 == 0 else 128
        if geglu or x.dim() != 3 or not (0 < transposed_from < N) or transposed_from % bn:
            return False
        ld = (x.shape[1] + 7) // 8 * 8
        if x.shape[0] * (N - transposed_from) * ld * 2 >= (1 << 31):
            return False
        n_out = transposed_from
    return max(M * 320, N * 320, M * n_out) * 2 < (1 << 31)


def ln_linear(x: torch.Tensor, ln_weight, ln_bias, eps: float, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
              act: Optional[str] = None, transposed_from: Optional[int] = None, residual: Optional[torch.Tensor] = None):
    """y = act(LayerNorm(x) @ weight^T + bias) in ONE launch (crg_ln_gemm): nn.LayerNorm + the Linear behind it.
    act: None | 'geglu'.  Falls back to nothing - callers test ln_linear_ok() and otherwise run layer_norm + linear.
    transposed_from = n0 (x must be [B, T, K]): output columns >= n0 are returned as a second tensor [B, N - n0, ld] with
    ld = roundup(T, 8), i.e. transposed per sample (the V^T operand of `attention`); the first tensor then has n0 columns.
    ln_weight = ln_bias = None: no LayerNorm (row-resident plain GEMM); `residual` ([.., N], bf16) is added after the bias."""
    _need_cuda(x, ln_weight, ln_bias, weight, bias)
    if not ln_linear_ok(x, weight, act, transposed_from):
        raise L.CrgError("ln_linear: shape outside the row-resident kernel's domain (test ln_linear_ok, else layer_norm + linear)")
    x = x.contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    geglu = act == "geglu"
    if act not in (None, "geglu"):
        raise L.CrgError(f"ln_linear: activation {act!r} unsupported")
    hi, _ = packed_weight(weight, L.PACK_GEGLU if geglu else L.PACK_LINEAR, False)
    n_out = N // 2 if geglu else (transposed_from if transposed_from is not None else N)
    y = torch.empty(x.shape[:-1] + (n_out,), dtype=x.dtype, device=x.device)
    b = None
    if bias is not None:
        b = packed_geglu_bias(bias) if geglu else f32_vec(bias)
    vt, tokens, ld = None, 0, 0
    if transposed_from is not None:
        if x.dim() != 3 or geglu:
            raise L.CrgError("ln_linear: transposed_from needs [B, T, K] tokens and no activation")
        tokens = x.shape[1]
        ld = (tokens + 7) // 8 * 8
        vt = torch.empty((x.shape[0], N - transposed_from, ld), dtype=x.dtype, device=x.device)
        if ld != tokens:
            vt[:, :, tokens:].zero_()
    if residual is not None:
        residual = residual.contiguous()
        if residual.shape != y.shape or residual.dtype != y.dtype:
            raise L.CrgError("ln_linear: residual must match the output in shape and dtype")
    a = L.LnGemmArgs(x=x.data_ptr(), ldx=K, gamma=_p(f32_vec(ln_weight)).value, beta=_p(f32_vec(ln_bias)).value, eps=float(eps),
                     w=hi.data_ptr(), ldw=K, bias=_p(b).value, y=y.data_ptr(), ldy=n_out, M=M, N=N, K=K,
                     epilogue=L.EPI_GEGLU if geglu else L.EPI_NONE, vt=_p(vt).value, vt_n0=transposed_from or 0, vt_tokens=tokens, vt_ld=ld,
                     residual=_p(residual).value, ldr=n_out)
    h = _h(x)
    L.check(L.load().crg_ln_gemm(h, _st(), C.byref(a)), h, "crg_ln_gemm")
    return (y, vt) if transposed_from is not None else y


def linear_transposed(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = weight @ x[b]^T (+ bias per row): [B, T, K] tokens -> [B, N, ld] with ld = roundup(T, 8).
    Emits the V projection already transposed ([channel][key]) for crg_attention; pad columns are
    never read unmasked (the kernel masks keys >= T)."""
    _need_cuda(x, weight, bias)
    x = x.contiguous()
    B, T, K = x.shape
    N = weight.shape[0]
    split = x.dtype == torch.float32
    hi, lo = packed_weight(weight, L.PACK_LINEAR, split)
    ld = (T + 7) // 8 * 8
    y = torch.empty((B, N, ld), dtype=x.dtype, device=x.device)
    if ld != T:
        y[:, :, T:].zero_()
    b = f32_vec(bias)
    h = _h(x)
    if not split:
        # A := weight (bf16 packed), W := activations (bf16 rows are already the "packed" layout)
        _gemm(h, a=hi.data_ptr(), lda=K, a_bstride=0, w=x.data_ptr(), ldw=K, w_bstride=T * K, w_lo=None,
              bias=b.data_ptr() if b is not None else None, bias_mode=L.BIAS_ROW if b is not None else L.BIAS_NONE,
              residual=None, ldr=0, r_bstride=0, y=y.data_ptr(), ldy=ld, y_bstride=N * ld, M=N, N=T, K=K, batch=B,
              epilogue=L.EPI_NONE, a_dtype=L.BF16, y_dtype=L.BF16, prec=L.PREC_BF16, a_is_weight=1, a_lo=None)
    else:
        # fp32 activations cannot sit on the pre-split W side: split them once on the fly (crg_split_bf16, no torch arithmetic)
        xh, xl = torch.empty(x.shape, dtype=HALF, device=x.device), torch.empty(x.shape, dtype=HALF, device=x.device)
        L.check(L.load().crg_split_bf16(h, _st(), _p(x), _p(xh), _p(xl), x.numel()), h, "crg_split_bf16")
        _gemm(h, a=hi.data_ptr(), lda=K, a_bstride=0, a_lo=lo.data_ptr(), w=xh.data_ptr(), w_lo=xl.data_ptr(), ldw=K,
              w_bstride=T * K, bias=b.data_ptr() if b is not None else None,
              bias_mode=L.BIAS_ROW if b is not None else L.BIAS_NONE, residual=None, ldr=0, r_bstride=0, y=y.data_ptr(), ldy=ld,
              y_bstride=N * ld, M=N, N=T, K=K, batch=B, epilogue=L.EPI_NONE, a_dtype=L.BF16, y_dtype=L.F32,
              prec=L.PREC_BF16X3, a_is_weight=1)
    return y


# ---------------------------------------------------------------------------------- conv
def split_bf16(x: torch.Tensor):
    """fp32 image -> (hi, lo) bf16 planes, hi = bf16(x), lo = bf16(x - hi): the fp32-class conv's operand format."""
    _need_cuda(x)
    x = to_channels_last(x)
    n, c, hh, ww = x.shape
    hi, lo = empty_image(n, c, hh, ww, HALF, x.device), empty_image(n, c, hh, ww, HALF, x.device)
    h = _h(x)
    L.check(L.load().crg_split_bf16(h, _st(), _p(x), _p(hi), _p(lo), x.numel()), h, "crg_split_bf16")
    return hi, lo


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, padding=1,
           upsample2x: bool = False, x2: Optional[torch.Tensor] = None, cvec: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, x_lo: Optional[torch.Tensor] = None, gn_stats: bool = False, gn=None,
           x_mx: Optional[torch.Tensor] = None):
    """Implicit-GEMM conv over channels-last images.
    `gn` = (weight, bias, groups, eps, silu): also return silu?(GroupNorm(y)) - the norm that follows the conv inside a ResBlock
    (crg_conv_args.gn_y: on the small images of the two lowest UNet levels the launch that sums the K slices normalises as well);
    the call then returns (y, y_norm).  bf16 outputs only; anything else raises.
    padding: int (symmetric) or (top, left, bottom, right).  `upsample2x`: nearest-2x of the input is
    folded into the gather.  `x2`: second half of a virtual channel concat.  `cvec` fp32 [N, Cout] is
    added per sample (timestep embedding); `residual` is added after.  `x_lo`: x is the bf16 hi plane of a pre-split
    fp32 activation and x_lo its lo plane (group_norm(split=True) / split_bf16): fp32-class conv with fp32 output.
    `gn_stats`: the output feeds a GroupNorm - emit its statistics side channel where the path supports it (bf16, see above).
    `x_mx`: x is the fp16 plane and x_mx the e4m3 pair plane of MX-split fp32 activations (split_mx / group_norm(split='mx')): the
    fp32-class conv on CRG_PREC_F16MX (3x3, stride 1, pad 1, Cin % 64 == 0; fp32 output)."""
    _need_cuda(x, weight, bias, x2, cvec, residual, x_lo, x_mx)
    if x_mx is not None:
        return _conv2d_mx(x, x_mx, weight, bias, stride, padding, upsample2x, x2, cvec, residual, x_lo, gn_stats, gn)
    x = to_channels_last(x)
    if x_lo is not None:
        x_lo = to_channels_last(x_lo)
        if x.dtype != HALF or x_lo.dtype != HALF or x_lo.shape != x.shape or x2 is not None:
            raise L.CrgError("conv2d: x / x_lo must be two bf16 planes of one shape (and no second input)")
    n, c1, hh, ww = x.shape
    c2 = 0
    if x2 is not None:
        x2 = to_channels_last(x2)
        c2 = x2.shape[1]
        if x2.shape[0] != n or x2.shape[2:] != x.shape[2:] or x2.dtype != x.dtype:
            raise L.CrgError("conv2d: concat halves disagree in shape/dtype")
    cout, cin, ks, _ = weight.shape
    if cin != c1 + c2:
        raise L.CrgError(f"conv2d: weight expects {cin} input channels, got {c1}+{c2}")
    if isinstance(padding, int):
        pt = pl = pb = pr = padding
    else:
        pt, pl, pb, pr = padding
    hv, wv = (2 * hh, 2 * ww) if upsample2x else (hh, ww)
    ho = (hv + pt + pb - ks) // stride + 1
    wo = (wv + pl + pr - ks) // stride + 1
    if cin <= 8 or (cout <= 8 and (cin % 8 != 0 or cin < 64)):
        if x_lo is not None:
            raise L.CrgError("conv2d: thin-channel convs take the fp32 tensor, not split planes")
        if stride != 1 or upsample2x or x2 is not None or cvec is not None or residual is not None or gn is not None or (pt, pl, pb, pr) != (ks // 2,) * 4:
            raise L.CrgError("conv2d: thin-channel convs support only stride 1, 'same' padding, no fusions (no fused GroupNorm either)")
        y = empty_image(n, cout, ho, wo, x.dtype, x.device)
        w32 = f32_vec(weight)
        h = _h(x)
        L.check(L.load().crg_conv_small(h, _st(), _p(x), _p(w32), _p(f32_vec(bias)), _p(y), n, hh, ww, cin, cout, ks, _act_dt(x),
                                        _act_dt(y)), h, "crg_conv_small")
        return y
    planes = x_lo is not None
    split = planes or x.dtype == torch.float32
    hi, lo = packed_weight(weight, L.PACK_CONV, split)
    y = empty_image(n, cout, ho, wo, torch.float32 if planes else x.dtype, x.device)
    if residual is not None:
        residual = to_channels_last(residual)
        if residual.shape != y.shape or residual.dtype != y.dtype:
            raise L.CrgError("conv2d: residual must match the output in shape and dtype")
    if cvec is not None:
        if cvec.dtype != torch.float32 or tuple(cvec.shape) != (n, cout) or cvec.stride(1) != 1 or cvec.stride(0) % 4:
            raise L.CrgError("conv2d: cvec must be fp32 [N, Cout] with unit column stride and 16-byte aligned rows")
    a = L.ConvArgs(x=x.data_ptr(), x2=x2.data_ptr() if x2 is not None else None, C1=c1, C2=c2, w=hi.data_ptr(),
                   w_lo=lo.data_ptr() if lo is not None else None, bias=_p(f32_vec(bias)).value, cvec=_p(cvec).value,
                   cvec_ld=cvec.stride(0) if cvec is not None else 0, residual=_p(residual).value, y=y.data_ptr(), N=n, H=hh, W=ww, Cout=cout, Ho=ho, Wo=wo, ksize=ks, stride=stride,
                   pad_t=pt, pad_l=pl, upsample2x=int(upsample2x), x_dtype=_act_dt(x), y_dtype=_act_dt(y),
                   prec=L.PREC_BF16X3 if planes else _prec(x), x_lo=x_lo.data_ptr() if planes else None)
    y_norm = keep = None
    if gn is not None:
        gw, gb, groups, eps, silu = gn
        if planes or y.dtype != HALF or cout % groups or (cout // groups) % 8:
            raise L.CrgError("conv2d: the fused GroupNorm takes bf16 outputs with 8-aligned groups")
        y_norm = empty_image(n, cout, ho, wo, y.dtype, x.device)
        keep = (f32_vec(gw), f32_vec(gb))
        a.gn_gamma, a.gn_beta, a.gn_y = keep[0].data_ptr(), keep[1].data_ptr(), y_norm.data_ptr()
        a.gn_groups, a.gn_silu, a.gn_eps = int(groups), int(bool(silu)), float(eps)
    stats = _gn_stats_buffer(n * ho * wo, cout, ho * wo, x.dtype, y.dtype, x.device) if (gn_stats and gn is None and (planes or x.dtype == HALF)) else None
    if stats is not None and planes and (cout % 4 or VAE_GN_STATS is False):
        stats = None
    rows = C.c_int(32)
    if stats is not None:
        a.gn_stats = stats.data_ptr()
        if GN_TILE and not planes:
            a.gn_stats_rows = C.pointer(rows)  # the launch reports the granularity it wrote
    h = _h(x)
    L.check(L.load().crg_conv2d(h, _st(), C.byref(a)), h, "crg_conv2d")
    if stats is not None:
        y._crg_gn = (stats, y._version, ho * wo, rows.value)
    return y if gn is None else (y, y_norm)


def _conv2d_mx(x16, x8, weight, bias, stride, padding, upsample2x, x2, cvec, residual, x_lo, gn_stats, gn):
    x16 = to_channels_last(x16)
    n, c, hh, ww = x16.shape
    cout, cin, ks, _ = weight.shape
    if x16.dtype != torch.float16 or x8.dtype != torch.uint8 or tuple(x8.shape) != (n, hh, ww, 2 * c) or not x8.is_contiguous():
        raise L.CrgError("conv2d(x_mx=): x must be the fp16 plane [N, C, H, W] (channels-last) and x_mx the uint8 plane [N, H, W, 2 C]")
    if ks != 3 or stride != 1 or padding not in (1, (1, 1, 1, 1)) or upsample2x or x2 is not None or x_lo is not None or gn is not None or cvec is not None \
            or cin != c or c % 64:
        raise L.CrgError("conv2d(x_mx=): 3x3 / stride 1 / pad 1 convs with Cin % 64 == 0 only (no upsample / concat / fused GroupNorm)")
    w16, w8, whi, wlo = packed_weight_mx(weight)
    y = empty_image(n, cout, hh, ww, torch.float32, x16.device)
    if residual is not None:
        residual = to_channels_last(residual)
        if residual.shape != y.shape or residual.dtype != y.dtype:
            raise L.CrgError("conv2d: residual must match the output in shape and dtype")
    a = L.ConvArgs(x=x16.data_ptr(), x2=None, C1=c, C2=0, w=w16.data_ptr(), w_lo=w8.data_ptr(), bias=_p(f32_vec(bias)).value, cvec=None, cvec_ld=0,
                   residual=_p(residual).value, y=y.data_ptr(), N=n, H=hh, W=ww, Cout=cout, Ho=hh, Wo=ww, ksize=3, stride=1, pad_t=1, pad_l=1,
                   upsample2x=0, x_dtype=L.BF16, y_dtype=L.F32, prec=L.PREC_F16MX, x_lo=x8.data_ptr())
    a.mx_log2[0], a.mx_log2[1], a.mx_log2[2], a.mx_log2[3] = whi, wlo, MX_X_LOG2[0], MX_X_LOG2[1]
    stats = _gn_stats_buffer(n * hh * ww, cout, hh * ww, HALF, torch.float32, x16.device) if (gn_stats and VAE_GN_STATS and cout % 4 == 0) else None
    if stats is not None:
        a.gn_stats = stats.data_ptr()
    h = _h(x16)
    L.check(L.load().crg_conv2d(h, _st(), C.byref(a)), h, "crg_conv2d")
    if stats is not None:
        y._crg_gn = (stats, y._version, hh * ww)
    return y


def mx_conv_ok(x: torch.Tensor, conv_weight: torch.Tensor) -> bool:
    """Would conv2d(x16, w, x_mx=...) take this fp32 activation / 3x3 weight pair (and is the MX form enabled)?"""
    return (VAE_MX and x.is_cuda and x.dtype == torch.float32 and conv_weight.dim() == 4 and conv_weight.shape[2] == 3 and x.shape[1] % 64 == 0
            and conv_weight.shape[1] == x.shape[1] and conv_weight.shape[0] > 8 and conv_weight.shape[0] % 4 == 0)


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias=None, residual: Optional[torch.Tensor] = None, gn_stats: bool = False) -> torch.Tensor:
    """1x1 conv on a channels-last image == linear over its token view (no copy either way)."""
    x = to_channels_last(x)
    n, c, hh, ww = x.shape
    res_t = tokens_of(to_channels_last(residual)) if residual is not None else None
    y = linear(tokens_of(x), weight, bias, residual=res_t, gn_hw=hh * ww if gn_stats else None)
    return image_of_stats(y, hh, ww)


def image_of_stats(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """image_of() that carries the GroupNorm statistics `linear(..., gn_hw=h*w)` produced over to the image view."""
    img = image_of(t, h, w)
    st = getattr(t, "_crg_gn_pending", None)
    if st is not None:
        img._crg_gn = (st, img._version, h * w)
    return img


# ---------------------------------------------------------------------------------- attention
SCORE_BUDGET_BYTES = 256 << 20  # fp32 score buffer of the unfused attention path (queries are processed in chunks that fit it)


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, n_keys: int, scale: float) -> torch.Tensor:
    """softmax(Q K^T * scale) V with heads split along the channel dim.
    q: [B, Nq, C], k: [B, Nk, C], vt: [B, C, ld] (V transposed, ld = roundup(Nk, 8)) -> [B, Nq, C]."""
    _need_cuda(q, k, vt)
    B, Nq, Cc = q.shape
    Dh = Cc // heads
    ld = vt.shape[-1]
    h = _h(q)
    if q.dtype == HALF and Dh <= 160:
        # q / k may be column slices of one fused projection output: rows keep their parent stride
        def rows(t):
            if t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
                t = t.contiguous()
            return t, t.stride(1)
        q, ldq = rows(q)
        k, ldk = rows(k)
        o = torch.empty((B, Nq, Cc), dtype=q.dtype, device=q.device)
        L.check(L.load().crg_attention(h, _st(), _p(q), ldq, _p(k), ldk, _p(vt), ld, _p(o), Cc, B, heads, Nq, n_keys, Dh, scale,
                                       L.BF16), h, "crg_attention")
        return o
    q, k = q.contiguous(), k.contiguous()
    # Unfused path: S = QK^T (fp32) -> row softmax -> PV, all on the GEMM kernels.  Serves the fp32-class (BF16X3)
    # configuration and head dims beyond the flash kernel (the VAE's single-head C=512 AttnBlock, model.py:185-209).
    # The score matrix is never held whole: queries are processed in chunks sized so that the fp32 score buffer stays
    # <= ~256 MB (the reference's original path materialises all of it - 1 GiB per image at 1024x1024, attention.py:389-429),
    # hi / lo planes come from crg_split_bf16, and no PyTorch arithmetic runs.
    split = q.dtype == torch.float32
    o = torch.empty_like(q)
    kp = (n_keys + 7) // 8 * 8
    prec = L.PREC_BF16X3 if split else L.PREC_BF16

    def planes(t):
        if not split:
            return t, None
        th, tl = torch.empty(t.shape, dtype=HALF, device=t.device), torch.empty(t.shape, dtype=HALF, device=t.device)
        L.check(L.load().crg_split_bf16(h, _st(), _p(t), _p(th), _p(tl), t.numel()), h, "crg_split_bf16")
        return th, tl
    kh, kl = planes(k)            # [B, Nk, C]
    vh, vl = planes(vt)           # [B, C, ld]
    qc = max(8, min(Nq, SCORE_BUDGET_BYTES // (4 * heads * kp) // 8 * 8))  # queries per chunk
    s = torch.empty((heads, min(qc, Nq), kp), dtype=torch.float32, device=q.device)
    if kp != n_keys:
        s[:, :, n_keys:].zero_()
    for b in range(B):
        for q0 in range(0, Nq, qc):
            nq = min(qc, Nq - q0)
            sv = s[:, :nq]
            ld_s = s.shape[1] * kp  # head stride of the score buffer
            _gemm(h, a=q[b, q0:].data_ptr(), lda=Cc, a_bstride=Dh, w=kh[b].data_ptr(), w_lo=kl[b].data_ptr() if split else None, ldw=Cc,
                  w_bstride=Dh, bias=None, bias_mode=L.BIAS_NONE, residual=None, ldr=0, r_bstride=0, y=sv.data_ptr(), ldy=kp,
                  y_bstride=ld_s, M=nq, N=n_keys, K=Dh, batch=heads, epilogue=L.EPI_NONE, a_dtype=_act_dt(q), y_dtype=L.F32,
                  prec=prec, a_is_weight=0, a_lo=None)
            if nq == s.shape[1]:
                softmax_rows_(s, n_keys, scale)
            else:  # last, shorter chunk: the rows of one head are contiguous [nq, kp]
                for hd in range(heads):
                    softmax_rows_(s[hd, :nq], n_keys, scale)
            _gemm(h, a=sv.data_ptr(), lda=kp, a_bstride=ld_s, w=vh[b].data_ptr(), w_lo=vl[b].data_ptr() if split else None, ldw=ld,
                  w_bstride=Dh * ld, bias=None, bias_mode=L.BIAS_NONE, residual=None, ldr=0, r_bstride=0, y=o[b, q0:].data_ptr(), ldy=Cc,
                  y_bstride=Dh, M=nq, N=Dh, K=kp, batch=heads, epilogue=L.EPI_NONE, a_dtype=L.F32, y_dtype=_act_dt(q), prec=prec,
                  a_is_weight=0, a_lo=None)
    return o


def attention_rows_v(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """softmax(Q K^T * scale) V with V ROW-MAJOR ([B, Nk, C], like K): q, k, v may be column slices of one fused projection
    output (rows keep the parent's stride), which is what makes `to_q | to_k | to_v` ONE GEMM launch.  bf16, d_head <= 160
    (crg_attention_v: the flash kernel transposes V on its LDS read)."""
    _need_cuda(q, k, v)
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    Dh = Cc // heads
    if q.dtype != HALF or Dh > 160 or k.shape != v.shape or k.shape[0] != B or k.shape[2] != Cc:
        raise L.CrgError("attention_rows_v: bf16 q / k / v with matching shapes and d_head <= 160 expected")

    def rows(t):
        if t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            t = t.contiguous()
        return t, t.stride(1)
    q, ldq = rows(q)
    k, ldk = rows(k)
    v, ldv = rows(v)
    o = torch.empty((B, Nq, Cc), dtype=q.dtype, device=q.device)
    h = _h(q)
    L.check(L.load().crg_attention_v(h, _st(), _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(o), Cc, B, heads, Nq, Nk, Dh, scale, L.BF16), h,
            "crg_attention_v")
    return o


# ---------------------------------------------------------------------------------- small ops
def timestep_embedding(t: torch.Tensor, dim: int, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    _need_cuda(t)
    t = t.detach().float().contiguous()
    y = torch.empty((t.shape[0], dim), dtype=dtype, device=t.device)
    h = _h(t)
    L.check(L.load().crg_timestep_embedding(h, _st(), _p(t), _p(y), t.shape[0], dim, _DT[dtype]), h, "crg_timestep_embedding")
    return y


def silu(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    h = _h(x)
    L.check(L.load().crg_silu(h, _st(), _p(x), _p(y), x.numel(), _act_dt(x)), h, "crg_silu")
    return y


def axpby_(y: torch.Tensor, x: torch.Tensor, a: float, b: float = 1.0) -> torch.Tensor:
    """In place y = a*x + b*y (same dense layout on both sides)."""
    _need_cuda(x, y)
    if x.shape != y.shape or x.stride() != y.stride() or x.dtype != y.dtype:
        raise L.CrgError("axpby_: operands must agree in shape, strides and dtype")
    h = _h(x)
    L.check(L.load().crg_axpby(h, _st(), _p(x), _p(y), x.numel(), a, b, _act_dt(x)), h, "crg_axpby")
    return _written(y)


def cfg_euler_step_(x: torch.Tensor, eps2: torch.Tensor, noise: Optional[torch.Tensor], sigma: float, dt: float, cfg_scale: float,
                    noise_scale: float = 0.0) -> torch.Tensor:
    """In-place fused sampler step (crg_cfg_euler_step): x fp32 [b, ...], eps2 fp32 [2b, ...] (uncond half first)."""
    _need_cuda(x, eps2, noise)
    if x.dtype != torch.float32 or eps2.dtype != torch.float32 or not x.is_contiguous() or not eps2.is_contiguous() \
            or eps2.numel() != 2 * x.numel() or (noise is not None and (noise.dtype != torch.float32 or not noise.is_contiguous()
                                                                         or noise.numel() != x.numel())):
        raise L.CrgError("cfg_euler_step_: contiguous fp32 x [b,...], eps [2b,...] and noise [b,...] expected")
    h = _h(x)
    L.check(L.load().crg_cfg_euler_step(h, _st(), _p(x), _p(eps2), _p(noise), x.numel(), float(sigma), float(dt), float(cfg_scale),
                                        float(noise_scale)), h, "crg_cfg_euler_step")
    return _written(x)


# ---------------------------------------------------------------------------------- profiling
class profile:
    """Context manager: per-kernel device time (HIP events on the launch stream) + algorithmic FLOPs/bytes of
    every kernel launched by crg_* calls inside it (bench.py's `roofline`).  `kernels` is keyed by kernel slot
    (one per kernel symbol, with its rocprofv3 name), `result` sums the slots into families."""

    def __init__(self, device=None):
        self.dev = torch.cuda.current_device() if device is None else device
        self.result = None

    def __enter__(self):
        h = L.ctx(self.dev)
        L.check(L.load().crg_profile_begin(h), h, "crg_profile_begin")
        return self

    def __exit__(self, *exc):
        h = L.ctx(self.dev)
        p = L.Profile()
        L.check(L.load().crg_profile_end(h, _st(), C.byref(p)), h, "crg_profile_end")
        lib = L.load()
        self.kernels = {L.SLOT_NAMES[i]: dict(ms=p.ms[i], flops=p.flops[i], bytes=p.bytes[i], launches=p.launches[i],
                                              symbol=lib.crg_kernel_name(i).decode(), family=L.SLOT_FAMILY[i])
                        for i in range(L.K_SLOTS)}
        self.result = {}
        for k in self.kernels.values():
            f = self.result.setdefault(k["family"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            for key in ("ms", "flops", "bytes", "launches"):
                f[key] += k[key]
        return False
