"""ctypes binding of libcrg_hip.so (include/crg_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a Python exception
is raised (the reference's ML process has no handler, modules/cremage/mp/mp.py:125, so clean
exceptions - never aborts - are the contract).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRG_LIB: developer override used for A/B runs of two builds on the same GPU box (tools/); the default is the in-tree build
# CRG_HALF=f16: the fp16-operand build of the same kernels (libcrg_hip_f16.so); activations / packed weights are torch.float16 then
HALF_F16 = os.environ.get("CRG_HALF", "bf16").lower() in ("f16", "fp16", "float16", "half")
LIB_PATH = os.path.abspath(os.environ["CRG_LIB"]) if os.environ.get("CRG_LIB") else os.path.join(_HERE, "libcrg_hip_f16.so" if HALF_F16 else "libcrg_hip.so")

BF16, F32, F16 = 0, 1, 2
PREC_BF16, PREC_BF16X3, PREC_F16MX = 0, 1, 2
EPI_NONE, EPI_SILU, EPI_GEGLU = 0, 1, 2
BIAS_NONE, BIAS_COL, BIAS_ROW = 0, 1, 2
PACK_LINEAR, PACK_CONV, PACK_GEGLU = 0, 1, 2
K_SLOTS = 17  # enum crg_kernel_slot
SLOT_NAMES = ["gemm_w1", "gemm_w4", "gemm_w5", "gemm_x3", "conv_w1", "conv_w4", "conv_w5", "conv_x3", "splitk_reduce", "attention",
              "gn_stats", "gn_apply", "layernorm", "elementwise", "conv_small", "softmax", "lngemm"]
SLOT_FAMILY = ["gemm", "gemm", "gemm", "gemm", "conv", "conv", "conv", "conv", "splitk_reduce", "attention", "groupnorm", "groupnorm",
               "layernorm", "elementwise", "conv_small", "softmax", "gemm"]

c_void_p, c_int, c_int64, c_float, c_size_t = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", c_void_p), ("lda", c_int64), ("a_bstride", c_int64),
        ("w", c_void_p), ("ldw", c_int64), ("w_bstride", c_int64),
        ("w_lo", c_void_p),
        ("bias", c_void_p), ("bias_mode", c_int),
        ("residual", c_void_p), ("ldr", c_int64), ("r_bstride", c_int64),
        ("y", c_void_p), ("ldy", c_int64), ("y_bstride", c_int64),
        ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int),
        ("epilogue", c_int),
        ("a_dtype", c_int), ("y_dtype", c_int), ("prec", c_int),
        ("a_is_weight", c_int),
        ("a_lo", c_void_p),
        ("gn_stats", c_void_p),
        ("vt", c_void_p), ("vt_n0", c_int), ("vt_tokens", c_int), ("vt_ld", c_int64),
        ("row_stats", c_void_p), ("row_stats_parts", c_int),
        ("ln_stats", c_void_p), ("ln_parts", c_int), ("ln_colsum", c_void_p), ("ln_eps", c_float),
    ]


class LnGemmArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int64),
        ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
        ("w", c_void_p), ("ldw", c_int64),
        ("bias", c_void_p),
        ("y", c_void_p), ("ldy", c_int64),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("epilogue", c_int),
        ("vt", c_void_p), ("vt_n0", c_int), ("vt_tokens", c_int), ("vt_ld", c_int64),
        ("residual", c_void_p), ("ldr", c_int64),
    ]


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", c_void_p), ("x2", c_void_p), ("C1", c_int), ("C2", c_int),
        ("w", c_void_p), ("w_lo", c_void_p),
        ("bias", c_void_p), ("cvec", c_void_p),
        ("cvec_ld", c_int64),
        ("residual", c_void_p),
        ("y", c_void_p),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cout", c_int), ("Ho", c_int), ("Wo", c_int),
        ("ksize", c_int), ("stride", c_int), ("pad_t", c_int), ("pad_l", c_int),
        ("upsample2x", c_int),
        ("x_dtype", c_int), ("y_dtype", c_int), ("prec", c_int),
        ("x_lo", c_void_p),
        ("gn_stats", c_void_p),
        ("gn_gamma", c_void_p), ("gn_beta", c_void_p), ("gn_y", c_void_p),
        ("gn_groups", c_int), ("gn_silu", c_int), ("gn_eps", C.c_float),
        ("mx_log2", c_int * 4),
        ("gn_stats_rows", C.POINTER(c_int)),
    ]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * K_SLOTS), ("flops", C.c_double * K_SLOTS), ("bytes", C.c_double * K_SLOTS),
                ("launches", C.c_int64 * K_SLOTS)]


# name -> (restype, argtypes); every symbol include/crg_hip.h declares
SIGNATURES = {
    "crg_version": (c_int, []),
    "crg_half_kind": (c_int, []),
    "crg_ctx_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "crg_ctx_destroy": (None, [c_void_p]),
    "crg_last_error": (C.c_char_p, [c_void_p]),
    "crg_ctx_reserve": (c_int, [c_void_p, c_size_t]),
    "crg_kernel_name": (C.c_char_p, [c_int]),
    "crg_profile_begin": (c_int, [c_void_p]),
    "crg_profile_end": (c_int, [c_void_p, c_void_p, C.POINTER(Profile)]),
    "crg_groupnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_int, c_float, c_int, c_int]),
    "crg_groupnorm_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_float, c_int]),
    "crg_groupnorm_pre_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_float, c_int]),
    "crg_groupnorm_pre": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_int, c_float, c_int, c_int, c_int, c_int]),
    "crg_split_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "crg_split_mx": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int]),
    "crg_groupnorm_mx": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                 c_int, c_int, c_int]),
    "crg_pack_weight_mx": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int]),
    "crg_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int]),
    "crg_gemm": (c_int, [c_void_p, c_void_p, C.POINTER(GemmArgs)]),
    "crg_conv2d": (c_int, [c_void_p, c_void_p, C.POINTER(ConvArgs)]),
    "crg_ln_gemm": (c_int, [c_void_p, c_void_p, C.POINTER(LnGemmArgs)]),
    "crg_pack_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "crg_pack_geglu_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "crg_pack_ln_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "crg_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                              c_int, c_int, c_int, c_int, c_int, c_float, c_int]),
    "crg_attention_v": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                              c_int, c_int, c_int, c_int, c_int, c_float, c_int]),
    "crg_softmax_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_float, c_int]),
    "crg_conv_small": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int]),
    "crg_timestep_embedding": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "crg_silu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int]),
    "crg_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "crg_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "crg_cfg_euler_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float]),
    "crg_axpby": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_int]),
    "crg_affine_cast": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int,
                                c_int]),
}


class CrgError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrgError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(cremage_amd has no CPU or PyTorch fallback)")
    import torch  # noqa: F401  (before the dlopen: the library must resolve the HIP runtime to the one PyTorch loads; with the system
    #                             runtime loaded first the process ends up with two, and the one behind this library sees no device)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.crg_version() != 103:
        raise CrgError(f"libcrg_hip.so version {lib.crg_version()} does not match the binding (103)")
    if lib.crg_half_kind() != (1 if HALF_F16 else 0):
        raise CrgError(f"{LIB_PATH} computes in {'fp16' if lib.crg_half_kind() else 'bf16'} but CRG_HALF asks for {'fp16' if HALF_F16 else 'bf16'}")
    _lib = lib
    return lib


_ctxs = {}
_lane = 0


def set_lane(i: int) -> int:
    """Select which of the device's contexts the following crg_* calls use.  A context owns scratch (split-K slabs, GroupNorm
    partials) and is not re-entrant, so work that runs CONCURRENTLY on several HIP streams needs one context per stream
    ("lane"); the single-stream caller never touches this.  Returns the previous lane."""
    global _lane
    prev, _lane = _lane, int(i)
    return prev


def ctx(device_index: int):
    """One context per (device, lane) (include/crg_hip.h: not re-entrant, single-threaded caller; lane 0 unless set_lane)."""
    key = (device_index, _lane)
    h = _ctxs.get(key)
    if h is None:
        lib = load()
        out = c_void_p()
        rc = lib.crg_ctx_create(device_index, C.byref(out))
        if rc != 0:
            raise CrgError(f"crg_ctx_create(device={device_index}) failed with {rc}: no usable HIP device")
        h = out
        _ctxs[key] = h
    return h


def check(rc: int, h, what: str):
    if rc != 0:
        msg = load().crg_last_error(h)
        raise CrgError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
