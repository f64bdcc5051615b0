// Shared device/host helpers for libcrg_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/crg_hip.h"

// The library's 16-bit element type.  Default build: bfloat16 (BASELINE.json configs[1] says bf16).  -DCRG_F16_BUILD compiles the SAME
// kernels for IEEE fp16 operands (libcrg_hip_f16.so): the matrix instructions are the _f16 forms - same cycles as the _bf16 ones
// (MI355X_MICROARCH.md, Matrix cores) - every conversion / rounding is the type's own, accumulation stays fp32.  That is the arithmetic
// of the reference's own GPU flow (fp16 weights under torch.autocast, modules/sd/image_generator.py:489-493,748-751): three more
// mantissa bits per operand than bf16, at the price of fp16's range.  The identifier stays `bf16` in the sources: "the half type".
#ifdef CRG_F16_BUILD
typedef _Float16 bf16;
#define CRG_HALF_KIND 1
#define CRG_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define CRG_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 bf16;
#define CRG_HALF_KIND 0
#define CRG_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define CRG_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CRG_WAVE 64

// ---- context ------------------------------------------------------------------------------
struct crg_prof_rec {
  hipEvent_t e0, e1;
  int family;
  double flops, bytes;
};

struct crg_ctx {
  int device = 0;
  int n_cu = 0;  // compute units of the device (residency bound of the kernels whose blocks wait for each other)
  std::string err;
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  std::vector<void*> retired;  // outgrown scratch buffers (kept alive: captured graphs / queued kernels may reference them)
  void* zero_page = nullptr;  // 4 KiB of zeros: LDS-DMA source for conv padding and tile tails
  std::vector<const void*> lds_attr;  // kernels whose dynamic-LDS limit was raised ON THIS CONTEXT'S DEVICE (crg_set_dyn_lds)
  bool gn_fused = false;  // set by the split-K reduce launch when it also ran the GroupNorm behind the conv (crg_conv_args.gn_y)
  bool profiling = false;
  std::vector<crg_prof_rec> recs;
  std::vector<hipEvent_t> event_pool;
};

int crg_fail(crg_ctx* ctx, int code, const char* fmt, ...);
// scratch of at least `bytes` (grows with hipMalloc when too small: NOT capture-safe, so callers
// running under graph capture must crg_ctx_reserve first)
void* crg_scratch(crg_ctx* ctx, size_t bytes);

// RAII bracket used by every launcher: records events when profiling is on.
struct crg_prof_scope {
  crg_ctx* ctx;
  hipStream_t st;
  int idx = -1;
  crg_prof_scope(crg_ctx* c, hipStream_t s, int family, double flops, double bytes);
  ~crg_prof_scope();
};

#define CRG_CHECK_LAUNCH(ctx, what)                                                    \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) return crg_fail(ctx, -5, "%s: launch failed: %s", what, hipGetErrorString(e_)); \
  } while (0)

#define CRG_REQUIRE(ctx, cond, ...)                         \
  do {                                                      \
    if (!(cond)) return crg_fail(ctx, -22, __VA_ARGS__);    \
  } while (0)

static inline size_t crg_dtype_size(int dt) { return dt == CRG_F32 ? 4 : 2; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the function: remembered per CONTEXT (= per device and
// lane), not per process - a process that drives several GPUs sets it once on each.  Called outside graph capture by the warm-up.
static inline int crg_set_dyn_lds(crg_ctx* ctx, const void* kern, size_t bytes, const char* what) {
  for (const void* k : ctx->lds_attr)
    if (k == kern) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return crg_fail(ctx, -5, "%s: cannot set %zu B dynamic LDS: %s", what, bytes, hipGetErrorString(e));
  ctx->lds_attr.push_back(kern);
  return 0;
}

// ---- device helpers -------------------------------------------------------------------------
// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float crg_silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf-GELU (F.gelu default, attention.py:96 GEGLU).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, branch-free,
// ~14 VALU ops): libm's erff is a ~40-instruction branchy sequence, which made the GEGLU epilogue cost twice the MFMA
// time of the K=320 feed-forward GEMM.
__device__ __forceinline__ float crg_erf_f(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  q = __builtin_fmaf(q, t, 1.421413741f);
  q = __builtin_fmaf(q, t, -0.284496736f);
  q = __builtin_fmaf(q, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = __builtin_fmaf(-q * t, e, 1.0f);
  return __builtin_copysignf(r, x);
}
#ifdef CRG_ABL_GELU  // timing ablation only (tools/build_variant.sh): what the GEGLU epilogue's erf costs
__device__ __forceinline__ float crg_gelu_erf_f(float x) { return x; }
#else
__device__ __forceinline__ float crg_gelu_erf_f(float x) { return 0.5f * x * (1.0f + crg_erf_f(x * 0.70710678118654752440f)); }
#endif

// four floats -> four OCP e4m3 bytes (saturating at +-448), lowest byte first
__device__ __forceinline__ unsigned crg_pack4_e4m3(float a, float b, float c, float d) {
  int v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
  b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
  c = __builtin_fminf(__builtin_fmaxf(c, -448.f), 448.f);
  d = __builtin_fminf(__builtin_fmaxf(d, -448.f), 448.f);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
#endif
  return (unsigned)v;
}
// MX planes of eight consecutive channels of one pixel / weight row (CRG_PREC_F16MX): fp16 values to p16[0..7], e4m3(half * 2^sh) to
// p8[0..7] and e4m3((f - half) * 2^sl) to p8[64..71] (p8 = the 128-byte record of the 64-channel chunk, already offset to the channel)
__device__ __forceinline__ void crg_store_mx8(const float (&f)[8], void* p16, unsigned char* p8, float sh, float sl) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 h;
  float hf[8], lf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (_Float16)f[e];
    hf[e] = (float)h[e];
    lf[e] = f[e] - hf[e];
  }
  *reinterpret_cast<h8*>(p16) = h;
  uint2 a, b;
  a.x = crg_pack4_e4m3(hf[0] * sh, hf[1] * sh, hf[2] * sh, hf[3] * sh);
  a.y = crg_pack4_e4m3(hf[4] * sh, hf[5] * sh, hf[6] * sh, hf[7] * sh);
  b.x = crg_pack4_e4m3(lf[0] * sl, lf[1] * sl, lf[2] * sl, lf[3] * sl);
  b.y = crg_pack4_e4m3(lf[4] * sl, lf[5] * sl, lf[6] * sl, lf[7] * sl);
  *reinterpret_cast<uint2*>(p8) = a;
  *reinterpret_cast<uint2*>(p8 + 64) = b;
}

template <typename T>
struct crg_vec8;  // 8 consecutive elements of T, loaded/stored as one (bf16) or two (f32) 16-byte accesses
template <>
struct crg_vec8<bf16> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void store(bf16* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
  __device__ __forceinline__ void zero() { v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16)f; }
};
template <>
struct crg_vec8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
  __device__ __forceinline__ void zero() { a = f32x4{0, 0, 0, 0}; b = f32x4{0, 0, 0, 0}; }
  __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
  __device__ __forceinline__ void set(int i, float f) { if (i < 4) a[i] = f; else b[i - 4] = f; }
};
