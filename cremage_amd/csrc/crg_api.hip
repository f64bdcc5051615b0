// Context, error reporting, scratch, per-launch event timing and weight packing.
#include "crg_common.h"

int crg_fail(crg_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  return code;
}

void* crg_scratch(crg_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return ctx->scratch;
  // Grow geometrically and RETIRE (do not free) the old buffer: a captured hipGraph may have its address baked into
  // kernel arguments, and kernels already queued on the stream may still be using it.  Retired buffers are released
  // with the context.  Growth itself is a hipMalloc, i.e. not capture-safe: reserve before capturing.
  size_t want = ctx->scratch_bytes ? ctx->scratch_bytes * 2 : (size_t(64) << 20);
  while (want < bytes) want *= 2;
  void* fresh = nullptr;
  if (hipMalloc(&fresh, want) != hipSuccess) return nullptr;
  if (ctx->scratch) ctx->retired.push_back(ctx->scratch);
  ctx->scratch = fresh;
  ctx->scratch_bytes = want;
  return ctx->scratch;
}

crg_prof_scope::crg_prof_scope(crg_ctx* c, hipStream_t s, int family, double flops, double bytes) : ctx(c), st(s) {
  if (!c || !c->profiling) return;
  crg_prof_rec r;
  r.family = family;
  r.flops = flops;
  r.bytes = bytes;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, s);
  c->recs.push_back(r);
  idx = (int)c->recs.size() - 1;
}
crg_prof_scope::~crg_prof_scope() {
  if (idx >= 0) (void)hipEventRecord(ctx->recs[idx].e1, st);
}

extern "C" int crg_version(void) { return CRG_VERSION; }
extern "C" int crg_half_kind(void) { return CRG_HALF_KIND; }

extern "C" int crg_ctx_create(int device, crg_ctx** out) {
  if (!out) return -22;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -19;  // ENODEV
  if (hipSetDevice(device) != hipSuccess) return -19;
  crg_ctx* c = new (std::nothrow) crg_ctx();
  if (!c) return -12;
  c->device = device;
  {
    hipDeviceProp_t prop;
    c->n_cu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 0;
  }
  if (hipMalloc(&c->zero_page, 4096) != hipSuccess || hipMemset(c->zero_page, 0, 4096) != hipSuccess) {
    delete c;
    return -12;
  }
  *out = c;
  return 0;
}

extern "C" void crg_ctx_destroy(crg_ctx* ctx) {
  if (!ctx) return;
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  for (void* p : ctx->retired) (void)hipFree(p);
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  for (auto& r : ctx->recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  delete ctx;
}

extern "C" const char* crg_last_error(crg_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int crg_ctx_reserve(crg_ctx* ctx, size_t bytes) {
  if (!ctx) return -22;
  return crg_scratch(ctx, bytes) ? 0 : crg_fail(ctx, -12, "cannot reserve %zu bytes of scratch", bytes);
}

extern "C" const char* crg_kernel_name(int slot) {
  static const char* const names[CRG_K_SLOTS] = {
      "gemm_glds_kernel<1, YT, false, STAGES, WMT, KG>", "gemm_glds_kernel<4, YT, false, STAGES, WMT, KG>",
      "gemm_glds_kernel<5, YT, false, STAGES, WMT, KG>", "gemm_kernel<WNT, NSPLIT, AT, YT, false, KG>",
      "gemm_glds_kernel<1, YT, true, STAGES, WMT, KG>",  "conv3_pp_kernel<4, ...> | conv3_rowhalo_kernel<4, ...> | gemm_glds_kernel<4, YT, true, STAGES, WMT, KG>",
      "conv3_pp_kernel<5, ...> | conv3_rowhalo_kernel<5, ...> | gemm_glds_kernel<5, YT, true, STAGES, WMT, KG>", "conv3_rowhalo_kernel<WNT, float, false, 2, 2> | gemm_kernel<WNT, NSPLIT, AT, YT, true, KG>",
      "splitk_reduce_kernel<YT>", "attn_kernel<KS, NV>", "gn_stats_kernel<T>", "gn_apply_kernel<T> / gn_small_kernel<T, VPT>", "layernorm_kernel<T>",
      "elementwise (silu / axpby / affine_cast / transpose / timestep_embedding kernels)", "conv_small_*_kernel", "softmax_rows_kernel<T>",
      "lngemm_kernel<WNT, KT, PAIR>"};
  return (slot >= 0 && slot < CRG_K_SLOTS) ? names[slot] : "?";
}

extern "C" int crg_profile_begin(crg_ctx* ctx) {
  if (!ctx) return -22;
  for (auto& r : ctx->recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  ctx->recs.clear();
  ctx->profiling = true;
  return 0;
}

extern "C" int crg_profile_end(crg_ctx* ctx, void* stream, crg_profile* out) {
  if (!ctx || !out) return -22;
  ctx->profiling = false;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return crg_fail(ctx, -5, "profile_end: stream sync failed");
  for (int i = 0; i < CRG_K_SLOTS; ++i) {
    out->ms[i] = out->flops[i] = out->bytes[i] = 0.0;
    out->launches[i] = 0;
  }
  for (auto& r : ctx->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && r.family >= 0 && r.family < CRG_K_SLOTS) {
      out->ms[r.family] += ms;
      out->flops[r.family] += r.flops;
      out->bytes[r.family] += r.bytes;
      out->launches[r.family] += 1;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  ctx->recs.clear();
  return 0;
}

// ---- weight packing ------------------------------------------------------------------------------
namespace {

template <typename ST>
__global__ void pack_kernel(const ST* __restrict__ src, bf16* __restrict__ hi, bf16* __restrict__ lo, int kind, int n_out,
                            int n_in, int ks, long total) {
  // destination element index d -> (row o, col k); CONV: k = tap * n_in + ci <- src[o][ci][tap]
  for (long d = (long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long)gridDim.x * blockDim.x) {
    const int kk = ks * ks * n_in;
    const int o = (int)(d / kk);
    const int k = (int)(d - (long)o * kk);
    long s;
    if (kind == CRG_PACK_CONV) {
      int tap, ci;
      if (ks == 3 && n_in % 64 == 0) {  // chunk-major K: [Cin/64][tap][64] (see GemmP::cm in gemm_conv.hip)
        const int taps = ks * ks;
        const int chunk = k / (taps * 64), r = k - chunk * taps * 64;
        tap = r / 64;
        ci = chunk * 64 + (r - tap * 64);
      } else {
        tap = k / n_in;
        ci = k - tap * n_in;
      }
      s = ((long)o * n_in + ci) * (ks * ks) + tap;
    } else if (kind == CRG_PACK_GEGLU) {
      // packed row o: group q = o/32, r = o%32 -> source row j + half*F, j = q*16 + r%16, half = r/16
      const int F = n_out / 2;
      const int q = o >> 5, r = o & 31;
      const int j = q * 16 + (r & 15);
      const int srow = j + (r >> 4) * F;
      s = (long)srow * n_in + k;
    } else {
      s = d;
    }
    const float f = (float)src[s];
    const bf16 h = (bf16)f;
    hi[d] = h;
    if (lo) lo[d] = (bf16)(f - (float)h);
  }
}

__global__ void pack_geglu_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int n2) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n2) return;
  const int F = n2 / 2;
  const int q = o >> 5, r = o & 31;
  dst[o] = src[q * 16 + (r & 15) + (r >> 4) * F];
}

// one block per packed row o: dst_w = half(W * gamma), colsum over the rounded values, bias' = W beta + bias (fixed-order block reduce)
template <typename ST>
__global__ __launch_bounds__(256) void pack_ln_kernel(const ST* __restrict__ src, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias, int kind, int n_out, int n_in, bf16* __restrict__ dw,
                                                      float* __restrict__ ds, float* __restrict__ db) {
  __shared__ float red[2][4];
  const int o = blockIdx.x, t = threadIdx.x;
  int srow = o;
  if (kind == CRG_PACK_GEGLU) {
    const int F = n_out / 2, q = o >> 5, r = o & 31;
    srow = q * 16 + (r & 15) + (r >> 4) * F;
  }
  const ST* w = src + (long)srow * n_in;
  float s = 0.f, b = 0.f;
  for (int k = t; k < n_in; k += 256) {
    const float f = (float)w[k];
    const bf16 h = (bf16)(f * gamma[k]);
    dw[(long)o * n_in + k] = h;
    s += (float)h;
    b = __builtin_fmaf(f, beta[k], b);
  }
#pragma unroll
  for (int x = 32; x > 0; x >>= 1) {
    s += __shfl_xor(s, x);
    b += __shfl_xor(b, x);
  }
  if ((t & 63) == 0) {
    red[0][t >> 6] = s;
    red[1][t >> 6] = b;
  }
  __syncthreads();
  if (t == 0) {
    ds[o] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    db[o] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) + (bias ? bias[srow] : 0.f);
  }
}

// conv weight [Cout][Cin][3][3] -> MX planes in the conv's chunk-major K order [Cin / 64][tap][64]: w16 fp16 [Cout][K], w8 [Cout][K / 64][128]
// (64 e4m3 bytes of half(w) * 2^sh, then 64 of (w - half(w)) * 2^sl); thread = eight consecutive k of one row
template <typename ST>
__global__ __launch_bounds__(256) void pack_mx_kernel(const ST* __restrict__ src, _Float16* __restrict__ w16, unsigned char* __restrict__ w8,
                                                      int n_in, long total8, float sh, float sl) {
  const int kk = 9 * n_in;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
    const long d = i * 8;
    const int o = (int)(d / kk), k = (int)(d - (long)o * kk);
    const int chunk = k / (9 * 64), r = k - chunk * 9 * 64, tap = r / 64, c0 = r - tap * 64;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)src[((long)o * n_in + chunk * 64 + c0 + e) * 9 + tap];
    crg_store_mx8(f, w16 + d, w8 + 2 * (d - c0) + c0, sh, sl);
  }
}

}  // namespace

extern "C" int crg_pack_weight_mx(crg_ctx* ctx, void* stream, const void* src, int src_dtype, int n_out, int n_in, void* dst16, void* dst8,
                                  int hi_log2, int lo_log2) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, src && dst16 && dst8 && n_out > 0 && n_in > 0 && n_in % 64 == 0, "pack_weight_mx: 3x3 conv weights with Cin %% 64 == 0 (got %d)", n_in);
  const long total8 = (long)n_out * n_in * 9 / 8;
  const int grid = (int)((total8 + 255) / 256 < 4096 ? (total8 + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  const float sh = ldexpf(1.f, hi_log2), sl = ldexpf(1.f, lo_log2);
  if (src_dtype == CRG_F32) hipLaunchKernelGGL(pack_mx_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)src, (_Float16*)dst16, (unsigned char*)dst8, n_in, total8, sh, sl);
  else if (src_dtype == CRG_BF16) hipLaunchKernelGGL(pack_mx_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)src, (_Float16*)dst16, (unsigned char*)dst8, n_in, total8, sh, sl);
  else if (src_dtype == CRG_F16) hipLaunchKernelGGL(pack_mx_kernel<_Float16>, dim3(grid), dim3(256), 0, st, (const _Float16*)src, (_Float16*)dst16, (unsigned char*)dst8, n_in, total8, sh, sl);
  else return crg_fail(ctx, -22, "pack_weight_mx: unsupported source dtype %d", src_dtype);
  CRG_CHECK_LAUNCH(ctx, "pack_weight_mx");
  return 0;
}

extern "C" int crg_pack_ln_weight(crg_ctx* ctx, void* stream, const void* src, int src_dtype, const float* gamma, const float* beta,
                                  const float* bias, int kind, int n_out, int n_in, void* dst_w, float* dst_colsum, float* dst_bias) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, src && gamma && beta && dst_w && dst_colsum && dst_bias && n_out > 0 && n_in > 0, "pack_ln_weight: bad arguments");
  CRG_REQUIRE(ctx, kind == CRG_PACK_LINEAR || kind == CRG_PACK_GEGLU, "pack_ln_weight: kind %d unsupported", kind);
  if (kind == CRG_PACK_GEGLU) CRG_REQUIRE(ctx, n_out % 32 == 0, "pack_ln_weight: GEGLU needs n_out %% 32 == 0, got %d", n_out);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == CRG_F32)
    hipLaunchKernelGGL(pack_ln_kernel<float>, dim3(n_out), dim3(256), 0, st, (const float*)src, gamma, beta, bias, kind, n_out, n_in, (bf16*)dst_w, dst_colsum, dst_bias);
  else if (src_dtype == CRG_BF16)
    hipLaunchKernelGGL(pack_ln_kernel<bf16>, dim3(n_out), dim3(256), 0, st, (const bf16*)src, gamma, beta, bias, kind, n_out, n_in, (bf16*)dst_w, dst_colsum, dst_bias);
  else if (src_dtype == CRG_F16)
    hipLaunchKernelGGL(pack_ln_kernel<_Float16>, dim3(n_out), dim3(256), 0, st, (const _Float16*)src, gamma, beta, bias, kind, n_out, n_in, (bf16*)dst_w, dst_colsum, dst_bias);
  else
    return crg_fail(ctx, -22, "pack_ln_weight: unsupported source dtype %d", src_dtype);
  CRG_CHECK_LAUNCH(ctx, "pack_ln_weight");
  return 0;
}

extern "C" int crg_pack_weight(crg_ctx* ctx, void* stream, const void* src, int src_dtype, int kind, int n_out, int n_in,
                               int ksize, void* dst_hi, void* dst_lo) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, src && dst_hi && n_out > 0 && n_in > 0, "pack_weight: bad arguments");
  if (kind != CRG_PACK_CONV) ksize = 1;
  CRG_REQUIRE(ctx, ksize == 1 || ksize == 3, "pack_weight: ksize %d unsupported", ksize);
  if (kind == CRG_PACK_GEGLU) CRG_REQUIRE(ctx, n_out % 32 == 0, "pack_weight: GEGLU needs n_out %% 32 == 0, got %d", n_out);
  const long total = (long)n_out * n_in * ksize * ksize;
  const int block = 256;
  const int grid = (int)((total + block - 1) / block < 4096 ? (total + block - 1) / block : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == CRG_F32)
    hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(block), 0, st, (const float*)src, (bf16*)dst_hi, (bf16*)dst_lo, kind, n_out, n_in, ksize, total);
  else if (src_dtype == CRG_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16>, dim3(grid), dim3(block), 0, st, (const bf16*)src, (bf16*)dst_hi, (bf16*)dst_lo, kind, n_out, n_in, ksize, total);
  else if (src_dtype == CRG_F16)
    hipLaunchKernelGGL(pack_kernel<_Float16>, dim3(grid), dim3(block), 0, st, (const _Float16*)src, (bf16*)dst_hi, (bf16*)dst_lo, kind, n_out, n_in, ksize, total);
  else
    return crg_fail(ctx, -22, "pack_weight: unsupported source dtype %d", src_dtype);
  CRG_CHECK_LAUNCH(ctx, "pack_weight");
  return 0;
}

extern "C" int crg_pack_geglu_bias(crg_ctx* ctx, void* stream, const float* src, int n_out2, float* dst) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, src && dst && n_out2 > 0 && n_out2 % 32 == 0, "pack_geglu_bias: n_out2 %% 32 != 0");
  hipLaunchKernelGGL(pack_geglu_bias_kernel, dim3((n_out2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, n_out2);
  CRG_CHECK_LAUNCH(ctx, "pack_geglu_bias");
  return 0;
}
