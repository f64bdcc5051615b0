// Small / HBM-bound helpers: thin-channel 3x3 conv (conv_in / conv_out), timestep embedding, SiLU,
// NCHW<->NHWC boundary transposes with dtype conversion, affine+clamp cast.
#include "crg_common.h"

namespace {

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

// ---- conv (3x3 or 1x1), Cin <= 8 (conv_in): thread = (PX consecutive pixels of one image row, 8 output channels); weights in
// LDS as [tap*Cin + ci][Cout] fp32; the 8 outputs are one 16-byte store.  Round 3: the kernel size is a template parameter (the
// runtime tap / ks, tap % ks and the 64-bit index divisions were more instructions than the FMAs), the weight image is filled by a
// coalesced linear read, the PX + 2 input pixels of a kernel row are loaded once and shared by its three taps, and the FMAs are packed
// pairs (v_pk_fma_f32): conv_in 4 -> 320 at 8 x 64 x 64 57 -> see DESIGN us.  Widths that are not a multiple of 4 run PX = 1.
template <typename XT, typename YT, int CI, int PX, int KS>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, YT* __restrict__ y, int N, int H, int W,
                                                             int Cout) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [KS*KS*CI][Cout]
  constexpr int taps = KS * KS, pad = KS / 2;
  // weight image: the source [Cout][CI][taps] is read LINEARLY (coalesced) and scattered into LDS
  for (int i = threadIdx.x; i < taps * CI * Cout; i += 256) {
    const int co = i / (taps * CI), r = i - co * (taps * CI);
    const int ci = r / taps, tap = r - ci * taps;
    wl[(tap * CI + ci) * Cout + co] = w[i];
  }
  __syncthreads();
  const int cg = Cout >> 3;  // 8-channel groups
  const int WQ = W / PX;
  const int total = N * H * WQ * cg;  // < 2^31 (checked by the host)
  constexpr int SEG = PX + KS - 1;    // input pixels of one kernel row
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int g = idx % cg;
    const int pq = idx / cg;
    const int wq = pq % WQ, r2 = pq / WQ;
    const int wo0 = wq * PX;
    const int ho = r2 % H;
    const int n = r2 / H;
    f32x2 acc[PX][4];
    {
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (bias) {
        b0 = *reinterpret_cast<const f32x4*>(bias + g * 8);
        b1 = *reinterpret_cast<const f32x4*>(bias + g * 8 + 4);
      }
#pragma unroll
      for (int px = 0; px < PX; ++px) {
        acc[px][0] = f32x2{b0[0], b0[1]}; acc[px][1] = f32x2{b0[2], b0[3]};
        acc[px][2] = f32x2{b1[0], b1[1]}; acc[px][3] = f32x2{b1[2], b1[3]};
      }
    }
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      const int hi = ho + kh - pad;
      if ((unsigned)hi >= (unsigned)H) continue;
      float xr[SEG][CI];
      const XT* xrow = x + ((long)n * H + hi) * W * CI;
#pragma unroll
      for (int sx = 0; sx < SEG; ++sx) {
        const int wi = wo0 + sx - pad;
        const bool ok = (unsigned)wi < (unsigned)W;
        const XT* xp = xrow + (ok ? wi : 0) * CI;
        if constexpr (CI == 4 && sizeof(XT) == 2) {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>(xp);
#pragma unroll
          for (int c = 0; c < 4; ++c) xr[sx][c] = ok ? (float)v[c] : 0.f;
        } else if constexpr (CI == 4 && sizeof(XT) == 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xp);
#pragma unroll
          for (int c = 0; c < 4; ++c) xr[sx][c] = ok ? v[c] : 0.f;
        } else {
#pragma unroll
          for (int c = 0; c < CI; ++c) xr[sx][c] = ok ? (float)xp[c] : 0.f;
        }
      }
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const float* wp = wl + ((kh * KS + kw) * CI + ci) * Cout + g * 8;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
          const f32x2 wa = {w0[0], w0[1]}, wb = {w0[2], w0[3]}, wc = {w1[0], w1[1]}, wd = {w1[2], w1[3]};
#pragma unroll
          for (int px = 0; px < PX; ++px) {
            const float xv = xr[px + kw][ci];
            const f32x2 xx = {xv, xv};
            acc[px][0] = __builtin_elementwise_fma(xx, wa, acc[px][0]);
            acc[px][1] = __builtin_elementwise_fma(xx, wb, acc[px][1]);
            acc[px][2] = __builtin_elementwise_fma(xx, wc, acc[px][2]);
            acc[px][3] = __builtin_elementwise_fma(xx, wd, acc[px][3]);
          }
        }
      }
    }
#pragma unroll
    for (int px = 0; px < PX; ++px) {
      crg_vec8<YT> o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o.set(2 * q, acc[px][q][0]);
        o.set(2 * q + 1, acc[px][q][1]);
      }
      o.store(y + (((long)n * H + ho) * W + wo0 + px) * Cout + g * 8);
    }
  }
}

// ---- conv (3x3 or 1x1), Cout <= 8 (conv_out): LPP lanes share one output pixel, each lane walks every LPP-th
// 8-channel chunk of Cin (consecutive lanes -> consecutive 16-byte chunks: coalesced rows), partial sums are
// combined with a shuffle tree.  Weights in LDS as [tap][Cin][CO] fp32, CO = 4 or 8 (the VAE's 3 output channels used to pay for 8),
// compile-time kernel size, packed FMAs.
template <typename XT, typename YT, int CO, int LPP, int KS>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, YT* __restrict__ y, int N, int H, int W,
                                                              int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [KS*KS][Cin][CO]
  constexpr int taps = KS * KS, pad = KS / 2;
  // weight image: zero-fill, then the source [Cout][Cin][taps] read linearly and scattered (see conv_small_cin_kernel)
  for (int i = threadIdx.x; i < taps * Cin * CO; i += 256) wl[i] = 0.f;
  __syncthreads();
  {
    const int per_co = Cin * taps;
    for (int i = threadIdx.x; i < Cout * per_co; i += 256) {
      const int co = i / per_co, r = i - co * per_co;
      const int ci = r / taps, tap = r - ci * taps;
      wl[(tap * Cin + ci) * CO + co] = w[i];
    }
  }
  __syncthreads();
  const int total = N * H * W;  // < 2^31 (checked by the host)
  const int sub = threadIdx.x % LPP;
  constexpr int ppb = 256 / LPP;  // pixels per block iteration
  for (int pix0 = blockIdx.x * ppb; pix0 < total; pix0 += gridDim.x * ppb) {
    const int pix = pix0 + threadIdx.x / LPP;
    const bool live = pix < total;
    const int pp = live ? pix : total - 1;
    const int wo = pp % W, r2 = pp / W;
    const int ho = r2 % H;
    const int n = r2 / H;
    f32x2 acc[CO / 2];
#pragma unroll
    for (int e = 0; e < CO / 2; ++e) acc[e] = f32x2{0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < taps; ++tap) {
      const int hi = ho + tap / KS - pad, wi = wo + tap % KS - pad;
      if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
      const XT* xp = x + (((long)n * H + hi) * W + wi) * Cin;
      const float* wt = wl + tap * Cin * CO;
      if ((Cin & 7) == 0) {
        for (int c0 = sub * 8; c0 < Cin; c0 += LPP * 8) {
          crg_vec8<XT> v;
          v.load(xp + c0);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xv = v.get(j);
            const f32x2 xx = {xv, xv};
            const float* wp = wt + (c0 + j) * CO;
#pragma unroll
            for (int e = 0; e < CO / 2; ++e) acc[e] = __builtin_elementwise_fma(xx, *reinterpret_cast<const f32x2*>(wp + 2 * e), acc[e]);
          }
        }
      } else {
        for (int ci = sub; ci < Cin; ci += LPP) {
          const float xv = ldf(xp + ci);
          const f32x2 xx = {xv, xv};
#pragma unroll
          for (int e = 0; e < CO / 2; ++e) acc[e] = __builtin_elementwise_fma(xx, *reinterpret_cast<const f32x2*>(wt + ci * CO + 2 * e), acc[e]);
        }
      }
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1)
#pragma unroll
      for (int e = 0; e < CO / 2; ++e) {
        acc[e][0] += __shfl_xor(acc[e][0], o);
        acc[e][1] += __shfl_xor(acc[e][1], o);
      }
    if (live && sub == 0) {
      YT* yp = y + (long)pix * Cout;
#pragma unroll
      for (int e = 0; e < CO; ++e)
        if (e < Cout) yp[e] = (YT)(acc[e >> 1][e & 1] + (bias ? bias[e] : 0.f));
    }
  }
}

template <typename YT>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, YT* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * dim) return;
  const int b = idx / dim, j = idx - b * dim;
  float v = 0.f;
  if (j < 2 * half) {
    const int i = j < half ? j : j - half;
    // freqs = exp(-ln(10000) * i / half) in fp32, as util.py:162-164 builds them
    const float f = expf(-9.210340371976184f * (float)i / (float)half);
    const float a = t[b] * f;
    v = j < half ? cosf(a) : sinf(a);
  }
  out[idx] = (YT)v;
}

template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (T)crg_silu_f((float)x[i]);
}

// NCHW -> NHWC via a 32x32 LDS tile transpose: src [N][C][HW], dst [N][HW][C]
template <typename ST, typename DT>
__global__ __launch_bounds__(256) void transpose_kernel(const ST* __restrict__ src, DT* __restrict__ dst, int R, int Cc) {
  // src viewed as [N][R][Cc], dst as [N][Cc][R]
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const ST* s = src + (long)n * R * Cc;
  DT* d = dst + (long)n * R * Cc;
  for (int i = ty; i < 32; i += 8) {
    const int rr = r0 + i, cc = c0 + tx;
    tile[i][tx] = (rr < R && cc < Cc) ? (float)s[(long)rr * Cc + cc] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int cc = c0 + i, rr = r0 + tx;
    if (rr < R && cc < Cc) d[(long)cc * R + rr] = (DT)tile[tx][i];
  }
}

template <typename ST, typename DT>
__global__ void affine_cast_kernel(const ST* __restrict__ x, DT* __restrict__ y, long n, float a, float b, float lo, float hi) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = (float)x[i] * a + b;
    v = fminf(fmaxf(v, lo), hi);
    y[i] = (DT)v;
  }
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, T* __restrict__ y, long n, float a, float b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (T)(a * (float)x[i] + b * (float)y[i]);
}

template <typename F>
int by_dtype2(crg_ctx* ctx, int sdt, int ddt, const char* what, F&& f) {
  if (sdt == CRG_BF16 && ddt == CRG_BF16) return f((const bf16*)nullptr, (bf16*)nullptr);
  if (sdt == CRG_BF16 && ddt == CRG_F32) return f((const bf16*)nullptr, (float*)nullptr);
  if (sdt == CRG_F32 && ddt == CRG_BF16) return f((const float*)nullptr, (bf16*)nullptr);
  if (sdt == CRG_F32 && ddt == CRG_F32) return f((const float*)nullptr, (float*)nullptr);
  return crg_fail(ctx, -22, "%s: unsupported dtype pair %d -> %d", what, sdt, ddt);
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// grid for kernels that first stage an `lds`-byte weight image per block: as many blocks as fit the chip at once
// (256 CUs x blocks/CU limited by the 160 KiB LDS), never more than the work needs
inline int grid_resident(long n_threads, size_t lds) {
  long per_cu = lds ? (long)((160 * 1024) / lds) : 8;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 8) per_cu = 8;
  long g = (n_threads + 255) / 256;
  const long cap = 256 * per_cu;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int crg_conv_small(crg_ctx* ctx, void* stream, const void* x, const float* w, const float* bias, void* y,
                              int N, int H, int W, int Cin, int Cout, int ksize, int x_dtype, int y_dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv_small: empty problem");
  CRG_REQUIRE(ctx, Cin <= 8 || Cout <= 8, "conv_small: needs Cin <= 8 or Cout <= 8 (Cin=%d Cout=%d)", Cin, Cout);
  hipStream_t st = (hipStream_t)stream;
  const int ks = ksize;
  CRG_REQUIRE(ctx, ks == 1 || ks == 3, "conv_small: ksize %d unsupported", ks);
  const double flops = 2.0 * N * H * W * (double)Cin * Cout * ks * ks;
  const double bytes = (double)N * H * W * (Cin * crg_dtype_size(x_dtype) + Cout * crg_dtype_size(y_dtype));
  crg_prof_scope ps(ctx, st, CRG_K_CONV_SMALL, flops, bytes);
  CRG_REQUIRE(ctx, (double)N * H * W * (Cout > 8 ? Cout / 8 : 1) < 2147483648.0, "conv_small: more than 2^31 work items");
  int rc;
  if (Cout <= 8 && (Cin > 8 || Cout <= Cin)) {
    const int CO = Cout <= 4 ? 4 : 8;
    const size_t lds = (size_t)ks * ks * Cin * CO * sizeof(float);
    CRG_REQUIRE(ctx, lds <= 160 * 1024, "conv_small: Cin=%d too large for the LDS weight image", Cin);
    if ((Cin & 7) == 0) CRG_REQUIRE(ctx, ((uintptr_t)x & 15) == 0, "conv_small: x must be 16-byte aligned");
    rc = by_dtype2(ctx, x_dtype, y_dtype, "conv_small", [&](auto* xs, auto* ys) {
      using XT = std::remove_const_t<std::remove_pointer_t<decltype(xs)>>;
      using YT = std::remove_pointer_t<decltype(ys)>;
      const long pixels = (long)N * H * W;
      const bool wide = Cin >= 64;  // eight lanes per pixel
      auto go = [&](auto kern) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid_resident(wide ? pixels * 8 : pixels, lds)), dim3(256), lds, st, (const XT*)x, w, bias, (YT*)y, N, H, W, Cin, Cout);
      };
      if (ks == 3) {
        if (CO == 4) wide ? go(conv_small_cout_kernel<XT, YT, 4, 8, 3>) : go(conv_small_cout_kernel<XT, YT, 4, 1, 3>);
        else wide ? go(conv_small_cout_kernel<XT, YT, 8, 8, 3>) : go(conv_small_cout_kernel<XT, YT, 8, 1, 3>);
      } else {
        if (CO == 4) wide ? go(conv_small_cout_kernel<XT, YT, 4, 8, 1>) : go(conv_small_cout_kernel<XT, YT, 4, 1, 1>);
        else wide ? go(conv_small_cout_kernel<XT, YT, 8, 8, 1>) : go(conv_small_cout_kernel<XT, YT, 8, 1, 1>);
      }
      return 0;
    });
  } else {
    CRG_REQUIRE(ctx, Cout % 8 == 0, "conv_small: Cout=%d must be a multiple of 8 when Cin <= 8", Cout);
    CRG_REQUIRE(ctx, Cin == 3 || Cin == 4 || Cin == 8, "conv_small: Cin=%d unsupported on the thin-input path (3, 4 or 8)", Cin);
    const size_t lds = (size_t)ks * ks * Cin * Cout * sizeof(float);
    CRG_REQUIRE(ctx, lds <= 160 * 1024, "conv_small: Cout=%d too large for the LDS weight image", Cout);
    rc = by_dtype2(ctx, x_dtype, y_dtype, "conv_small", [&](auto* xs, auto* ys) {
      using XT = std::remove_const_t<std::remove_pointer_t<decltype(xs)>>;
      using YT = std::remove_pointer_t<decltype(ys)>;
      const bool quad = W % 4 == 0 && Cin <= 4;  // 4 pixels per thread (8-channel inputs would need 32 more registers: one pixel)
      const dim3 grid(grid_resident((long)N * H * (quad ? W / 4 : W) * (Cout / 8), lds));
      auto go = [&](auto kern) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const XT*)x, w, bias, (YT*)y, N, H, W, Cout);
      };
      if (ks == 3) {
        if (Cin == 3) quad ? go(conv_small_cin_kernel<XT, YT, 3, 4, 3>) : go(conv_small_cin_kernel<XT, YT, 3, 1, 3>);
        else if (Cin == 4) quad ? go(conv_small_cin_kernel<XT, YT, 4, 4, 3>) : go(conv_small_cin_kernel<XT, YT, 4, 1, 3>);
        else go(conv_small_cin_kernel<XT, YT, 8, 1, 3>);
      } else {
        if (Cin == 3) quad ? go(conv_small_cin_kernel<XT, YT, 3, 4, 1>) : go(conv_small_cin_kernel<XT, YT, 3, 1, 1>);
        else if (Cin == 4) quad ? go(conv_small_cin_kernel<XT, YT, 4, 4, 1>) : go(conv_small_cin_kernel<XT, YT, 4, 1, 1>);
        else go(conv_small_cin_kernel<XT, YT, 8, 1, 1>);
      }
      return 0;
    });
  }
  if (rc) return rc;
  CRG_CHECK_LAUNCH(ctx, "conv_small");
  return 0;
}

extern "C" int crg_timestep_embedding(crg_ctx* ctx, void* stream, const float* t, void* out, int B, int dim, int dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, B > 0 && dim > 0, "timestep_embedding: empty");
  hipStream_t st = (hipStream_t)stream;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 4.0 * B * dim, (double)B * dim * crg_dtype_size(dtype));
  const int n = B * dim;
  if (dtype == CRG_F32)
    hipLaunchKernelGGL(timestep_embedding_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, st, t, (float*)out, B, dim);
  else if (dtype == CRG_BF16)
    hipLaunchKernelGGL(timestep_embedding_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, st, t, (bf16*)out, B, dim);
  else
    return crg_fail(ctx, -22, "timestep_embedding: dtype %d unsupported", dtype);
  CRG_CHECK_LAUNCH(ctx, "timestep_embedding");
  return 0;
}

extern "C" int crg_silu(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t n, int dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, n > 0, "silu: empty");
  hipStream_t st = (hipStream_t)stream;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 4.0 * n, 2.0 * n * crg_dtype_size(dtype));
  if (dtype == CRG_F32)
    hipLaunchKernelGGL(silu_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (float*)y, (long)n);
  else if (dtype == CRG_BF16)
    hipLaunchKernelGGL(silu_kernel<bf16>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, (long)n);
  else
    return crg_fail(ctx, -22, "silu: dtype %d unsupported", dtype);
  CRG_CHECK_LAUNCH(ctx, "silu");
  return 0;
}

static int transpose_impl(crg_ctx* ctx, void* stream, const void* src, void* dst, int N, int R, int Cc, int sdt, int ddt,
                          const char* what) {
  CRG_REQUIRE(ctx, N > 0 && R > 0 && Cc > 0, "%s: empty", what);
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)N * R * Cc;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 0.0, elems * (crg_dtype_size(sdt) + crg_dtype_size(ddt)));
  dim3 grid((Cc + 31) / 32, (R + 31) / 32, N);
  int rc = by_dtype2(ctx, sdt, ddt, what, [&](auto* xs, auto* ys) {
    using ST = std::remove_const_t<std::remove_pointer_t<decltype(xs)>>;
    using DT = std::remove_pointer_t<decltype(ys)>;
    hipLaunchKernelGGL((transpose_kernel<ST, DT>), grid, dim3(256), 0, st, (const ST*)src, (DT*)dst, R, Cc);
    return 0;
  });
  if (rc) return rc;
  CRG_CHECK_LAUNCH(ctx, what);
  return 0;
}

extern "C" int crg_nchw_to_nhwc(crg_ctx* ctx, void* stream, const void* src, void* dst, int N, int C, int HW, int src_dtype,
                                int dst_dtype) {
  if (!ctx) return -22;
  return transpose_impl(ctx, stream, src, dst, N, C, HW, src_dtype, dst_dtype, "nchw_to_nhwc");
}

extern "C" int crg_nhwc_to_nchw(crg_ctx* ctx, void* stream, const void* src, void* dst, int N, int C, int HW, int src_dtype,
                                int dst_dtype) {
  if (!ctx) return -22;
  return transpose_impl(ctx, stream, src, dst, N, HW, C, src_dtype, dst_dtype, "nhwc_to_nchw");
}

extern "C" int crg_affine_cast(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t n, float a, float b, float lo,
                               float hi, int src_dtype, int dst_dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, n > 0, "affine_cast: empty");
  hipStream_t st = (hipStream_t)stream;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 2.0 * n, (double)n * (crg_dtype_size(src_dtype) + crg_dtype_size(dst_dtype)));
  int rc = by_dtype2(ctx, src_dtype, dst_dtype, "affine_cast", [&](auto* xs, auto* ys) {
    using ST = std::remove_const_t<std::remove_pointer_t<decltype(xs)>>;
    using DT = std::remove_pointer_t<decltype(ys)>;
    hipLaunchKernelGGL((affine_cast_kernel<ST, DT>), dim3(grid_for(n)), dim3(256), 0, st, (const ST*)x, (DT*)y, (long)n, a, b, lo, hi);
    return 0;
  });
  if (rc) return rc;
  CRG_CHECK_LAUNCH(ctx, "affine_cast");
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ hi, bf16* __restrict__ lo, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
    bf16x8 h8, l8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = e < 4 ? a[e] : b[e - 4];
      const bf16 h = (bf16)f;
      h8[e] = h;
      l8[e] = (bf16)(f - (float)h);
    }
    *reinterpret_cast<bf16x8*>(hi + i * 8) = h8;
    *reinterpret_cast<bf16x8*>(lo + i * 8) = l8;
  }
}
}  // namespace

namespace {
// fp32 [pixels][C] -> MX planes (CRG_PREC_F16MX): thread = eight channels of one pixel
__global__ __launch_bounds__(256) void split_mx_kernel(const float* __restrict__ x, _Float16* __restrict__ x16, unsigned char* __restrict__ x8,
                                                       long n8, int C, float sh, float sl) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long e = i * 8;  // element index = pixel * C + c0
    const int c0 = (int)(e % C);
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + e), b = *reinterpret_cast<const f32x4*>(x + e + 4);
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    crg_store_mx8(f, x16 + e, x8 + 2 * (e - (c0 & 63)) + (c0 & 63), sh, sl);
  }
}
}  // namespace

extern "C" int crg_split_mx(crg_ctx* ctx, void* stream, const void* x, void* x16, void* x8, int64_t pixels, int C, int hi_log2, int lo_log2) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, pixels > 0 && C > 0 && C % 64 == 0, "split_mx: C=%d must be a positive multiple of 64", C);
  CRG_REQUIRE(ctx, (((uintptr_t)x | (uintptr_t)x16 | (uintptr_t)x8) & 15) == 0, "split_mx: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)pixels * C;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 6.0 * n, 8.0 * n);
  hipLaunchKernelGGL(split_mx_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const float*)x, (_Float16*)x16, (unsigned char*)x8, n / 8, C,
                     ldexpf(1.f, hi_log2), ldexpf(1.f, lo_log2));
  CRG_CHECK_LAUNCH(ctx, "split_mx");
  return 0;
}

extern "C" int crg_split_bf16(crg_ctx* ctx, void* stream, const void* x, void* hi, void* lo, int64_t n) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, n > 0 && n % 8 == 0, "split_bf16: element count %ld must be a positive multiple of 8", (long)n);
  CRG_REQUIRE(ctx, (((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0, "split_bf16: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 3.0 * n, 8.0 * n);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, st, (const float*)x, (bf16*)hi, (bf16*)lo, (long)(n / 8));
  CRG_CHECK_LAUNCH(ctx, "split_bf16");
  return 0;
}

namespace {
// One fused sampler step (see include/crg_hip.h); the arithmetic follows the reference's operation order one rounding at a
// time (no FMA contraction), so the result equals the chain of PyTorch elementwise kernels it replaces.
__global__ __launch_bounds__(256) void cfg_euler_step_kernel(float* __restrict__ x, const float* __restrict__ eps,
                                                             const float* __restrict__ noise, long n, float sigma, float dt,
                                                             float cfg, float noise_scale) {
#pragma clang fp contract(off)
  const float c_out = -sigma;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float xv = x[i];
    const float den_u = xv + eps[i] * c_out;
    const float den_c = xv + eps[n + i] * c_out;
    const float den = den_u + cfg * (den_c - den_u);
    const float d = (xv - den) / sigma;
    float xn = xv + d * dt;
    if (noise) xn = xn + noise[i] * noise_scale;
    x[i] = xn;
  }
}
}  // namespace

extern "C" int crg_cfg_euler_step(crg_ctx* ctx, void* stream, void* x, const void* eps, const void* noise, int64_t n, float sigma,
                                  float dt, float cfg_scale, float noise_scale) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, n > 0 && x && eps, "cfg_euler_step: empty input");
  CRG_REQUIRE(ctx, sigma > 0.f, "cfg_euler_step: sigma must be positive (got %g)", (double)sigma);
  hipStream_t st = (hipStream_t)stream;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 10.0 * n, 4.0 * n * (noise ? 5 : 4));
  hipLaunchKernelGGL(cfg_euler_step_kernel, dim3(grid_for(n)), dim3(256), 0, st, (float*)x, (const float*)eps, (const float*)noise, (long)n,
                     sigma, dt, cfg_scale, noise_scale);
  CRG_CHECK_LAUNCH(ctx, "cfg_euler_step");
  return 0;
}

extern "C" int crg_axpby(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t n, float a, float b, int dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, n > 0, "axpby: empty");
  hipStream_t st = (hipStream_t)stream;
  crg_prof_scope ps(ctx, st, CRG_K_ELEMENTWISE, 3.0 * n, 3.0 * n * crg_dtype_size(dtype));
  if (dtype == CRG_F32)
    hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (float*)y, (long)n, a, b);
  else if (dtype == CRG_BF16)
    hipLaunchKernelGGL(axpby_kernel<bf16>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, (long)n, a, b);
  else
    return crg_fail(ctx, -22, "axpby: dtype %d unsupported", dtype);
  CRG_CHECK_LAUNCH(ctx, "axpby");
  return 0;
}
