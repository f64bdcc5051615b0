// HBM-bound normalisation kernels: GroupNorm(+SiLU) over channels-last images, LayerNorm over token
// rows, row softmax.  All loads/stores are 16 bytes per lane (8 bf16 / 2x4 f32), rows are walked
// with consecutive lanes on consecutive channels (fully coalesced NHWC rows).
#include "crg_common.h"
#include <cstdlib>

namespace {

constexpr int GN_MAX_GROUPS = 32;
constexpr int GN_TILE_MAX_C = 1280;  // widest input whose tile partials the normalising launch folds itself (2 x C floats of LDS)

// ---- GroupNorm pass 1: partial (sum, sumsq) of (x - K_g) per (sample, chunk, group) --------------
// K_g = first element of group g in the sample's first pixel: the classic shifted-data form, so the
// fp32 accumulation error scales with the spread around K_g and not with |mean|^2 / var.
// grid (chunks, N); thread = (row lane, 8-channel column).
template <typename T>
__global__ __launch_bounds__(512) void gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, int C, int HW,
                                                       int groups, int rows_per_chunk, float* __restrict__ part,
                                                       float* __restrict__ kbuf) {
  __shared__ float shiftv[GN_MAX_GROUPS];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int tpr = C >> 3;                 // threads per row
  const int rpi = blockDim.x / tpr;       // rows per iteration
  const int t = threadIdx.x;
  const int col = t % tpr, rl = t / tpr;
  const int gs = C / groups;
  const int C2 = C - C1;
  if (t < groups) {
    const int c = t * gs;
    shiftv[t] = (c < C1) ? (float)x[(long)n * HW * C1 + c] : (float)x2[(long)n * HW * C2 + (c - C1)];
    if (chunk == 0) kbuf[n * groups + t] = shiftv[t];  // the apply pass must not re-read x (y may alias x)
  }
  __syncthreads();
  const int c0 = col * 8;
  float a1[8], a2[8], sh[8];
  int gid[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a1[e] = a2[e] = 0.f;
    gid[e] = (c0 + e) / gs;
    sh[e] = shiftv[gid[e] < groups ? gid[e] : 0];
  }
  if (rl < rpi) {
    const bool second = c0 >= C1;
    const T* base = second ? x2 + (long)n * HW * C2 + (c0 - C1) : x + (long)n * HW * C1 + c0;
    const int Cs = second ? C2 : C1;
    const int r_begin = chunk * rows_per_chunk;
    const int r_end = min(HW, r_begin + rows_per_chunk);
    int r = r_begin + rl;
    for (; r + 3 * rpi < r_end; r += 4 * rpi) {  // 4 independent 16-byte loads in flight per lane
      crg_vec8<T> v0, v1, v2, v3;
      v0.load(base + (long)r * Cs);
      v1.load(base + (long)(r + rpi) * Cs);
      v2.load(base + (long)(r + 2 * rpi) * Cs);
      v3.load(base + (long)(r + 3 * rpi) * Cs);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d0 = v0.get(e) - sh[e], d1 = v1.get(e) - sh[e], d2 = v2.get(e) - sh[e], d3 = v3.get(e) - sh[e];
        a1[e] += (d0 + d1) + (d2 + d3);
        a2[e] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    for (; r < r_end; r += rpi) {
      crg_vec8<T> v;
      v.load(base + (long)r * Cs);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v.get(e) - sh[e];
        a1[e] += d;
        a2[e] += d * d;
      }
    }
  }
  // Deterministic block reduction (no float atomics: results must not depend on arrival order): every lane parks its
  // 8 per-channel partials in LDS, then one lane per group adds its group's channels x row-lanes in a fixed order.
  __shared__ __attribute__((aligned(16))) float red[512 * 16];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[t * 16 + e] = (rl < rpi) ? a1[e] : 0.f;
    red[t * 16 + 8 + e] = (rl < rpi) ? a2[e] : 0.f;
  }
  __syncthreads();
  // two short fixed-order phases instead of one long serial chain per group: (1) one thread per 8-channel column adds the
  // row lanes, (2) one thread per group adds its channels
  float c1[8], c2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) c1[e] = c2[e] = 0.f;
  if (t < tpr) {
    for (int r2 = 0; r2 < rpi; ++r2) {
      const f32x4* q = reinterpret_cast<const f32x4*>(red + (r2 * tpr + t) * 16);
      const f32x4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c1[e] += q0[e];
        c1[4 + e] += q1[e];
        c2[e] += q2[e];
        c2[4 + e] += q3[e];
      }
    }
  }
  __syncthreads();  // all row-lane partials consumed: the first tpr slots are rewritten with the column sums
  if (t < tpr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[t * 16 + e] = c1[e];
      red[t * 16 + 8 + e] = c2[e];
    }
  }
  __syncthreads();
  if (t < groups) {
    float b1 = 0.f, b2 = 0.f;
    for (int c = t * gs; c < (t + 1) * gs; ++c) {
      const float* q = red + (c >> 3) * 16 + (c & 7);
      b1 += q[0];
      b2 += q[8];
    }
    float* o = part + (((long)n * gridDim.x + chunk) * groups + t) * 2;
    o[0] = b1;
    o[1] = b2;
  }
}

// ---- GroupNorm pass 2: finalise statistics (fp64 combine) and apply affine (+SiLU) ---------------
template <typename T>
__global__ __launch_bounds__(512) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, int C, int HW,
                                                       int groups, int rows_per_block, int n_chunks,
                                                       const float* __restrict__ part, const float* __restrict__ kbuf,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu, T* __restrict__ y,
                                                       bf16* __restrict__ yh, bf16* __restrict__ yl, unsigned char* __restrict__ y8 = nullptr,
                                                       float mx_sh = 0.f, float mx_sl = 0.f, int rows2 = 0) {
  // y8 (fp32 inputs only): MX planes (CRG_PREC_F16MX) instead - yh is then the fp16 plane, y8 the e4m3 pair plane (crg_store_mx8)
  // yh / yl (fp32 inputs only): write the result as two bf16 planes, hi = bf16(f) and lo = bf16(f - hi), instead of y - the
  // operand format of the split-bf16 (fp32-class) LDS-DMA conv, so the split costs no extra pass over the tensor
  __shared__ float meanv[GN_MAX_GROUPS], rstdv[GN_MAX_GROUPS];
  const int n = blockIdx.y;
  const int t = threadIdx.x;
  const int gs = C / groups;
  const int C2 = C - C1;
  if (n_chunks < -1) {
    // TILE partials from the producer(s) (crg_conv_args.gn_stats_rows >= 64: per output tile of -n_chunks (source 2: rows2) rows and per
    // channel the sum and the sum of squares, planes at the 32-row layout's stride): few enough - HW / rows x C x 2 floats per sample, 41 KB
    // at 8 x 64 x 64 x 320 on 256-row tiles - that every block folds its sample's itself and the gn_finalize launch disappears.
    // part = source 1 (channels < C1), kbuf = source 2; fp32 per channel, fp64 across a group's channels and for the moments.
    // thread = (four channels, tile subset): 16-byte loads, coalesced along a tile's partial row, every load of the block in flight at once
    // (a first form with one channel per thread and a second round for C > 256 put two load latencies in front of the block: +3.4 us per
    // launch against the 5.4 us finalise launch it replaces); then thread = group over the LDS sums
    __shared__ __attribute__((aligned(16))) float csum[2][2048];
    const int rows1 = -n_chunks;
    const int T1 = HW / rows1, T2 = rows2 > 0 ? HW / rows2 : 0;
    const long plane1 = (long)gridDim.y * (HW >> 5) * C1, plane2 = (long)gridDim.y * (HW >> 5) * C2;
    const int nq = C >> 2;
    int TS = blockDim.x / nq;
    if (TS < 1) TS = 1;
    for (int w = t; w < nq * TS; w += blockDim.x) {
      const int ts = w / nq, c = (w - ts * nq) * 4;
      const bool second = c >= C1;
      const int Cs = second ? C2 : C1, nt = second ? T2 : T1;
      const long plane = second ? plane2 : plane1;
      const float* q = (second ? kbuf + (c - C1) : part + c) + (long)n * nt * Cs;
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
#pragma unroll 4
      for (int i = ts; i < nt; i += TS) {
        a += *reinterpret_cast<const f32x4*>(q + (long)i * Cs);
        b += *reinterpret_cast<const f32x4*>(q + (long)i * Cs + plane);
      }
      *reinterpret_cast<f32x4*>(&csum[0][ts * C + c]) = a;
      *reinterpret_cast<f32x4*>(&csum[1][ts * C + c]) = b;
    }
    __syncthreads();
    if (t < groups) {
      double da = 0.0, db = 0.0;
      for (int ts = 0; ts < TS; ++ts)
        for (int cl = 0; cl < gs; ++cl) {
          da += (double)csum[0][ts * C + t * gs + cl];
          db += (double)csum[1][ts * C + t * gs + cl];
        }
      const double cnt = (double)HW * gs;
      const double md = da / cnt;
      double var = db / cnt - md * md;
      if (var < 0.0) var = 0.0;
      meanv[t] = (float)md;
      rstdv[t] = (float)(1.0 / sqrt(var + (double)eps));
    }
  } else if (n_chunks < 0) {
    // statistics already final (gn_finalize_kernel): part = [N][groups] (mean, rstd)
    if (t < groups) {
      meanv[t] = part[((long)n * groups + t) * 2];
      rstdv[t] = part[((long)n * groups + t) * 2 + 1];
    }
  } else {
    // 8 lanes per group sum the chunk partials in fp64, then a 3-step shuffle reduction
    const int g = t >> 3, sub = t & 7;
    double a = 0.0, b = 0.0;
    if (g < groups) {
      for (int ch = sub; ch < n_chunks; ch += 8) {
        const float* pp = part + (((long)n * n_chunks + ch) * groups + g) * 2;
        a += (double)pp[0];
        b += (double)pp[1];
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    if (g < groups && sub == 0) {
      const double K = (double)kbuf[n * groups + g];
      const double cnt = (double)HW * gs;
      const double md = a / cnt;
      double var = b / cnt - md * md;
      if (var < 0.0) var = 0.0;
      meanv[g] = (float)(K + md);
      rstdv[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int tpr = C >> 3;
  const int rpi = blockDim.x / tpr;
  const int col = t % tpr, rl = t / tpr;
  if (rl >= rpi) return;
  const int c0 = col * 8;
  float sc[8], sf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (c0 + e) / gs;
    const float ga = gamma[c0 + e], be = beta[c0 + e];
    sc[e] = rstdv[g] * ga;
    sf[e] = be - meanv[g] * rstdv[g] * ga;
  }
  const bool second = c0 >= C1;
  const T* base = second ? x2 + (long)n * HW * C2 + (c0 - C1) : x + (long)n * HW * C1 + c0;
  const int Cs = second ? C2 : C1;
  T* yb = y + (long)n * HW * C + c0;
  const long plane0 = (long)n * HW * C + c0;
  auto emit = [&](const float (&f)[8], long roff) {
    if (y8) {
      const long e = plane0 + roff;  // element index = pixel * C + c0
      crg_store_mx8(f, reinterpret_cast<_Float16*>(yh) + e, y8 + 2 * (e - (c0 & 63)) + (c0 & 63), mx_sh, mx_sl);
    } else if (yh) {
      bf16x8 h8, l8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bf16 h = (bf16)f[e];
        h8[e] = h;
        l8[e] = (bf16)(f[e] - (float)h);
      }
      *reinterpret_cast<bf16x8*>(yh + plane0 + roff) = h8;
      *reinterpret_cast<bf16x8*>(yl + plane0 + roff) = l8;
    } else {
      crg_vec8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, f[e]);
      o.store(yb + roff);
    }
  };
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(HW, r_begin + rows_per_block);
  int r = r_begin + rl;
  for (; r + rpi < r_end; r += 2 * rpi) {
    crg_vec8<T> v0, v1;
    float g0[8], g1[8];
    v0.load(base + (long)r * Cs);
    v1.load(base + (long)(r + rpi) * Cs);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f0 = v0.get(e) * sc[e] + sf[e], f1 = v1.get(e) * sc[e] + sf[e];
      if (silu) {
        f0 = crg_silu_f(f0);
        f1 = crg_silu_f(f1);
      }
      g0[e] = f0;
      g1[e] = f1;
    }
    emit(g0, (long)r * C);
    emit(g1, (long)(r + rpi) * C);
  }
  for (; r < r_end; r += rpi) {
    crg_vec8<T> v;
    float g0[8];
    v.load(base + (long)r * Cs);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = v.get(e) * sc[e] + sf[e];
      if (silu) f = crg_silu_f(f);
      g0[e] = f;
    }
    emit(g0, (long)r * C);
  }
}

// ---- GroupNorm statistics from the PRODUCER's side channel -------------------------------------------------------------
// The conv / GEMM that wrote x also wrote, per 32-row block and channel, the sum and the sum of squares of its (rounded) outputs
// (crg_conv_args.gn_stats / crg_gemm_args.gn_stats: planes [2][rows / 32][C]).  One block per (sample, group) folds its group's
// channels x the sample's row blocks in a fixed order (thread-strided fp32 partials, fp64 across the threads) into (mean, rstd):
// the statistics pass over the tensor itself (gn_stats_kernel: a full read of x) disappears.  x2 != null: channels >= C1 of the
// virtual concat come from the second producer's planes.  No float atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ st1, const float* __restrict__ st2, int C1, int C,
                                                           int HW, int groups, long plane1, long plane2, float eps,
                                                           float* __restrict__ mr, int rows1 = 32, int rows2 = 32) {
  // block = (group g, sample n); thread = (row-block lane rl, channel cl of the group): a fixed channel per thread, no division
  // in the loop, four independent row blocks (eight loads) in flight per thread
  __shared__ double red[2][4];
  const int g = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int gs = C / groups, C2 = C - C1;
  const int nrl = 256 / gs;          // row-block lanes (gs <= 128: at least two)
  const int rl = t / gs, cl = t - rl * gs;
  const int c = g * gs + cl;
  float a = 0.f, b = 0.f;
  if (rl < nrl) {
    const bool second = c >= C1;
    const int Cs = second ? C2 : C1;
    const int nrb = HW / (second ? rows2 : rows1);  // partials per sample of this channel's producer (32-row blocks, or its tiles)
    const float* q = (second ? st2 + (c - C1) : st1 + c) + (long)n * nrb * Cs;
    const long plane = second ? plane2 : plane1;
    int rb = rl;
    for (; rb + 3 * nrl < nrb; rb += 4 * nrl) {
      const float* q0 = q + (long)rb * Cs;
      const float* q1 = q0 + (long)nrl * Cs;
      const float* q2 = q1 + (long)nrl * Cs;
      const float* q3 = q2 + (long)nrl * Cs;
      const float a0 = q0[0], a1 = q1[0], a2 = q2[0], a3 = q3[0];
      const float b0 = q0[plane], b1 = q1[plane], b2 = q2[plane], b3 = q3[plane];
      a += (a0 + a1) + (a2 + a3);
      b += (b0 + b1) + (b2 + b3);
    }
    for (; rb < nrb; rb += nrl) {
      a += q[(long)rb * Cs];
      b += q[(long)rb * Cs + plane];
    }
  }
  double da = a, db = b;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    da += __shfl_xor(da, o);
    db += __shfl_xor(db, o);
  }
  if ((t & 63) == 0) {
    red[0][t >> 6] = da;
    red[1][t >> 6] = db;
  }
  __syncthreads();
  if (t == 0) {
    da = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    db = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double cnt = (double)HW * gs;
    const double mean = da / cnt;
    double var = db / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[((long)n * groups + g) * 2] = (float)mean;
    mr[((long)n * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// (A one-launch form for the 64x64 / 32x32 levels - rows kept in registers, per-sample arrival counters between the blocks,
//  write-through partials, no release fence - was built and measured in round 2: correct and bitwise deterministic, but
//  34.9 us against 18.2 us for the stats + apply pair on 8 x 320 x 64 x 64: waiting for the other blocks of the sample costs
//  far more than the kernel boundary it replaces, exactly as the guide's price list says (barrier-counter 12.7 us at two
//  workgroups per CU vs boundary 1.7 us).  Removed; the pair below stays.)

// ---- GroupNorm, small images (8x8 / 16x16 UNet levels): ONE launch, one block per (sample, group) ----
// The group's HW x gs slab (gs % 8 == 0, at most 256 * VPT 16-byte vectors) is read once into registers, reduced
// (shifted sums, wave shuffles + a fixed-order cross-wave sum: deterministic), normalised and written back.  Replaces the
// two-pass stats/apply pair where that pair is pure launch + latency (a 1.3 MB tensor took ~15 us in two kernels).
template <typename T, int VPT>
__global__ __launch_bounds__(256) void gn_small_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, int C, int HW,
                                                       int groups, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int silu, T* __restrict__ y) {
  __shared__ float red[2][4];
  const int n = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  const int gs = C / groups, wv = gs >> 3;
  const int c_beg = g * gs;
  const int C2 = C - C1;
  const bool second = c_beg >= C1;  // C1 % gs == 0 is checked by the launcher: a group never straddles the two inputs
  const T* base = second ? x2 + (long)n * HW * C2 + (c_beg - C1) : x + (long)n * HW * C1 + c_beg;
  const int Cs = second ? C2 : C1;
  const int total = HW * wv;
  const float shift = (float)base[0];
  crg_vec8<T> v[VPT];
  int row[VPT], cv[VPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = t + 256 * k;
    row[k] = i / wv;
    cv[k] = i - row[k] * wv;
    if (i < total) v[k].load(base + (long)row[k] * Cs + cv[k] * 8);
  }
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if (t + 256 * k < total) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[k].get(e) - shift;
        s1 += d;
        s2 += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((t & 63) == 0) {
    red[0][t >> 6] = s1;
    red[1][t >> 6] = s2;
  }
  __syncthreads();
  const double a = ((double)red[0][0] + (double)red[0][1]) + ((double)red[0][2] + (double)red[0][3]);
  const double b = ((double)red[1][0] + (double)red[1][1]) + ((double)red[1][2] + (double)red[1][3]);
  const double cnt = (double)HW * gs;
  const double md = a / cnt;
  double var = b / cnt - md * md;
  if (var < 0.0) var = 0.0;
  const float mean = (float)((double)shift + md);
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  T* yb = y + (long)n * HW * C + c_beg;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if (t + 256 * k < total) {
      const int c = c_beg + cv[k] * 8;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c), g1 = *reinterpret_cast<const f32x4*>(gamma + c + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c), b1 = *reinterpret_cast<const f32x4*>(beta + c + 4);
      crg_vec8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
        float f = (v[k].get(e) - mean) * (rstd * ga) + be;
        if (silu) f = crg_silu_f(f);
        o.set(e, f);
      }
      o.store(yb + (long)row[k] * C + cv[k] * 8);
    }
  }
}

// ---- LayerNorm: RW rows per wave and iteration, rows kept in registers (dim <= 64 lanes * 8 * NCH, NCH <= LN_MAXC) ----------
// One row per wave and one wave-round per launch left a single 16-byte load per lane in flight and ran the launch as "everybody
// loads, then everybody stores" (2.4 TB/s at dim 320).  Waves now walk the rows in a grid-stride loop, RW rows per step, and
// request the next step's rows before reducing the current ones, so the read and the write streams overlap.
constexpr int LN_MAXC = 4;  // dim <= 2048
template <typename T, int NCH, int RW>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, long rows, int dim,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const long stride = (long)gridDim.x * 4 * RW;
  long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= rows) return;
  const int nch = dim >> 3;
  f32x4 g0[NCH], g1[NCH], b0[NCH], b1[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i < nch ? lane + 64 * i : 0;
    g0[i] = *reinterpret_cast<const f32x4*>(gamma + ch * 8);
    g1[i] = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
    b0[i] = *reinterpret_cast<const f32x4*>(beta + ch * 8);
    b1[i] = *reinterpret_cast<const f32x4*>(beta + ch * 8 + 4);
  }
  crg_vec8<T> v[RW][NCH], nx[RW][NCH];
  auto fetch = [&](crg_vec8<T> (&dst)[RW][NCH], long r0) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const long row = r0 + r < rows ? r0 + r : rows - 1;  // clamped: a tail step re-reads the last row, stores are guarded
      const T* xr = x + row * dim;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < nch) dst[r][i].load(xr + (lane + 64 * i) * 8);
    }
  };
  fetch(v, row0);
  for (; row0 < rows; row0 += stride) {
    const bool more = row0 + stride < rows;  // wave-uniform
    if (more) fetch(nx, row0 + stride);
    float s[RW], q[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) s[r] += v[r][i].get(e);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < RW; ++r) s[r] += __shfl_xor(s[r], o);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float mean = s[r] / (float)dim;
      s[r] = mean;
      q[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[r][i].get(e) - mean;
            q[r] += d * d;
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < RW; ++r) q[r] += __shfl_xor(q[r], o);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      if (row0 + r < rows) {
        const float mean = s[r], rstd = rsqrtf(q[r] / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int ch = lane + 64 * i;
          if (ch < nch) {
            crg_vec8<T> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float ga = e < 4 ? g0[i][e] : g1[i][e - 4], be = e < 4 ? b0[i][e] : b1[i][e - 4];
              o.set(e, (v[r][i].get(e) - mean) * rstd * ga + be);
            }
            o.store(y + (row0 + r) * dim + ch * 8);
          }
        }
      }
    }
    if (more) {
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[r][i] = nx[r][i];
    }
  }
}

template <typename T>
static void launch_layernorm(hipStream_t st, const T* x, const float* gamma, const float* beta, T* y, long rows, int dim, float eps) {
  const int nchunks = (dim / 8 + 63) / 64;
  // few rows: one row per wave so that the grid still covers the chip; many rows: at most 8 blocks per CU, the waves loop
  const bool small = rows < 4 * 256 * 8;
  static const int bpc = getenv("CRG_LN_BPC") ? atoi(getenv("CRG_LN_BPC")) : 8;  // dev knob: resident blocks per CU
#define CRG_LN(NCH, RW)                                                                                                     \
  do {                                                                                                                      \
    long blocks = (rows + 4 * RW - 1) / (4 * RW);                                                                           \
    if (blocks > 256 * bpc) blocks = 256 * bpc;                                                                                 \
    hipLaunchKernelGGL((layernorm_kernel<T, NCH, RW>), dim3((unsigned)blocks), dim3(256), 0, st, x, gamma, beta, y, rows, dim, eps); \
  } while (0)
  if (nchunks == 1) { if (small) CRG_LN(1, 1); else CRG_LN(1, 2); }
  else if (nchunks == 2) { if (small) CRG_LN(2, 1); else CRG_LN(2, 2); }
  else if (nchunks == 3) { if (small) CRG_LN(3, 1); else CRG_LN(3, 1); }
  else CRG_LN(4, 1);
#undef CRG_LN
}

// ---- row softmax: one block per row, y = softmax(x * scale) -----------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int cols, long ld, float scale) {
  __shared__ float red[8];
  const long row = blockIdx.x;
  const T* xr = x + row * ld;
  T* yr = y + row * ld;
  const int t = threadIdx.x;
  float mx = -INFINITY;
  for (int c = t; c < cols; c += 256) mx = fmaxf(mx, (float)xr[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = t; c < cols; c += 256) s += __expf((float)xr[c] * scale - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) red[4 + (t >> 6)] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = t; c < cols; c += 256) yr[c] = (T)(__expf((float)xr[c] * scale - mx) * inv);
}

}  // namespace

static int groupnorm_impl(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* gamma,
                          const float* beta, void* y, bf16* yh, bf16* yl, int N, int HW, int C, int groups, float eps, int fuse_silu,
                          int dtype, unsigned char* y8 = nullptr, float mx_sh = 0.f, float mx_sl = 0.f) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, N > 0 && HW > 0 && C > 0, "groupnorm: empty input");
  CRG_REQUIRE(ctx, groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, "groupnorm: groups=%d C=%d unsupported", groups, C);
  CRG_REQUIRE(ctx, C % 8 == 0 && (C >> 3) <= 512, "groupnorm: C=%d must be a multiple of 8 and <= 4096", C);
  const int threads = (C >> 3) <= 256 ? 256 : 512;
  if (!x2) C1 = C;
  CRG_REQUIRE(ctx, C1 > 0 && C1 <= C && C1 % 8 == 0 && (C - C1) % 8 == 0, "groupnorm: concat split C1=%d of C=%d unsupported", C1, C);
  CRG_REQUIRE(ctx, dtype == CRG_BF16 || dtype == CRG_F32, "groupnorm: dtype %d unsupported", dtype);
  // small images: single-launch path (one block per (sample, group), slab held in registers)
  {
    const int gs = C / groups;
    const long vecs = (long)HW * (gs >> 3);
    if (!yh && gs % 8 == 0 && C1 % gs == 0 && vecs <= 2560 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0) {
      hipStream_t st = (hipStream_t)stream;
      const double elems = (double)N * HW * C;
      crg_prof_scope ps(ctx, st, CRG_K_GN_APPLY, 8.0 * elems, elems * crg_dtype_size(dtype) * 2);
      dim3 grid(groups, N);
#define CRG_GN_SMALL(TT, V) hipLaunchKernelGGL((gn_small_kernel<TT, V>), grid, dim3(256), 0, st, (const TT*)x, (const TT*)x2, C1, C, HW, groups, gamma, beta, eps, fuse_silu, (TT*)y)
      if (dtype == CRG_BF16) {
        if (vecs <= 512) CRG_GN_SMALL(bf16, 2);
        else if (vecs <= 1280) CRG_GN_SMALL(bf16, 5);
        else CRG_GN_SMALL(bf16, 10);
      } else {
        if (vecs <= 512) CRG_GN_SMALL(float, 2);
        else if (vecs <= 1280) CRG_GN_SMALL(float, 5);
        else CRG_GN_SMALL(float, 10);
      }
#undef CRG_GN_SMALL
      CRG_CHECK_LAUNCH(ctx, "groupnorm(small)");
      return 0;
    }
  }
  // rows per block: ~64 rows each (the per-block reduction epilogue is the fixed cost), at most 256 chunks per sample (the apply prologue re-reduces them),
  // and at least ~512 blocks over the chip when the image is large enough
  int chunks = HW / 64;
  if (chunks < 1) chunks = 1;
  if (chunks > 256) chunks = 256;
  while ((long)chunks * N < 512 && chunks * 8 <= HW && chunks < 256) chunks *= 2;
  if (chunks < 1) chunks = 1;
  const int rpc = (HW + chunks - 1) / chunks;
  chunks = (HW + rpc - 1) / rpc;
  float* part = (float*)crg_scratch(ctx, ((size_t)N * chunks * groups * 2 + (size_t)N * groups) * sizeof(float));
  if (!part) return crg_fail(ctx, -12, "groupnorm: out of scratch");
  float* kbuf = part + (size_t)N * chunks * groups * 2;
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)N * HW * C;
  const size_t es = crg_dtype_size(dtype);
  dim3 grid(chunks, N);
  {
    crg_prof_scope ps(ctx, st, CRG_K_GN_STATS, 3.0 * elems, elems * es);
    if (dtype == CRG_BF16)
      hipLaunchKernelGGL(gn_stats_kernel<bf16>, grid, dim3(threads), 0, st, (const bf16*)x, (const bf16*)x2, C1, C, HW, groups, rpc, part, kbuf);
    else
      hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(threads), 0, st, (const float*)x, (const float*)x2, C1, C, HW, groups, rpc, part, kbuf);
  }
  {
    crg_prof_scope ps(ctx, st, CRG_K_GN_APPLY, 5.0 * elems, elems * es * 2);
    if (dtype == CRG_BF16)
      hipLaunchKernelGGL(gn_apply_kernel<bf16>, grid, dim3(threads), 0, st, (const bf16*)x, (const bf16*)x2, C1, C, HW, groups, rpc, chunks,
                         part, kbuf, gamma, beta, eps, fuse_silu, (bf16*)y, (bf16*)nullptr, (bf16*)nullptr);
    else
      hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(threads), 0, st, (const float*)x, (const float*)x2, C1, C, HW, groups, rpc, chunks,
                         part, kbuf, gamma, beta, eps, fuse_silu, (float*)y, yh, yl, y8, mx_sh, mx_sl);
  }
  CRG_CHECK_LAUNCH(ctx, "groupnorm");
  return 0;
}

extern "C" int crg_groupnorm(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* gamma,
                             const float* beta, void* y, int N, int HW, int C, int groups, float eps, int fuse_silu,
                             int dtype) {
  return groupnorm_impl(ctx, stream, x, x2, C1, gamma, beta, y, nullptr, nullptr, N, HW, C, groups, eps, fuse_silu, dtype);
}

extern "C" int crg_groupnorm_pre(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* stats1,
                                 const float* stats2, const float* gamma, const float* beta, void* y, int N, int HW, int C,
                                 int groups, float eps, int fuse_silu, int dtype, int rows1, int rows2) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, N > 0 && HW > 0 && C > 0, "groupnorm_pre: empty input");
  CRG_REQUIRE(ctx, dtype == CRG_BF16, "groupnorm_pre: the statistics side channel is a bf16-path feature (dtype %d)", dtype);
  CRG_REQUIRE(ctx, groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, "groupnorm_pre: groups=%d C=%d unsupported", groups, C);
  CRG_REQUIRE(ctx, C % 8 == 0 && (C >> 3) <= 512, "groupnorm_pre: C=%d must be a multiple of 8 and <= 4096", C);
  CRG_REQUIRE(ctx, HW % 32 == 0, "groupnorm_pre: HW=%d must be a multiple of the 32-row statistics blocks", HW);
  CRG_REQUIRE(ctx, C / groups <= 128, "groupnorm_pre: group size %d unsupported (<= 128)", C / groups);
  if (!x2) C1 = C;
  CRG_REQUIRE(ctx, C1 > 0 && C1 <= C && C1 % 8 == 0 && (C - C1) % 8 == 0, "groupnorm_pre: concat split C1=%d of C=%d unsupported", C1, C);
  CRG_REQUIRE(ctx, stats1 && (x2 == nullptr) == (stats2 == nullptr), "groupnorm_pre: one statistics buffer per input");
  if (rows1 <= 0) rows1 = 32;
  if (!x2) rows2 = 0;
  else if (rows2 <= 0) rows2 = 32;
  CRG_REQUIRE(ctx, rows1 % 32 == 0 && HW % rows1 == 0 && (!x2 || (rows2 % 32 == 0 && HW % rows2 == 0)),
              "groupnorm_pre: rows per partial (%d, %d) must be multiples of 32 that divide HW=%d", rows1, rows2, HW);
  const long rbs = (long)N * (HW >> 5);
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)N * HW * C;
  // tile partials on every input (crg_conv_args.gn_stats_rows / crg_gemm_args.gn_stats_rows >= 64): each apply block folds its sample's
  // itself - no finalise launch.  32-row partials (327 KB per sample at 8 x 64 x 64 x 320) keep the finalise kernel.
  // ... while a sample's partials stay small next to the block's own share of the tensor (80 KB: 16 tiles x 640 channels; a 128 x 128 level
  // has 64 tiles per sample - 490 KB in front of every block of a 960-channel concat - and keeps the finalise launch)
  const long fold_floats = (long)(HW / rows1) * C1 + (x2 ? (long)(HW / rows2) * (C - C1) : 0);
  const bool fold_in_apply = rows1 >= 64 && (!x2 || rows2 >= 64) && C <= GN_TILE_MAX_C && fold_floats <= 10240 &&
                             (((uintptr_t)stats1 | (uintptr_t)stats2) & 15) == 0;  // (16-byte loads of the partial rows)
  float* mr = nullptr;
  if (!fold_in_apply) {
    mr = (float*)crg_scratch(ctx, (size_t)N * groups * 2 * sizeof(float));
    if (!mr) return crg_fail(ctx, -12, "groupnorm_pre: out of scratch");
    crg_prof_scope ps(ctx, st, CRG_K_GN_STATS, 2.0 * rbs * C, 8.0 * rbs * C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, st, stats1, stats2, C1, C, HW, groups, rbs * C1, rbs * (C - C1), eps, mr,
                       rows1, x2 ? rows2 : 32);
  }
  {
    const int threads = (C >> 3) <= 256 ? 256 : 512;
    int chunks = HW / 64;
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    while ((long)chunks * N < 512 && chunks * 8 <= HW && chunks < 256) chunks *= 2;
    const int rpc = (HW + chunks - 1) / chunks;
    chunks = (HW + rpc - 1) / rpc;
    crg_prof_scope ps(ctx, st, CRG_K_GN_APPLY, 5.0 * elems, elems * 4.0);
    if (fold_in_apply)
      hipLaunchKernelGGL(gn_apply_kernel<bf16>, dim3(chunks, N), dim3(threads), 0, st, (const bf16*)x, (const bf16*)x2, C1, C, HW, groups, rpc, -rows1,
                         stats1, stats2, gamma, beta, eps, fuse_silu, (bf16*)y, (bf16*)nullptr, (bf16*)nullptr, (unsigned char*)nullptr, 0.f, 0.f, rows2);
    else
      hipLaunchKernelGGL(gn_apply_kernel<bf16>, dim3(chunks, N), dim3(threads), 0, st, (const bf16*)x, (const bf16*)x2, C1, C, HW, groups, rpc, -1,
                         mr, (const float*)nullptr, gamma, beta, eps, fuse_silu, (bf16*)y, (bf16*)nullptr, (bf16*)nullptr);
  }
  CRG_CHECK_LAUNCH(ctx, "groupnorm_pre");
  return 0;
}

extern "C" int crg_groupnorm_pre_split(crg_ctx* ctx, void* stream, const void* x, const float* stats, const float* gamma, const float* beta,
                                       void* y_hi, void* y_lo, int N, int HW, int C, int groups, float eps, int fuse_silu) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, N > 0 && HW > 0 && C > 0 && x && stats, "groupnorm_pre_split: empty input");
  CRG_REQUIRE(ctx, y_hi && y_lo && (((uintptr_t)y_hi | (uintptr_t)y_lo | (uintptr_t)x | (uintptr_t)stats) & 15) == 0, "groupnorm_pre_split: pointers must be 16-byte aligned");
  CRG_REQUIRE(ctx, groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, "groupnorm_pre_split: groups=%d C=%d unsupported", groups, C);
  CRG_REQUIRE(ctx, C % 8 == 0 && (C >> 3) <= 512, "groupnorm_pre_split: C=%d must be a multiple of 8 and <= 4096", C);
  CRG_REQUIRE(ctx, HW % 32 == 0, "groupnorm_pre_split: HW=%d must be a multiple of the 32-row statistics blocks", HW);
  CRG_REQUIRE(ctx, C / groups <= 128, "groupnorm_pre_split: group size %d unsupported (<= 128)", C / groups);
  const long rbs = (long)N * (HW >> 5);
  float* mr = (float*)crg_scratch(ctx, (size_t)N * groups * 2 * sizeof(float));
  if (!mr) return crg_fail(ctx, -12, "groupnorm_pre_split: out of scratch");
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)N * HW * C;
  {
    crg_prof_scope ps(ctx, st, CRG_K_GN_STATS, 2.0 * rbs * C, 8.0 * rbs * C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, st, stats, (const float*)nullptr, C, C, HW, groups, rbs * C, 0L, eps, mr);
  }
  {
    const int threads = (C >> 3) <= 256 ? 256 : 512;
    int chunks = HW / 64;
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    while ((long)chunks * N < 512 && chunks * 8 <= HW && chunks < 256) chunks *= 2;
    const int rpc = (HW + chunks - 1) / chunks;
    chunks = (HW + rpc - 1) / rpc;
    crg_prof_scope ps(ctx, st, CRG_K_GN_APPLY, 5.0 * elems, elems * 8.0);
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(chunks, N), dim3(threads), 0, st, (const float*)x, (const float*)nullptr, C, C, HW, groups, rpc, -1,
                       mr, (const float*)nullptr, gamma, beta, eps, fuse_silu, (float*)nullptr, (bf16*)y_hi, (bf16*)y_lo);
  }
  CRG_CHECK_LAUNCH(ctx, "groupnorm_pre_split");
  return 0;
}

extern "C" int crg_groupnorm_mx(crg_ctx* ctx, void* stream, const void* x, const float* stats, const float* gamma, const float* beta,
                                void* y16, void* y8, int N, int HW, int C, int groups, float eps, int fuse_silu, int hi_log2, int lo_log2) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, N > 0 && HW > 0 && C > 0 && x && y16 && y8, "groupnorm_mx: empty input");
  CRG_REQUIRE(ctx, C % 64 == 0 && (((uintptr_t)y16 | (uintptr_t)y8 | (uintptr_t)x) & 15) == 0, "groupnorm_mx: C=%d must be a multiple of 64 and the pointers 16-byte aligned", C);
  const float sh = ldexpf(1.f, hi_log2), sl = ldexpf(1.f, lo_log2);
  if (!stats)  // own statistics: the two-pass form of crg_groupnorm_split
    return groupnorm_impl(ctx, stream, x, nullptr, C, gamma, beta, y16, (bf16*)y16, (bf16*)y16, N, HW, C, groups, eps, fuse_silu, CRG_F32,
                          (unsigned char*)y8, sh, sl);
  CRG_REQUIRE(ctx, groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0 && (C >> 3) <= 512, "groupnorm_mx: groups=%d C=%d unsupported", groups, C);
  CRG_REQUIRE(ctx, HW % 32 == 0 && C / groups <= 128 && ((uintptr_t)stats & 15) == 0, "groupnorm_mx: producer statistics need HW %% 32 == 0 and groups of <= 128 channels");
  const long rbs = (long)N * (HW >> 5);
  float* mr = (float*)crg_scratch(ctx, (size_t)N * groups * 2 * sizeof(float));
  if (!mr) return crg_fail(ctx, -12, "groupnorm_mx: out of scratch");
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)N * HW * C;
  {
    crg_prof_scope ps(ctx, st, CRG_K_GN_STATS, 2.0 * rbs * C, 8.0 * rbs * C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, st, stats, (const float*)nullptr, C, C, HW, groups, rbs * C, 0L, eps, mr);
  }
  {
    const int threads = (C >> 3) <= 256 ? 256 : 512;
    int chunks = HW / 64;
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    while ((long)chunks * N < 512 && chunks * 8 <= HW && chunks < 256) chunks *= 2;
    const int rpc = (HW + chunks - 1) / chunks;
    chunks = (HW + rpc - 1) / rpc;
    crg_prof_scope ps(ctx, st, CRG_K_GN_APPLY, 5.0 * elems, elems * 8.0);
    hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(chunks, N), dim3(threads), 0, st, (const float*)x, (const float*)nullptr, C, C, HW, groups, rpc, -1,
                       mr, (const float*)nullptr, gamma, beta, eps, fuse_silu, (float*)nullptr, (bf16*)y16, (bf16*)y16, (unsigned char*)y8, sh, sl);
  }
  CRG_CHECK_LAUNCH(ctx, "groupnorm_mx");
  return 0;
}

extern "C" int crg_groupnorm_split(crg_ctx* ctx, void* stream, const void* x, const void* x2, int C1, const float* gamma,
                                   const float* beta, void* y_hi, void* y_lo, int N, int HW, int C, int groups, float eps,
                                   int fuse_silu) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, y_hi && y_lo && (((uintptr_t)y_hi | (uintptr_t)y_lo) & 15) == 0, "groupnorm_split: output planes must be 16-byte aligned");
  return groupnorm_impl(ctx, stream, x, x2, C1, gamma, beta, y_hi /* unused as y */, (bf16*)y_hi, (bf16*)y_lo, N, HW, C, groups, eps,
                        fuse_silu, CRG_F32);
}

extern "C" int crg_layernorm(crg_ctx* ctx, void* stream, const void* x, const float* gamma, const float* beta, void* y,
                             int64_t rows, int dim, float eps, int dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, rows > 0 && dim > 0, "layernorm: empty input");
  CRG_REQUIRE(ctx, dim % 8 == 0 && dim <= 64 * 8 * LN_MAXC, "layernorm: dim=%d must be a multiple of 8 and <= %d", dim, 64 * 8 * LN_MAXC);
  CRG_REQUIRE(ctx, ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0, "layernorm: gamma/beta must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)rows * dim;
  crg_prof_scope ps(ctx, st, CRG_K_LAYERNORM, 8.0 * elems, elems * crg_dtype_size(dtype) * 2);
  if (dtype == CRG_BF16)
    launch_layernorm<bf16>(st, (const bf16*)x, gamma, beta, (bf16*)y, (long)rows, dim, eps);
  else if (dtype == CRG_F32)
    launch_layernorm<float>(st, (const float*)x, gamma, beta, (float*)y, (long)rows, dim, eps);
  else
    return crg_fail(ctx, -22, "layernorm: dtype %d unsupported", dtype);
  CRG_CHECK_LAUNCH(ctx, "layernorm");
  return 0;
}

extern "C" int crg_softmax_rows(crg_ctx* ctx, void* stream, const void* x, void* y, int64_t rows, int cols, int64_t ld,
                                float scale, int dtype) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, rows > 0 && cols > 0 && ld >= cols, "softmax_rows: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const double elems = (double)rows * cols;
  crg_prof_scope ps(ctx, st, CRG_K_SOFTMAX, 5.0 * elems, elems * crg_dtype_size(dtype) * 2);
  if (dtype == CRG_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16>, dim3((unsigned)rows), dim3(256), 0, st, (const bf16*)x, (bf16*)y, cols, (long)ld, scale);
  else if (dtype == CRG_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)x, (float*)y, cols, (long)ld, scale);
  else
    return crg_fail(ctx, -22, "softmax_rows: dtype %d unsupported", dtype);
  CRG_CHECK_LAUNCH(ctx, "softmax_rows");
  return 0;
}
