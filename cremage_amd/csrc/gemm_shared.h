// Shared pieces of the MFMA GEMM / implicit-GEMM conv kernels (gemm_conv.hip, conv_pp.hip, gemm_ring.hip): problem descriptor,
// block -> tile mapping, LDS swizzle, epilogues, counted vmcnt wait.
#pragma once
#include "crg_common.h"
#include <type_traits>

namespace crg_mm {

struct GemmP {
  const void* a; const void* a_lo; long lda, a_bs;
  const bf16* w; const bf16* w_lo; long ldw, w_bs;
  const float* bias; int bias_mode;
  const void* res; long ldr, r_bs;
  void* y; long ldy, y_bs;
  int M, N, K, epi;
  const float* cvec; int cvec_rows; long cvec_ld;  // cvec[(m / cvec_rows) * cvec_ld + n]
  const bf16* zero_page;    // >= 16 bytes of zeros in device memory (LDS-DMA source for padding / tails)
  int splits; float* slab;  // split-K: fp32 partial tiles [batch][splits][M][N], reduced by splitk_reduce_kernel
  int a_is_weight;
  // conv geometry
  const void* x2; int C1, C2, Ctot, H, W, Ho, Wo, ks, stride, pad_t, pad_l, up;
  int cm;  // conv K order: 0 = tap-major [tap][Cin]; 1 = chunk-major [Cin/64][tap][64] (consecutive k-tiles re-read the
           // same 64-channel slab of neighbouring pixels -> the 9 taps hit in L1 instead of going back to L2)
  int tiles_n, tiles_m;
  // partial launches (tail of a grid that does not fill whole rounds, see launch()): tiles [tile_base, tile_base + tile_count)
  // of the n-fastest tile order; the split-K slab then only holds rows >= slab_row0
  int tile_base, tile_count, slab_row0;
  int pair;        // paired-column output mapping (see unpair_col): 16-byte epilogue accesses
  int rowhalo;     // conv: eligible for conv3_rowhalo_kernel (3x3, stride 1, pad 1, chunk-major K); 2 = on 256-row tiles
  int halo_lin;    // ... with the linear-pixel row buffer (widths that neither divide the tile nor are a multiple of it)
  int ks_q, ks_r;  // k-tiles per K-slice: nk_total = splits * ks_q + ks_r
  int ring;             // conv: runs on conv3_pp_kernel (conv_pp.hip: 256-pixel tiles, staggered waves)
  unsigned a_bytes, x2_bytes, w_bytes;  // conv: byte sizes of x, x2 and the packed weight (buffer descriptors of conv_pp.hip); 0 = unknown / >= 2 GiB
  int xg_m, xg_n, xg_s;  // XCD partition of the (m-tile, n-tile, k-slice) grid, product 8; xg_s == 0: legacy contiguous order
  // GroupNorm statistics side channel (crg_*_args.gn_stats): per 32-row block and output channel, the sum and the sum of squares of
  // the FINISHED (bf16-rounded) outputs: gstat[0][rb][n] / gstat[1][rb][n], plane stride gstat_plane floats.  Written by the paired
  // epilogue or, for split-K launches, by the reduce kernel; null = none.
  float* gstat; long gstat_plane;
  // rows per statistics partial: 32 (default), or the tile height where the kernel folds its waves' sums through LDS and writes ONE partial per
  // tile and channel (conv3_pp_kernel: 256; set by the launch path when the caller takes tile partials, crg_conv_args.gn_stats_rows)
  int gstat_rows;
  int gstat_tile_ok;  // the caller reads gstat_rows back (crg_conv_args.gn_stats_rows != NULL): tile partials allowed
  // transposed column range (crg_gemm_args.vt): output columns n >= vt_n0 (a multiple of the tile width) go to
  // vt[(m / vt_T) * (N - vt_n0) + n - vt_n0][m % vt_T] (row length vt_ld) instead of y: the V^T operand of crg_attention out of the
  // same launch as Q | K (gemm_glds_kernel, paired epilogue; null = none)
  bf16* vt; int vt_n0, vt_T; long vt_ld;
  // LayerNorm row statistics, PRODUCER side (crg_gemm_args.row_stats): rstat[m][q][0] / rstat[m][q][1] = sum / sum of squares of the
  // finished (rounded) outputs of row m over the columns of partial q - q = 2 * n-tile + wave column of the paired epilogue, i.e.
  // rstat_parts = 2 * tiles_n partials per row (the split-K reduce writes whole-row sums into partial 0 and zeros into the rest).
  // Row-major per row so that the consumer fetches a row's partials with a few 16-byte loads issued together (one latency)
  float* rstat; int rstat_parts;
  // LayerNorm as an epilogue correction, CONSUMER side (crg_gemm_args.ln_stats): the GEMM runs on the RAW rows x with the weight
  // W o gamma, and  y = rstd_m * (acc - mean_m * ln_s[n]) + bias'[n]  with (mean, rstd) folded from the producer's ln_parts partials per
  // row (ln_stat[m][q][2]), ln_s[n] = sum_k (W o gamma)[n][k] over the ROUNDED weight, bias' = W beta + bias (in p.bias)
  const float* ln_stat; int ln_parts; const float* ln_s; float ln_eps;
  // MX form of the fp32-class conv (crg_conv_args.prec = CRG_PREC_F16MX, conv3_rowhalo_kernel<.., MX>): plane 0 fp16, plane 1 the e4m3 pair
  // (a_lo / w_lo point at it); mx_scale[k-group][0 = W operand, 1 = X operand] = E8M0 byte of that group's cross-term operands
  int mx; int mx_scale[2][2];
  // GroupNorm(+SiLU) of the finished output (crg_conv_args.gn_y): when the launch is split along K and a (sample, group) slab fits one
  // block, the kernel that sums the K slices normalises as well (splitk_reduce_gn_kernel) - no reduce launch and no second read of y
  const float* gn_gamma; const float* gn_beta; float gn_eps; int gn_groups, gn_silu, gn_hw; void* gn_y;
};

constexpr int BM = 128;
constexpr int BK = 64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Block -> (m-tile, n-tile, k-slice).  The dispatcher deals workgroups round-robin to the 8 XCDs (block L runs on XCD
// L % 8) and every XCD has its own 4 MB L2, so whatever two XCDs both touch is fetched twice over the fabric.  The host
// picks a partition (xg_m x xg_n x xg_s = 8) of the tile grid that minimises xg_n * |A| + xg_m * |W| (k-slices share
// nothing, so cutting along split-K is free): activation-heavy shapes (64x64 levels) give every XCD a contiguous run of
// m-tiles and the whole small W, weight-heavy shapes (8x8 / 16x16 levels, 1280-wide) give every XCD its own slice of W.
// Inside an XCD the order is n fastest, then m, then k-slice, so co-resident blocks share A rows and W panels.
__device__ __forceinline__ void block_to_tile(const GemmP& p, int& tile_m, int& tile_n, int& sid) {
  const int L = blockIdx.x;
  if (p.xg_s) {
    const int xcd = L & 7, idx = L >> 3;
    const int xs = xcd % p.xg_s, xr = xcd / p.xg_s;
    const int xn = xr % p.xg_n, xm = xr / p.xg_n;
    const int nnl = p.tiles_n / p.xg_n, nml = p.tiles_m / p.xg_m, nsl = p.splits / p.xg_s;
    const int tn = idx % nnl, r = idx / nnl;
    const int tm = r % nml, ts = r / nml;
    tile_n = xn * nnl + tn;
    tile_m = xm * nml + tm;
    sid = xs * nsl + ts;
  } else {  // tile counts not divisible: contiguous runs of tiles per XCD (bijective for any count)
    const int T = p.tile_count;
    sid = L / T;
    int bid = L - sid * T;
    const int q = T >> 3, r = T & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) + p.tile_base;
    tile_n = bid % p.tiles_n;
    tile_m = bid / p.tiles_n;
  }
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane: pure VALU (quad permutes, then the two row
// mirrors), no LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
#endif
  return v;
}

// ---- shared epilogue: lane holds rows n = ..+fq*4+{0..3}, column m = ..+frow of each 16x16 tile ----
// fp32 outputs with p.gstat set (the fp32-class VAE convs, round 3): per 32-row block and channel the sum and the sum of squares of the
// stored values go to the statistics side channel (planes as in the bf16 path), so that the GroupNorm behind the conv
// (model.py:99-113 -> :116-121 of the next ResnetBlock) does not read the fp32 tensor - up to 1 GB - once more for them.
template <int WNT, typename YT, int WMT = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[WNT][WMT], int m0, int n0, int wm, int wn, int frow, int fq,
                                              int bz, int sid, const bf16x4 (&pre)[WNT][WMT], bool use_pre,
                                              const f32x4 (&bpre)[WNT], bool use_bpre) {
  if (p.splits > 1) {
    const long srows = p.M - p.slab_row0;
    float* S = p.slab + ((long)bz * p.splits + sid) * srows * p.N - (long)p.slab_row0 * p.N;
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const int m = m0 + wm * (16 * WMT) + j * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int n = n0 + wn * (16 * WNT) + i * 16 + fq * 4;
        if (n + 4 <= p.N) {
          *reinterpret_cast<f32x4*>(S + (long)m * p.N + n) = acc[i][j];
        } else {
          for (int e = 0; e < 4 && n + e < p.N; ++e) S[(long)m * p.N + n + e] = acc[i][j][e];
        }
      }
    }
    return;
  }
  YT* Y = reinterpret_cast<YT*>(p.y) + (long)bz * p.y_bs;
  const YT* R = p.res ? reinterpret_cast<const YT*>(p.res) + (long)bz * p.r_bs : nullptr;
  constexpr bool CAN_STATS = sizeof(YT) == 4 && (WMT % 2 == 0);
  const bool do_stats = CAN_STATS && p.gstat != nullptr;
  f32x4 gs1[CAN_STATS ? WNT : 1], gs2[CAN_STATS ? WNT : 1];
  auto flush_stats = [&](int jb) {  // rows of the tile pair (jb, jb + 1): fold the 16 lanes of each DPP row, lane frow == 0 stores
    if constexpr (CAN_STATS) {
      const int row0 = m0 + wm * (16 * WMT) + jb * 16;
      const long rb = row0 >> 5;
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = row16_sum(gs1[i][e]);
          b[e] = row16_sum(gs2[i][e]);
        }
        const int n = n0 + wn * (16 * WNT) + i * 16 + fq * 4;
        if (frow == 0 && n + 4 <= p.N && row0 < p.M) {
          *reinterpret_cast<f32x4*>(p.gstat + rb * p.N + n) = a;
          *reinterpret_cast<f32x4*>(p.gstat + p.gstat_plane + rb * p.N + n) = b;
        }
      }
    }
  };
#pragma unroll
  for (int j = 0; j < WMT; ++j) {
    const int m = m0 + wm * (16 * WMT) + j * 16 + frow;
    if constexpr (CAN_STATS) {
      if (do_stats && (j & 1) == 0) {
        if (j > 0) flush_stats(j - 2);
#pragma unroll
        for (int i = 0; i < WNT; ++i) gs1[i] = gs2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (m >= p.M) continue;
    const float brow = (p.bias_mode == CRG_BIAS_ROW) ? p.bias[m] : 0.f;
    const float* cv = p.cvec ? p.cvec + (long)(m / p.cvec_rows) * p.cvec_ld : nullptr;
    if (p.epi == CRG_EPI_GEGLU) {
      // packed columns: [v 16 | g 16] groups; tiles (2u, 2u+1) of this wave are value / gate
      if constexpr (WNT % 2 == 0) {
#pragma unroll
        for (int u = 0; u < WNT / 2; ++u) {
          const int pn = n0 + wn * (16 * WNT) + u * 32 + fq * 4;  // packed column of the value tile
          if (pn >= p.N) continue;
          const int jn = (n0 + wn * (16 * WNT)) / 2 + u * 16 + fq * 4;  // output column
          f32x4 v = acc[2 * u][j], g = acc[2 * u + 1][j];
          if (p.bias_mode == CRG_BIAS_COL) {
            v += *reinterpret_cast<const f32x4*>(p.bias + pn);
            g += *reinterpret_cast<const f32x4*>(p.bias + pn + 16);
          }
          YT out[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) out[e] = (YT)(v[e] * crg_gelu_erf_f(g[e]));
          YT* dst = Y + (long)m * p.ldy + jn;
          if constexpr (sizeof(YT) == 2) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(out);
          else *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<f32x4*>(out);
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < WNT; ++i) {
      const int n = n0 + wn * (16 * WNT) + i * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      const bool full = (n + 4 <= p.N);
      if (full) {
        if (use_bpre) v += bpre[i];
        else if (p.bias_mode == CRG_BIAS_COL) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        else if (p.bias_mode == CRG_BIAS_ROW) v += brow;
        if (p.epi == CRG_EPI_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = crg_silu_f(v[e]);
        }
        if (cv) v += *reinterpret_cast<const f32x4*>(cv + n);
        const long yo = (long)m * p.ldy + n;
        if (use_pre) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)pre[i][j][e];
        } else if (R) {
          const YT* rp = R + (long)m * p.ldr + n;
          if (((p.ldr | n) & 3) == 0) {
            if constexpr (sizeof(YT) == 2) {
              bf16x4 r4 = *reinterpret_cast<const bf16x4*>(rp);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
            } else {
              v += *reinterpret_cast<const f32x4*>(rp);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rp[e];
          }
        }
        if constexpr (CAN_STATS) {
          if (do_stats) {
            gs1[i] += v;
            gs2[i] += v * v;
          }
        }
        if (((p.ldy | n) & 3) == 0) {
          if constexpr (sizeof(YT) == 2) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (bf16)v[e];
            *reinterpret_cast<bf16x4*>(Y + yo) = o4;
          } else {
            *reinterpret_cast<f32x4*>(Y + yo) = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) Y[yo + e] = (YT)v[e];
        }
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) {
          float s = v[e];
          if (p.bias_mode == CRG_BIAS_COL) s += p.bias[n + e];
          else if (p.bias_mode == CRG_BIAS_ROW) s += brow;
          if (p.epi == CRG_EPI_SILU) s = crg_silu_f(s);
          if (cv) s += cv[n + e];
          if (R) s += (float)R[(long)m * p.ldr + n + e];
          Y[(long)m * p.ldy + n + e] = (YT)s;
        }
      }
    }
  }
  if constexpr (CAN_STATS) {
    if (do_stats) flush_stats(WMT - 2);
  }
}

// ---- paired output columns -------------------------------------------------------------------------------------------
// The weight tile is the MFMA A operand, so WHICH output column sits on MFMA row r of tile i is decided by the order of the
// weight rows in LDS.  With the natural order a lane owns columns 16 i + 4 fq .. +3 of every tile: 8-byte accesses, 16
// different cache lines per wave-instruction - and the loads of the residual / the stores of the result, not the MFMAs,
// were what a short-K block spent its time on (in-kernel stamps: 2-5 us of a 13-16 us block).  Storing the rows of each
// PAIR of tiles (2u, 2u+1) as  row(t, rho) <- column 32 u + 8 (rho >> 2) + 4 t + (rho & 3)  makes the lane's two 4-column
// groups adjacent (columns 32 u + 8 fq .. +7): one 16-byte access where there were two 8-byte ones.  The permutation is
// applied once, to the per-lane source row of the LDS-DMA; the fragment reads and the MFMAs are unchanged.
template <int WNT>
__device__ __forceinline__ int unpair_col(int pos) {  // LDS row position within the weight tile -> output column within the tile
  const int w = pos / (16 * WNT);                       // which wave column (wn) the row belongs to
  const int l = pos - w * (16 * WNT);                   // position inside that wave's 16*WNT rows
  if (l >= 32 * (WNT / 2)) return pos;                  // unpaired last tile (odd WNT)
  const int g = l >> 5, t = (l >> 4) & 1, rho = l & 15;
  return w * (16 * WNT) + 32 * g + 8 * (rho >> 2) + 4 * t + (rho & 3);
}

// r2[u][j] / r1[j] / bpre[i]: residual (16 bytes per tile pair, 8 for an odd last tile) and bias of this lane in the paired
// mapping, fetched ahead of the K loop (has_res / has_bias say whether they were).

// SM: 0 = this instantiation never emits GroupNorm statistics, 1 = always (p.gstat is set), 2 = decided at run time.  The statistics
// path keeps ~40 more values live; compiled into the two-blocks-per-CU GEMM kernel as a runtime branch it pushed that kernel from
// 160 to 214 VGPRs (+ 80 accumulators: past 256, i.e. ONE block per CU for every GEMM, statistics or not) - that kernel therefore
// takes it as a template parameter; the one-block-per-CU conv kernels, whose register budget is fixed by their launch bounds, keep
// the runtime form (no extra instantiations, no spills).
// RS: this instantiation also emits the LayerNorm row statistics of its outputs (p.rstat, see GemmP): per row the sum and the sum of
// squares of the ROUNDED values this wave stores (16 WNT columns), folded over the four 16-lane groups that share a row - two
// ds_bpermute per value - and stored by the lanes of group 0: 64 contiguous bytes per plane and 16-row tile.
// tile_lds (block-uniform; needs p.gstat): TILE statistics - the wave sums over ALL its rows and leaves its per-column sums in LDS
// (tile_lds[wave slot][2][16 WNT], wave slot = wm * 2 + wn) behind a block barrier (every wave is past its K loop: the LDS is free);
// the calling kernel folds the row waves and writes one partial per tile and column (p.gstat_rows = tile height).
template <int WNT, int WMT, int SM = 2, bool RS = false>
__device__ __forceinline__ void gemm_epilogue_pairs(const GemmP& p, f32x4 (&acc)[WNT][WMT], int m0, int n0, int wm, int wn, int frow,
                                                    int fq, int bz, const bf16x8 (&r2)[WNT / 2 > 0 ? WNT / 2 : 1][WMT],
                                                    const bf16x4 (&r1)[WMT], bool has_res, const f32x4 (&bpre)[WNT], bool has_bias,
                                                    float* tile_lds = nullptr) {
  bf16* Y = reinterpret_cast<bf16*>(p.y) + (long)bz * p.y_bs;
  const int nb = n0 + wn * (16 * WNT);
  float* RS1 = RS ? p.rstat + ((n0 / (32 * WNT)) * 2 + wn) * 2 : nullptr;
  auto body = [&](auto STATSc) {
    constexpr bool STATS = decltype(STATSc)::value;
    constexpr int NG = WNT / 2 > 0 ? WNT / 2 : 1;
    // STATS: per-channel sum / sum of squares of the rounded outputs over the wave's current 32-row block (two 16-row MFMA tiles)
    float s1[STATS ? NG : 1][8], s2[STATS ? NG : 1][8], t1[4], t2[4];
    const bool all_valid = m0 + wm * (16 * WMT) + 16 * WMT <= p.M;  // wave-uniform: the usual case, no per-row select
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const int m = m0 + wm * (16 * WMT) + j * 16 + frow;
      const bool valid = m < p.M;
      if constexpr (!STATS) {
        if (!valid) continue;
      } else if ((j & 1) == 0 && (tile_lds == nullptr || j == 0)) {
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) s1[u][e] = s2[u][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) t1[e] = t2[e] = 0.f;
      }
      const float brow = (p.bias_mode == CRG_BIAS_ROW) ? p.bias[valid ? m : 0] : 0.f;
      const float* cv = p.cvec ? p.cvec + (long)((valid ? m : 0) / p.cvec_rows) * p.cvec_ld : nullptr;
      float q1 = 0.f, q2 = 0.f;  // RS: this lane's share of row m
      auto finish = [&](f32x4 v, int i, int n) {
        if (has_bias) v += bpre[i];
        else if (p.bias_mode == CRG_BIAS_ROW) v += brow;
        if (p.epi == CRG_EPI_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = crg_silu_f(v[e]);
        }
        if (cv) v += *reinterpret_cast<const f32x4*>(cv + n);
        return v;
      };
#pragma unroll
      for (int u = 0; u < WNT / 2; ++u) {
        const int n = nb + 32 * u + 8 * fq;
        if (n >= p.N) continue;  // N % 8 == 0 in this mode: a group is in or out as a whole (wave-uniform per 16-lane row)
        f32x4 a = finish(acc[2 * u][j], 2 * u, n), b = finish(acc[2 * u + 1][j], 2 * u + 1, n + 4);
        if (has_res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[e] += (float)r2[u][j][e];
            b[e] += (float)r2[u][j][4 + e];
          }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)a[e];
          o[4 + e] = (bf16)b[e];
        }
        if (valid) *reinterpret_cast<bf16x8*>(Y + (long)m * p.ldy + n) = o;
        if constexpr (RS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)o[e];
            q1 += f;
            q2 = __builtin_fmaf(f, f, q2);
          }
        }
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (all_valid || valid) ? (float)o[e] : 0.f;
            s1[u][e] += f;
            s2[u][e] = __builtin_fmaf(f, f, s2[u][e]);
          }
        }
      }
      if constexpr (WNT & 1) {
        const int n = nb + 16 * (WNT - 1) + 4 * fq;
        if (n < p.N) {
          f32x4 a = finish(acc[WNT - 1][j], WNT - 1, n);
          if (has_res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += (float)r1[j][e];
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)a[e];
          if (valid) *reinterpret_cast<bf16x4*>(Y + (long)m * p.ldy + n) = o;
          if constexpr (RS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float f = (float)o[e];
              q1 += f;
              q2 = __builtin_fmaf(f, f, q2);
            }
          }
          if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float f = (all_valid || valid) ? (float)o[e] : 0.f;
              t1[e] += f;
              t2[e] = __builtin_fmaf(f, f, t2[e]);
            }
          }
        }
      }
      if constexpr (RS) {
        q1 += __shfl_xor(q1, 16);
        q2 += __shfl_xor(q2, 16);
        q1 += __shfl_xor(q1, 32);
        q2 += __shfl_xor(q2, 32);
        if (fq == 0 && valid) *reinterpret_cast<f32x2*>(RS1 + (long)m * p.rstat_parts * 2) = f32x2{q1, q2};
      }
      if constexpr (STATS) {
        if (tile_lds != nullptr) {
          if (j == WMT - 1) {
            float* L1 = tile_lds + (wm * 2 + wn) * (32 * WNT);
            float* L2 = L1 + 16 * WNT;
            __syncthreads();  // every wave of the block is past its K loop: the LDS is free
#pragma unroll
            for (int u = 0; u < WNT / 2; ++u) {
              f32x4 a1, b1, a2, b2;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                a1[e] = row16_sum(s1[u][e]);
                b1[e] = row16_sum(s1[u][4 + e]);
                a2[e] = row16_sum(s2[u][e]);
                b2[e] = row16_sum(s2[u][4 + e]);
              }
              if (frow == 0) {
                *reinterpret_cast<f32x4*>(L1 + 32 * u + 8 * fq) = a1;
                *reinterpret_cast<f32x4*>(L1 + 32 * u + 8 * fq + 4) = b1;
                *reinterpret_cast<f32x4*>(L2 + 32 * u + 8 * fq) = a2;
                *reinterpret_cast<f32x4*>(L2 + 32 * u + 8 * fq + 4) = b2;
              }
            }
            if constexpr (WNT & 1) {
              f32x4 c1, c2;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                c1[e] = row16_sum(t1[e]);
                c2[e] = row16_sum(t2[e]);
              }
              if (frow == 0) {
                *reinterpret_cast<f32x4*>(L1 + 16 * (WNT - 1) + 4 * fq) = c1;
                *reinterpret_cast<f32x4*>(L2 + 16 * (WNT - 1) + 4 * fq) = c2;
              }
            }
          }
        } else if ((j & 1) == 1) {
          // fold the 16 rows held by the lanes of each DPP row (one lane group = one fq = one 8-channel column group), then lane
          // frow == 0 of every group stores its channels' sums for row block rb
          const int row0 = m0 + wm * (16 * WMT) + (j >> 1) * 32;
          const long rb = row0 >> 5;
          float* S1 = p.gstat + rb * p.N;
          float* S2 = S1 + p.gstat_plane;
#pragma unroll
          for (int u = 0; u < WNT / 2; ++u) {
            f32x4 a1, b1, a2, b2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a1[e] = row16_sum(s1[u][e]);
              b1[e] = row16_sum(s1[u][4 + e]);
              a2[e] = row16_sum(s2[u][e]);
              b2[e] = row16_sum(s2[u][4 + e]);
            }
            const int n = nb + 32 * u + 8 * fq;
            if (frow == 0 && n < p.N && row0 < p.M) {
              *reinterpret_cast<f32x4*>(S1 + n) = a1;
              *reinterpret_cast<f32x4*>(S1 + n + 4) = b1;
              *reinterpret_cast<f32x4*>(S2 + n) = a2;
              *reinterpret_cast<f32x4*>(S2 + n + 4) = b2;
            }
          }
          if constexpr (WNT & 1) {
            f32x4 a1, a2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a1[e] = row16_sum(t1[e]);
              a2[e] = row16_sum(t2[e]);
            }
            const int n = nb + 16 * (WNT - 1) + 4 * fq;
            if (frow == 0 && n < p.N && row0 < p.M) {
              *reinterpret_cast<f32x4*>(S1 + n) = a1;
              *reinterpret_cast<f32x4*>(S2 + n) = a2;
            }
          }
        }
      }
    }
  };
  if constexpr (SM == 1) body(std::integral_constant<bool, true>{});
  else if constexpr (SM == 0) body(std::integral_constant<bool, false>{});
  else if (p.gstat) body(std::integral_constant<bool, true>{});
  else body(std::integral_constant<bool, false>{});
}

// (sum, sum of squares) of one row from its `parts` partials (stat[m][q][2], parts even, <= 16): all the 16-byte loads go out together,
// clamped duplicates where there are fewer partials (one load latency, not one per partial)
__device__ __forceinline__ f32x2 ln_fold_row(const float* stat, long m, int parts) {
  const f32x4* q = reinterpret_cast<const f32x4*>(stat + m * parts * 2);
  const int nq = parts >> 1;
  float a = 0.f, b = 0.f;
  if (nq <= 4) {
    f32x4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = q[u < nq ? u : 0];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float k = u < nq ? 1.f : 0.f;
      a += k * (t[u][0] + t[u][2]);
      b += k * (t[u][1] + t[u][3]);
    }
  } else {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = q[u < nq ? u : 0];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float k = u < nq ? 1.f : 0.f;
      a += k * (t[u][0] + t[u][2]);
      b += k * (t[u][1] + t[u][3]);
    }
  }
  return f32x2{a, b};
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields left at "no wait")
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// conv_pp.hip: 3x3 / stride 1 conv on 256-pixel tiles, 4-slot weight ring + halo'd row buffers, the two waves of a SIMD half a k-tile
// apart (bf16), launched in place of conv3_rowhalo_kernel<.., MT = 2>
int launch_conv_pp(crg_ctx* ctx, hipStream_t st, const GemmP& p, int wnt);
// gemm_ring.hip: persistent 256-row-tile GEMM (bf16, plain / GEGLU epilogue), launched in place of gemm_glds_kernel where it applies
bool ring_gemm_ok(const GemmP& p, int batch, int n_cu);
int launch_gemm_ring(crg_ctx* ctx, hipStream_t st, const GemmP& p, double flops, double bytes);

}  // namespace crg_mm
