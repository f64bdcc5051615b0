// MFMA GEMM / implicit-GEMM conv2d for gfx950.
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )      (both operands K-contiguous)
//
// Block tile 128(m) x 32*WNT(n) x 64(k), 256 threads = 4 waves as 2(m) x 2(n); each wave owns
// 64(m) x 16*WNT(n) as 4 x WNT accumulators of v_mfma_f32_16x16x32_bf16.  The MFMA is issued with
// the WEIGHT tile as the A operand and the ACTIVATION tile as the B operand, so that D rows = n:
// a lane then holds 4 CONSECUTIVE output channels of one output row, and the epilogue (bias,
// timestep-embedding vector, residual, SiLU/GEGLU) runs on 4-wide vectors with 8/16-byte stores.
//
// Staging: global -> registers -> LDS (ds_write_b128), double-buffered LDS, one barrier per k-tile;
// the loads of k-tile t+1 are issued before the MFMAs of tile t and written after them (guide
// §6 G15 "async-STAGE split").  LDS rows are 128 B with the 16-B chunk index XOR-swizzled by
// (row & 7) so that the ds_read_b128 fragment reads (16 rows x same chunk) are conflict-free.
//
// The conv variant only changes the activation loader: row m -> (image, ho, wo), k -> (tap, channel)
// with zero fill outside the (optionally nearest-2x-upsampled) image and a second input pointer
// for the UNet skip concat.  BF16X3 (NSPLIT=2) stages hi and lo bf16 planes of both operands and
// issues hi*hi + hi*lo + lo*hi.
#include "gemm_shared.h"
#include <cstdlib>

namespace {
using namespace crg_mm;



__device__ __forceinline__ bf16x8 split_hi(const crg_vec8<float>& v, bf16x8& lo) {
  bf16x8 hi;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float f = v.get(i);
    bf16 h = (bf16)f;
    hi[i] = h;
    lo[i] = (bf16)(f - (float)h);
  }
  return hi;
}


// KG = 2 (the split-bf16 path): 8 waves; the second group of four multiplies the second 32-wide k-step of every k-tile and
// every thread stages half as many rows.  That path fills the LDS with one block per CU, and with one wave per SIMD the
// staging VALU work (fp32 -> hi/lo split, conv addressing: as many issue cycles as the MFMAs) ran strictly after the MFMAs;
// two waves per SIMD overlap one wave's staging with the other's MFMAs.
template <int WNT, int NSPLIT, typename AT, typename YT, bool CONV, int KG>
__global__ __launch_bounds__(256 * KG) void gemm_kernel(GemmP p) {
  constexpr int BN = 32 * WNT;
  constexpr int RS = 32 * KG;              // rows staged per pass by the block
  constexpr int XL = BM / RS;              // A rows per thread
  constexpr int WL = (BN + RS - 1) / RS;   // W rows per thread (last one guarded when RS does not divide BN)
  constexpr int XS_BYTES = BM * 128;
  constexpr int WS_BYTES = BN * 128;
  constexpr int STAGE_BYTES = NSPLIT * (XS_BYTES + WS_BYTES);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int kg = wave >> 2;
  const int wm = (wave & 3) >> 1, wn = wave & 1;

  int tile_m, tile_n, sid;
  block_to_tile(p, tile_m, tile_n, sid);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int bz = blockIdx.y;

  const AT* A = reinterpret_cast<const AT*>(p.a) + (long)bz * p.a_bs;
  const bf16* A_lo = p.a_lo ? reinterpret_cast<const bf16*>(p.a_lo) + (long)bz * p.a_bs : nullptr;
  const bf16* Wp = p.w + (long)bz * p.w_bs;
  const bf16* Wlo = (NSPLIT == 2) ? p.w_lo + (long)bz * p.w_bs : nullptr;

  const int cc = t & 7;   // 16-byte chunk column inside the 64-wide k-tile
  const int r0 = t >> 3;  // 0..RS-1

  // ---- per-thread row descriptors (fixed for the whole K loop) ----
  long xrow_off[XL];   // linear: element offset of row start; conv: image index
  int xh[XL], xw[XL];  // conv: top-left input coordinate (virtual = after upsample)
  bool xok[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = m0 + r0 + RS * i;
    xok[i] = m < p.M;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      xrow_off[i] = img;
      xh[i] = ho * p.stride - p.pad_t;
      xw[i] = wo * p.stride - p.pad_l;
    } else {
      xrow_off[i] = (long)m * p.lda;
      xh[i] = xw[i] = 0;
    }
  }
  bool wok[WL];
  long wrow_off[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int n = n0 + r0 + RS * i;
    wok[i] = n < p.N && r0 + RS * i < BN;
    wrow_off[i] = (long)n * p.ldw;
  }

  crg_vec8<AT> xr[XL];
  bf16x8 xr_lo[XL];  // only when A is a pre-split weight (a_is_weight)
  bf16x8 wr[WL], wr_lo[WL];
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  auto load_tile = [&](int kt) {
    const int kc = kt * BK + cc * 8;
    const bool kok = kc < p.K;
    if (CONV) {
      int tap, c;
      if (p.cm) {
        const int taps = p.ks * p.ks;
        tap = kt % taps;
        c = (kt / taps) * BK + cc * 8;
      } else {
        tap = kc / p.Ctot;
        c = kc - tap * p.Ctot;
      }
      const int kh = tap / p.ks;
      const int kw = tap - kh * p.ks;
      const int Hv = p.up ? 2 * p.H : p.H, Wv = p.up ? 2 * p.W : p.W;
      const bool second = c >= p.C1;
      const AT* base = reinterpret_cast<const AT*>(second ? p.x2 : p.a);
      const int Cs = second ? p.C2 : p.C1;
      const int cs = second ? c - p.C1 : c;
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int hv = xh[i] + kh, wv = xw[i] + kw;
        const bool ok = kok && xok[i] && (unsigned)hv < (unsigned)Hv && (unsigned)wv < (unsigned)Wv;
        if (ok) {
          const int hs = p.up ? hv >> 1 : hv, ws = p.up ? wv >> 1 : wv;
          xr[i].load(base + ((xrow_off[i] * p.H + hs) * p.W + ws) * Cs + cs);
        } else {
          xr[i].zero();
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        if (kok && xok[i]) {
          xr[i].load(A + xrow_off[i] + kc);
          if (NSPLIT == 2 && sizeof(AT) == 2) xr_lo[i] = *reinterpret_cast<const bf16x8*>(A_lo + xrow_off[i] + kc);
        } else {
          xr[i].zero();
          if (NSPLIT == 2 && sizeof(AT) == 2) xr_lo[i] = zero8;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      if (kok && wok[i]) {
        wr[i] = *reinterpret_cast<const bf16x8*>(Wp + wrow_off[i] + kc);
        if (NSPLIT == 2) wr_lo[i] = *reinterpret_cast<const bf16x8*>(Wlo + wrow_off[i] + kc);
      } else {
        wr[i] = zero8;
        if (NSPLIT == 2) wr_lo[i] = zero8;
      }
    }
  };

  auto store_tile = [&](int stage) {
    char* xs = smem + stage * STAGE_BYTES;
    char* ws = xs + NSPLIT * XS_BYTES;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int row = r0 + RS * i;
      const int off = lds_off(row, cc);
      if constexpr (sizeof(AT) == 4) {
        bf16x8 lo;
        const crg_vec8<float>& xv = reinterpret_cast<const crg_vec8<float>&>(xr[i]);
        bf16x8 hi = split_hi(xv, lo);
        *reinterpret_cast<bf16x8*>(xs + off) = hi;
        if (NSPLIT == 2) *reinterpret_cast<bf16x8*>(xs + XS_BYTES + off) = lo;
      } else {
        *reinterpret_cast<bf16x8*>(xs + off) = reinterpret_cast<const crg_vec8<bf16>&>(xr[i]).v;
        if (NSPLIT == 2) *reinterpret_cast<bf16x8*>(xs + XS_BYTES + off) = xr_lo[i];
      }
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int row = r0 + RS * i;
      if (RS * (WL - 1) + RS > BN && row >= BN) continue;  // guarded last pass
      const int off = lds_off(row, cc);
      *reinterpret_cast<bf16x8*>(ws + off) = wr[i];
      if (NSPLIT == 2) *reinterpret_cast<bf16x8*>(ws + WS_BYTES + off) = wr_lo[i];
    }
  };

  f32x4 acc[WNT][4];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // k-tile range of this K-slice: the first ks_r slices take ks_q + 1 tiles, the others ks_q (ks_q, ks_r from the host - a
  // 64-bit division by the runtime split count here cost ~1 us of scalar work on every block's latency chain)
  const int kt_begin = sid * p.ks_q + (sid < p.ks_r ? sid : p.ks_r);
  const int nk = kt_begin + p.ks_q + (sid < p.ks_r ? 1 : 0);
  load_tile(kt_begin);
  store_tile(kt_begin & 1);
  __syncthreads();

  const int frow = lane & 15;
  const int fq = lane >> 4;
  for (int kt = kt_begin; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tile(kt + 1);
    const char* xs = smem + (kt & 1) * STAGE_BYTES;
    const char* ws = xs + NSPLIT * XS_BYTES;
#pragma unroll
    for (int k2 = 0; k2 < 2 / KG; ++k2) {
      const int ks = KG == 2 ? kg : k2;
      bf16x8 xf[4], xl[4], wf[WNT], wl[WNT];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int off = lds_off(wm * 64 + j * 16 + frow, ks * 4 + fq);
        xf[j] = *reinterpret_cast<const bf16x8*>(xs + off);
        if (NSPLIT == 2) xl[j] = *reinterpret_cast<const bf16x8*>(xs + XS_BYTES + off);
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int off = lds_off(wn * (16 * WNT) + i * 16 + frow, ks * 4 + fq);
        wf[i] = *reinterpret_cast<const bf16x8*>(ws + off);
        if (NSPLIT == 2) wl[i] = *reinterpret_cast<const bf16x8*>(ws + WS_BYTES + off);
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (NSPLIT == 2) {
            acc[i][j] = CRG_MFMA_16x16x32(wl[i], xf[j], acc[i][j]);
            acc[i][j] = CRG_MFMA_16x16x32(wf[i], xl[j], acc[i][j]);
          }
          acc[i][j] = CRG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
        }
    }
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  if constexpr (KG == 2) {
    // fold the second k-group's partial tile into the first through the (now idle) staging buffers
    static_assert(4 * WNT * 4 * 64 * 16 <= 2 * STAGE_BYTES, "reduction buffer must fit in the staging LDS");
    f32x4* red = reinterpret_cast<f32x4*>(smem) + ((wave & 3) * WNT * 4) * 64 + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(i * 4 + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += red[(i * 4 + j) * 64];
  }
  const bf16x4 no_pre[WNT][4] = {};
  const f32x4 no_bias[WNT] = {};
  gemm_epilogue<WNT, YT>(p, acc, m0, n0, wm, wn, frow, fq, bz, sid, no_pre, false, no_bias, false);
}

// ------------------------------------------------------------------------------------------------------
// bf16 fast path: same tile / MFMA / epilogue, but BOTH operands are staged global -> LDS with
// global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip, no ds_write): on gfx950 a ds_write_b128 costs ~13 LDS
// cycles per wave-instruction (MI355X_MICROARCH.md §LDS), which made the register-staged kernel LDS-bound.
// One wave-instruction writes 1 KiB = 8 rows x 128 B, lane-linear; the XOR swizzle of the 16-byte chunk is
// applied on the per-lane SOURCE address (guide rule 21).  Out-of-image taps / tile tails read a 16-byte zero
// page instead, which gives the zero padding for free.
// Pipeline: a ring of STAGES LDS buffers, k-tiles t+1 .. t+STAGES-1 in flight while tile t is multiplied.  Each wave
// waits for ITS OWN loads of tile t with a counted s_waitcnt vmcnt (loads retire in order, so "all but the newest
// (STAGES-2) tiles' worth"), then one s_barrier makes every wave's part of tile t visible and proves that everybody is
// done reading the buffer of tile t-1, which is the one the next DMA batch overwrites.  STAGES = 2 keeps two blocks per
// CU (2 x 74 KB); grids that cannot put two blocks on a CU anyway run STAGES = 4 (147 KB) so that a lone block still
// covers the ~1.5 us global -> LDS latency.

// s_waitcnt vmcnt(n * LOADS) for a wave-uniform number of k-tile batches n = 0..4 that may stay in flight
template <int LOADS>
__device__ __forceinline__ void wait_batches(int n) {
  if (n <= 0) wait_vmcnt<0>();
  else if (n == 1) wait_vmcnt<(1 * LOADS < 64 ? 1 * LOADS : 0)>();
  else if (n == 2) wait_vmcnt<(2 * LOADS < 64 ? 2 * LOADS : 0)>();
  else if (n == 3) wait_vmcnt<(3 * LOADS < 64 ? 3 * LOADS : 0)>();
  else wait_vmcnt<(4 * LOADS < 64 ? 4 * LOADS : 0)>();
}

// (launch bounds: configuration A - 128-row tiles, 4 waves, 2 stages - is built for TWO blocks per CU: its register budget is 256.  The
// plain instantiations fit it on their own (156 VGPRs + 80 accumulators); the one that emits GroupNorm statistics needs the cap.)
// XT: the extra side channel this instantiation carries (each keeps ~10-40 more values live, hence template arms rather than runtime branches):
//   0 none; 1 = GST, emits GroupNorm statistics (p.gstat); 2 = RST, emits LayerNorm row statistics (p.rstat); 3 = LNE, LayerNorm as an
//   epilogue correction from the producer's row statistics (p.ln_stat, see GemmP) - no residual in that form (none of its callers has one)
template <int WNT, typename YT, bool CONV, int STAGES, int WMT, int KG, int NS = 1, bool PAIR = false, int XT = 0>
#ifndef CRG_LB_A  // dev knob (tools/build_variant.sh): 1 (default) = every configuration-A instantiation is capped at 256 registers (unified file: 191-204
           // VGPRs, no accumulator registers; bench A/B in one call: 246.6 -> 245.6 ms), 0 = only the statistics one
#define CRG_LB_A 1
#endif
__global__ __launch_bounds__(256 * KG, ((XT != 0 || CRG_LB_A) && KG == 1 && WMT == 4 && STAGES == 2 && NS == 1) ? 2 : 1) void gemm_glds_kernel(GemmP p) {
  constexpr bool GST = XT == 1, RST = XT == 2, LNE = XT == 3;
  static_assert(!PAIR || (sizeof(YT) == 2 && NS == 1), "paired columns: bf16 output, single-plane operands");
  static_assert(!(GST || RST) || PAIR, "GroupNorm / LayerNorm statistics come from the paired epilogue");  // GST / RST: the instantiations that emit them
  static_assert(!LNE || (!CONV && NS == 1 && sizeof(YT) == 2), "LayerNorm epilogue: bf16 GEMM");
  // NS = 2: split-bf16 (fp32-class) operands - the activations arrive pre-split as two bf16 planes (hi = bf16(x),
  // lo = bf16(x - hi), written by the producing GroupNorm / split pass), the weights as their two packed planes; all four
  // are staged by LDS-DMA and every fragment pair costs three MFMAs (hi*hi + hi*lo + lo*hi).
  // Block tile (32*WMT) x (32*WNT) x 64; 4*KG waves.  The 4 waves of a k-group tile the block 2 x 2 (each 16*WMT rows x
  // 16*WNT columns); with KG == 2 the second group multiplies the second 32-wide k-step of every k-tile (in-block
  // split-K: twice the waves per SIMD on the same LDS traffic - for grids that cannot put two blocks on a CU) and its
  // accumulators are folded into group 0 through LDS before the epilogue.
  constexpr int BMT = 32 * WMT;
  constexpr int BN = 32 * WNT;
  constexpr int NW = 4 * KG;
  constexpr int XS_BYTES = BMT * 128;
  constexpr int WS_BYTES = BN * 128;
  constexpr int STAGE_BYTES = NS * (XS_BYTES + WS_BYTES);  // [X hi][X lo][W hi][W lo]
  constexpr int XRG = BMT / 8, WRG = BN / 8;      // 8-row groups (one 1 KiB wave-instruction each)
  constexpr int XL = (XRG + NW - 1) / NW;         // DMA instructions per wave per k-tile (uniform: waves without a row
  constexpr int WL = (WRG + NW - 1) / NW;         // group left issue a dummy load so that vmcnt counts stay uniform)
  constexpr int DUMMY = STAGES * STAGE_BYTES;     // 1 KiB scratch target of the dummy loads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kg = wave >> 2;  // k-group of this wave
  const int wm = (wave & 3) >> 1, wn = wave & 1;

  int tile_m, tile_n, sid;
  block_to_tile(p, tile_m, tile_n, sid);
  const int m0 = tile_m * BMT, n0 = tile_n * BN;
  const int bz = blockIdx.y;

  const bf16* A = reinterpret_cast<const bf16*>(p.a) + (long)bz * p.a_bs;
  const bf16* Wp = p.w + (long)bz * p.w_bs;
  // lo planes live at a fixed element distance from the hi planes (same layout): one offset each
  const long a_lo_off = NS == 2 ? reinterpret_cast<const bf16*>(p.a_lo) - reinterpret_cast<const bf16*>(p.a) : 0;
  const long w_lo_off = NS == 2 ? p.w_lo - p.w : 0;
  const bf16* zpage = p.zero_page;

  // this lane always stages LDS row (8*g + lane/8), physical chunk lane%8 of row group g = wave + NW*q;
  // the data it carries is logical chunk clog = (lane%8) ^ (row%8)
  const int rsub = lane >> 3;
  const int clog = (lane & 7) ^ rsub;

  long xrow_off[XL];
  int xh[XL], xw[XL];
  bool xok[XL];
#pragma unroll
  for (int q = 0; q < XL; ++q) {
    const int m = m0 + (wave + NW * q) * 8 + rsub;
    xok[q] = m < p.M && (wave + NW * q) < XRG;
    if (CONV) {
      const int hw = p.Ho * p.Wo;
      const int img = m / hw;
      const int rem = m - img * hw;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      xrow_off[q] = (long)img * p.H * p.W;
      xh[q] = ho * p.stride - p.pad_t;
      xw[q] = wo * p.stride - p.pad_l;
    } else {
      xrow_off[q] = (long)m * p.lda;
      xh[q] = xw[q] = 0;
    }
  }
  bool wok[WL];
  long wrow_off[WL];
#pragma unroll
  for (int q = 0; q < WL; ++q) {
    const int pos = (wave + NW * q) * 8 + rsub;                     // LDS row of the weight tile this lane stages
    const int n = n0 + (PAIR ? unpair_col<WNT>(pos) : pos);       // the output column (= weight row) that lives there
    wok[q] = n < p.N && (wave + NW * q) < WRG;
    wrow_off[q] = (long)n * p.ldw;
  }

  // k-tile range of this K-slice: the first ks_r slices take ks_q + 1 tiles, the others ks_q (ks_q, ks_r from the host - a
  // 64-bit division by the runtime split count here cost ~1 us of scalar work on every block's latency chain)
  const int kt_begin = sid * p.ks_q + (sid < p.ks_r ? sid : p.ks_r);
  const int nk = kt_begin + p.ks_q + (sid < p.ks_r ? 1 : 0);

  // running (tap, channel) of this lane's logical chunk; advanced by 64 channels per k-tile
  int kc = kt_begin * BK + clog * 8;
  int tap = 0, cch = kc;
  const int taps = p.ks * p.ks;
  if (CONV) {
    if (p.cm) {
      tap = kt_begin % taps;
      cch = (kt_begin / taps) * BK + clog * 8;
    } else {
      tap = kc / p.Ctot;
      cch = kc - tap * p.Ctot;
    }
  }
  const int Hv = p.up ? 2 * p.H : p.H, Wv = p.up ? 2 * p.W : p.W;

  auto stage = [&](int buf) {
    char* xs = smem + buf * STAGE_BYTES;
    char* ws = xs + NS * XS_BYTES;
    const bool kok = kc < p.K;
    if (CONV) {
      const int kh = (p.ks == 3) ? (tap * 11) >> 5 : 0;
      const int kw = (p.ks == 3) ? tap - kh * 3 : 0;
      const bool second = cch >= p.C1;
      const bf16* base = reinterpret_cast<const bf16*>(second ? p.x2 : p.a);
      const int Cs = second ? p.C2 : p.C1;
      const int cs = second ? cch - p.C1 : cch;
#pragma unroll
      for (int q = 0; q < XL; ++q) {
        const int hv = xh[q] + kh, wv = xw[q] + kw;
        const bool ok = kok && xok[q] && (unsigned)hv < (unsigned)Hv && (unsigned)wv < (unsigned)Wv;
        const int hs = p.up ? hv >> 1 : hv, wsrc = p.up ? wv >> 1 : wv;
        const bf16* src = ok ? base + (xrow_off[q] + (long)hs * p.W + wsrc) * Cs + cs : zpage;
        char* dst = (wave + NW * q) < XRG ? xs + (wave + NW * q) * 1024 : smem + DUMMY;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        if constexpr (NS == 2) {
          const bf16* src2 = ok ? src + a_lo_off : zpage;
          char* dst2 = (wave + NW * q) < XRG ? dst + XS_BYTES : smem + DUMMY;
          __builtin_amdgcn_global_load_lds((gptr_t)src2, (lptr_t)dst2, 16, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < XL; ++q) {
        const bool ok = kok && xok[q];
        const bf16* src = ok ? A + xrow_off[q] + kc : zpage;
        char* dst = (wave + NW * q) < XRG ? xs + (wave + NW * q) * 1024 : smem + DUMMY;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        if constexpr (NS == 2) {
          const bf16* src2 = ok ? src + a_lo_off : zpage;
          char* dst2 = (wave + NW * q) < XRG ? dst + XS_BYTES : smem + DUMMY;
          __builtin_amdgcn_global_load_lds((gptr_t)src2, (lptr_t)dst2, 16, 0, 0);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < WL; ++q) {
      const bool ok = kok && wok[q];
      const bf16* src = ok ? Wp + wrow_off[q] + kc : zpage;
      char* dst = (wave + NW * q) < WRG ? ws + (wave + NW * q) * 1024 : smem + DUMMY;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      if constexpr (NS == 2) {
        const bf16* src2 = ok ? src + w_lo_off : zpage;
        char* dst2 = (wave + NW * q) < WRG ? dst + WS_BYTES : smem + DUMMY;
        __builtin_amdgcn_global_load_lds((gptr_t)src2, (lptr_t)dst2, 16, 0, 0);
      }
    }
    // advance to the next k-tile
    kc += BK;
    if (CONV) {
      if (p.cm) {
        if (++tap == taps) {
          tap = 0;
          cch += BK;
        }
      } else {
        cch += BK;
        while (cch >= p.Ctot) {
          cch -= p.Ctot;
          ++tap;
        }
      }
    }
  };

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fq = lane >> 4;
  // residual tile of this lane, fetched first so that its latency hides under the whole K loop (bf16 outputs); being the
  // oldest loads in flight they are retired by the first counted wait
  // The loads below are UNCONDITIONAL on clamped, always-valid addresses (rows >= M / columns >= N are never used by the
  // epilogue): a `cond ? load : 0` select made the compiler zero-fill, branch and - worse - wait (s_waitcnt vmcnt(0)) right
  // behind every load, which serialised the residual latency in front of the first DMA of every block.
  // PAIR: one 16-byte load per tile pair (r2) plus an 8-byte one for an odd last tile (r1).
  bf16x4 rres[PAIR ? 1 : WNT][PAIR ? 1 : WMT];
  bf16x8 r2[PAIR ? (WNT / 2 > 0 ? WNT / 2 : 1) : 1][PAIR ? WMT : 1];
  bf16x4 r1[PAIR ? WMT : 1];
  f32x4 bpre[WNT];
  const bool pre_res = !LNE && sizeof(YT) == 2 && p.res && p.splits == 1 && (p.ldr & 3) == 0 && (p.N & 3) == 0 && p.epi != CRG_EPI_GEGLU;
  const bool pre_bias = p.bias_mode == CRG_BIAS_COL && p.splits == 1 && (p.N & 3) == 0 && p.epi != CRG_EPI_GEGLU;
  if (kg == 0) {
    const int nb = n0 + wn * (16 * WNT);
    if (pre_res) {
      const bf16* R = reinterpret_cast<const bf16*>(p.res) + (long)bz * p.r_bs;
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const int m = m0 + wm * (16 * WMT) + j * 16 + frow;
        const bf16* Rm = R + (long)(m < p.M ? m : p.M - 1) * p.ldr;
        if constexpr (PAIR) {
#pragma unroll
          for (int u = 0; u < WNT / 2; ++u) {
            const int n = nb + 32 * u + 8 * fq;
            r2[u][j] = *reinterpret_cast<const bf16x8*>(Rm + (n + 8 <= p.N ? n : p.N - 8));
          }
          if constexpr (WNT & 1) {
            const int n = nb + 16 * (WNT - 1) + 4 * fq;
            r1[j] = *reinterpret_cast<const bf16x4*>(Rm + (n + 4 <= p.N ? n : p.N - 4));
          }
        } else {
#pragma unroll
          for (int i = 0; i < WNT; ++i) {
            const int n = nb + i * 16 + fq * 4;
            rres[i][j] = *reinterpret_cast<const bf16x4*>(Rm + (n + 4 <= p.N ? n : p.N - 4));
          }
        }
      }
    }
    // column bias of this lane's 4-wide output groups: also fetched ahead of the K loop (a lone block per CU would otherwise
    // pay a full global-load latency between its last MFMA and its first store)
    if (pre_bias) {
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int n = (PAIR && i < 2 * (WNT / 2)) ? nb + 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : nb + 16 * i + 4 * fq;
        bpre[i] = *reinterpret_cast<const f32x4*>(p.bias + (n + 4 <= p.N ? n : p.N - 4));
      }
    }
  }

  constexpr int LOADS = NS * (XL + WL);  // LDS-DMA instructions per wave per k-tile
  constexpr int D = STAGES - 1;   // prefetch distance
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (kt_begin + s < nk) stage(s);
  // LNE: (rstd, mean * rstd) of this lane's rows and the column sums ln_s of its columns.  Lane l of the wave folds the partials of the
  // wave's row l (coalesced: consecutive lanes, consecutive rows of a plane); the values of rows 16 j + frow come over by ds_bpermute.
  // Placed BEHIND the first DMA batches: the fold needs the partials, so the wait it implies covers the DMA latency it runs beside
  // (ahead of them it would put a load round trip in front of every block's first DMA).
  float ln_a = 0.f, ln_b = 0.f;
  f32x4 spre[LNE ? WNT : 1];
  if constexpr (LNE) {
    if (kg == 0) {
      const int rl = lane < 16 * WMT ? lane : 16 * WMT - 1;
      const int mr = m0 + wm * (16 * WMT) + rl;
      const f32x2 ab = ln_fold_row(p.ln_stat, mr < p.M ? mr : p.M - 1, p.ln_parts);
      ln_a = ab[0];
      ln_b = ab[1];
      const int nb = n0 + wn * (16 * WNT);
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int n = (PAIR && i < 2 * (WNT / 2)) ? nb + 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : nb + 16 * i + 4 * fq;
        spre[i] = *reinterpret_cast<const f32x4*>(p.ln_s + (n + 4 <= p.N ? n : p.N - 4));
      }
    }
  }

  int buf = 0, nbuf = D;  // ring positions of tile kt and of tile kt + D
  // PF (the one-block-per-CU in-block split-K configuration: 64-row tiles, 8 waves, 4 stages): the fragments of k-tile kt + 1 are
  // requested at the top of k-tile kt and consumed one barrier later, so that no MFMA block waits out an LDS round trip (a k-tile
  // is 10 MFMAs per wave here: barrier + read latency + MFMAs in series was ~1100 cycles per k-tile for 160 cycles of matrix work).
  // Needs k-tile kt + 1 in LDS at the top of k-tile kt: the counted wait leaves one batch fewer in flight (two k-tiles of lead).
#ifdef CRG_GEMM_NOPF
  constexpr bool PF = false;
#else
  constexpr bool PF = KG == 2 && STAGES >= 4 && NS == 1;
#endif
  if constexpr (PF) {
    bf16x8 xf[2][WMT], wf[2][WNT];
    auto reads = [&](int set, int b) {
      const char* xs = smem + b * STAGE_BYTES;
      const char* ws = xs + XS_BYTES;
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xs + lds_off(wm * (16 * WMT) + j * 16 + frow, kg * 4 + fq));
        if (set == 0) xf[0][j] = v; else xf[1][j] = v;
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(ws + lds_off(wn * (16 * WNT) + i * 16 + frow, kg * 4 + fq));
        if (set == 0) wf[0][i] = v; else wf[1][i] = v;
      }
    };
    auto mm = [&](const bf16x8 (&x)[WMT], const bf16x8 (&w)[WNT]) {
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(w[i], x[j], acc[i][j]);
    };
    // top of k-tile kt (PARITY = which fragment set holds it): k-tile kt + 1 landed, barrier, issue kt + D, request kt + 1, multiply kt
    auto ktile = [&](int kt, auto PARc) {
      constexpr int par = decltype(PARc)::value;
      const int after = nk - 2 - kt;  // k-tiles issued after kt + 1 (they may stay in flight: at most D - 2)
      wait_batches<LOADS>(after < D - 2 ? after : D - 2);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_setprio(2);
      if (kt + D < nk) stage(nbuf);
      __builtin_amdgcn_s_setprio(0);
      const int b1 = (buf + 1 == STAGES) ? 0 : buf + 1;
      if (kt + 1 < nk) reads(par ^ 1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mm(xf[par], wf[par]);
      __builtin_amdgcn_sched_barrier(0);
      buf = b1;
      nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
    };
    if (kt_begin < nk) {
      // first k-tile: wait for it alone, publish, request its fragments (the only exposed read)
      const int newer = nk - 1 - kt_begin;
      wait_batches<LOADS>(newer < D - 1 ? newer : D - 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      reads(0, 0);
    }
    int kt = kt_begin;
    for (; kt + 1 < nk; kt += 2) {
      ktile(kt, std::integral_constant<int, 0>{});
      ktile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < nk) ktile(kt, std::integral_constant<int, 0>{});
  } else {
  for (int kt = kt_begin; kt < nk; ++kt) {
    const int newer = nk - 1 - kt;  // tiles issued after tile kt that may stay in flight (at most D - 1)
    if constexpr (D >= 4) wait_batches<LOADS>(newer < D - 1 ? newer : D - 1);
    else if (D >= 3 && newer >= 2) wait_vmcnt<(D >= 3 ? 2 : 0) * LOADS>();
    else if (D >= 2 && newer >= 1) wait_vmcnt<(D >= 2 ? 1 : 0) * LOADS>();
    else wait_vmcnt<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the DMA issue (address VALU + 1 KiB pieces) runs at raised wave priority: next to a wave of the CU's other block that
    // streams MFMAs it was issued late and its data landed late (conv -2..3 % per shape, GEMM neutral; raising the priority
    // of the MFMA phase instead was 4 % slower)
    __builtin_amdgcn_s_setprio(2);
    if (kt + D < nk) stage(nbuf);
    __builtin_amdgcn_s_setprio(0);
    const char* xs = smem + buf * STAGE_BYTES;
    const char* ws = xs + NS * XS_BYTES;
#pragma unroll
    for (int k2 = 0; k2 < 2 / KG; ++k2) {
      const int ks = KG == 2 ? kg : k2;
      bf16x8 xf[WMT], wf[WNT], xl[NS == 2 ? WMT : 1], wl[NS == 2 ? WNT : 1];
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const int off = lds_off(wm * (16 * WMT) + j * 16 + frow, ks * 4 + fq);
        xf[j] = *reinterpret_cast<const bf16x8*>(xs + off);
        if constexpr (NS == 2) xl[j] = *reinterpret_cast<const bf16x8*>(xs + XS_BYTES + off);
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int off = lds_off(wn * (16 * WNT) + i * 16 + frow, ks * 4 + fq);
        wf[i] = *reinterpret_cast<const bf16x8*>(ws + off);
        if constexpr (NS == 2) wl[i] = *reinterpret_cast<const bf16x8*>(ws + WS_BYTES + off);
      }
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          if constexpr (NS == 2) {
            acc[i][j] = CRG_MFMA_16x16x32(wl[i], xf[j], acc[i][j]);
            acc[i][j] = CRG_MFMA_16x16x32(wf[i], xl[j], acc[i][j]);
          }
          acc[i][j] = CRG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
        }
    }
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == STAGES) ? 0 : nbuf + 1;
  }
  }
  if constexpr (KG == 2) {
    // fold the second k-group's partial tile into the first: [wave&3][i][j][lane] f32x4 in the (now idle) ring
    static_assert(4 * WNT * WMT * 64 * 16 <= STAGES * STAGE_BYTES, "reduction buffer must fit in the DMA ring");
    __syncthreads();  // every wave is done reading the ring (all DMA batches were retired by the last counted wait)
    f32x4* red = reinterpret_cast<f32x4*>(smem) + ((wave & 3) * WNT * WMT) * 64 + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) red[(i * WMT + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] += red[(i * WMT + j) * 64];
  }
  if constexpr (LNE) {
    // y = rstd * (acc - mean * s[n]) (+ bias' in the epilogue): LayerNorm(x) W^T from the GEMM on the raw rows
    const float invk = 1.0f / (float)p.K;
    const float mean = ln_a * invk;
    float var = __builtin_fmaf(-mean, mean, ln_b * invk);
    var = var > 0.f ? var : 0.f;
    const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
    const float mrs = mean * rstd;
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const float rj = __shfl(rstd, j * 16 + frow), mj = __shfl(mrs, j * 16 + frow);
#pragma unroll
      for (int i = 0; i < WNT; ++i) acc[i][j] = rj * acc[i][j] - mj * spre[i];
    }
  }
  if constexpr (PAIR && !CONV && !GST && !RST) {
    if (p.vt && n0 >= p.vt_n0) {
      // transposed n-tile (the V third of a fused Q | K | V projection -> V^T [sample][channel][token], what the LDS-DMA / pipelined
      // attention kernels stage): lane = one token x 8 (4) consecutive channels -> one 2-byte store per channel; the 16 lanes of a
      // DPP row hold 16 consecutive tokens, i.e. 32 contiguous bytes of a V^T row per store instruction and row
      const int Cv = p.N - p.vt_n0;
      const int nb = n0 + wn * (16 * WNT);
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const int m = m0 + wm * (16 * WMT) + j * 16 + frow;
        if (m >= p.M) continue;
        const int bs = m / p.vt_T, tk = m - bs * p.vt_T;
        bf16* V = p.vt + ((long)bs * Cv - p.vt_n0) * p.vt_ld + tk;
#pragma unroll
        for (int u = 0; u < WNT / 2; ++u) {
          const int n = nb + 32 * u + 8 * fq;
          if (n >= p.N) continue;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = e < 4 ? acc[2 * u][j][e] : acc[2 * u + 1][j][e - 4];
            if (pre_bias) v += e < 4 ? bpre[2 * u][e] : bpre[2 * u + 1][e - 4];
            V[(long)(n + e) * p.vt_ld] = (bf16)v;
          }
        }
        if constexpr (WNT & 1) {
          const int n = nb + 16 * (WNT - 1) + 4 * fq;
          if (n < p.N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[WNT - 1][j][e];
              if (pre_bias) v += bpre[WNT - 1][e];
              V[(long)(n + e) * p.vt_ld] = (bf16)v;
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (PAIR) {
    gemm_epilogue_pairs<WNT, WMT, GST ? 1 : 0, RST>(p, acc, m0, n0, wm, wn, frow, fq, bz, r2, r1, pre_res, bpre, pre_bias);
  } else {
    gemm_epilogue<WNT, YT, WMT>(p, acc, m0, n0, wm, wn, frow, fq, bz, sid, rres, pre_res, bpre, pre_bias);
  }
}

// ------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convs whose image width divides 128 (16, 32, 64, 128: every UNet level above 8x8), chunk-major K.
// The plain implicit GEMM stages the activation tile once per TAP (9 x 16 KB per 64-channel chunk) although the three taps
// of one kernel row read the same image rows shifted by one pixel.  Here a 128-pixel tile = R = 128 / W whole image rows;
// per (chunk, kernel row kh) ONE buffer of R x (W + 2) pixels (the rows with a one-pixel halo left and right, zero outside
// the image) is staged and the three taps kw = 0..2 read their A fragments from it at a pixel offset of kw.  Activation
// bytes staged per chunk 144 -> 51 KB, all staged bytes per k-tile 36 -> 26 KB: the K loop of these kernels is bound by the
// LDS-DMA fill (in-kernel laps: DMA issue 46 % of the conv K loop), not by MFMA issue.
// LDS: weight ring 2 x BN x 128 B as before + 2 row buffers of <= 20 KB = 74-78 KB, still two blocks per CU.  The row buffer of
// group g + 1 is filled during the three k-tiles of group g (its pieces are spread over them); every k-tile starts with
// s_waitcnt vmcnt(0) + s_barrier, so each piece has landed and is visible before its first reader and the buffer it
// overwrites (group g - 1's) has no reader left.
// NS = 2 (with KG = 2: eight waves, the second k-group multiplies the second 32-wide k-step, folded through LDS at the end): the
// split-bf16 fp32-class form on pre-split planes (VAE) - activation hi / lo row buffers and weight hi / lo ring, three MFMAs per
// fragment pair; 132 KB, one block per CU like the 4-plane kernel it replaces, but 43 instead of 64 KB staged per k-tile.
// Widths that are multiples of 128 (VAE levels 128..512) use one row SEGMENT of 128 pixels per tile (R = 1, halo = the
// neighbouring pixels of the same row).
// MT = 2: 256-row tiles on eight waves (4 x 2), one block per CU: the weight k-tile is staged once per 256 rows (31 KB per
// k-tile and 256 x BN outputs instead of 2 x 26 KB).
// MX (round 4, NS = 2 / KG = 2 only): the fp32-class product on ONE fp16 pass plus two cross terms at fp8 precision on the block-scaled MX
// matrix instruction.  Plane 0 of both operands is fp16 (x16 = half(x), w16 = half(w)); plane 1 holds, per 64-deep k-tile and row, 64 e4m3
// bytes of the fp16 value ("hi8", scaled by a power of two) followed by 64 e4m3 bytes of the remainder x - x16 ("lo8") - the same 128 bytes
// per row and k-tile as a second bf16 plane, so the staging code does not change.  acc += w16 x16 (v_mfma_f32_16x16x32_f16, the group's
// 32-deep k-step) + BOTH cross terms of the k-tile in ONE v_mfma_scale_f32_16x16x128_f8f6f4 per output tile: its 128-deep contraction is
// the concatenation [w_lo8 | w_hi8] . [x_hi8 | x_lo8] - lane block l >> 4 = 0, 1 reads the lo8 half of the weight row and the hi8 half of the
// pixel row, block 2, 3 the other halves (32 bytes per lane and fragment) - which needs the two terms' scale products to be equal (they are:
// lo8 is stored at 2^11 x the hi8 scale on both sides).  The k-groups share the cross work by weight row tiles.  Per 16 x 16 x 64: 2 x 18
// cycles per k-group + 33.5 in one of them instead of 3 x 2 x 18 (tools/probes/mx_layout.hip); the cross terms only need ~4 bits: emulated
// pixel L-inf of the full VAE decode 8e-5 against 2e-5 for bf16 x 3, bound 1e-3 (tools/vae_precision_emul.py); measured 1.9e-5.
// STATUS (round 4): correct (per-op rel-L2 7.5e-6 against 3.2e-6 for bf16 x 3; full VAE decode pixel L-inf 1.9e-5 against 2.0e-5) but 3-13 %
// SLOWER than bf16 x 3 on the VAE's shapes (decoder 3x3 convs of a batch of four, tools/mx_probe.py: 23.1 ms against 21.5), so the VAE does
// not use it by default (ops.VAE_MX).  What was measured on the way: a first form that paired two k-tiles per cross
// instruction (exec-masked half-wave gathers, operands held across the barrier) 23.6 ms; that form without its cross MFMAs 15.5 ms and with
// neither gathers nor cross MFMAs 15.4 ms - the main fp16 pass alone, a third of today's matrix work, is 28 % faster; the eight scaled
// instructions per k-tile and wave then cost 7.7 ms where their bare issue time (33.5 cycles each beside 16 x 18 for the main pass) predicts
// about 3: in this loop the block-scaled instruction is about twice as expensive as in a bare loop.  Next: e2m3 cross operands (19.5
// cycles per instruction) and 32 x 32 x 64 tiles.
template <int WNT, typename YT, bool PAIR, int NS = 1, int KG = 1, int MT = 1, bool MX = false>
__global__ __launch_bounds__(256 * KG * MT, (KG == 1 && MT == 1) ? 2 : 1) void conv3_rowhalo_kernel(GemmP p) {
  static_assert(!PAIR || (sizeof(YT) == 2 && NS == 1), "paired columns: bf16 output, single-plane operands");
  static_assert(MT == 1 || (KG == 1 && NS == 1), "256-row tiles: single-plane operands, no in-block split-K");
  static_assert(!MX || (NS == 2 && KG == 2 && MT == 1), "MX: the split-plane form on two k-groups");
  constexpr int WMT = 4, NW = 4 * KG * MT;
  constexpr int TP = 128 * MT;  // tile pixels
  constexpr int BN = 32 * WNT;
  constexpr int WS_BYTES = BN * 128;
  constexpr int WRG = BN / 8;
  constexpr int WL = (WRG + NW - 1) / NW;
  constexpr int XI = ((MT == 2 ? 36 : 20) + NW - 1) / NW;  // row-buffer pieces (8 pixels = 1 KiB) per wave, group and plane: TP + 2 R pixels, W >= 16
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kg = MT == 2 ? 0 : wave >> 2;
  const int wm = MT == 2 ? wave >> 1 : (wave & 3) >> 1, wn = wave & 1;
  // geometry lives on the OUTPUT grid (Ho x Wo; the nearest-2x upsampled image when p.up), sources are read at (h >> up, w >> up)
  const int Wd = p.Wo, Hd = p.Ho, sh = p.up ? 1 : 0;
  // three buffer geometries: rows (W divides the tile: R whole image rows, each with a zero / neighbour pixel left and right),
  // seg (W a multiple of the tile: one row segment) and lin (any other width: the TP + 2 pixels m0 - 1 .. m0 + TP in linear
  // pixel order; a tap that would wrap around an image row is zeroed in the A fragment instead of in the buffer)
  const bool lin = p.halo_lin != 0;
  const bool seg = lin || Wd > TP;       // lin shares seg's addressing: buffer pixel b <-> linear pixel m0 - 1 + b
  const int WP = seg ? TP + 2 : Wd + 2;
  const int R = seg ? 1 : TP / Wd;
  const int xpix = R * WP;
  const int XP = (xpix + 7) >> 3;
  const int xbuf_bytes = XP * 1024;      // one plane of one row buffer
  char* const wring = smem;              // [stage][plane][BN x 128 B]
  char* const xbuf = smem + 2 * NS * WS_BYTES;  // [parity][plane][xbuf_bytes]

  int tile_m, tile_n, sid;
  block_to_tile(p, tile_m, tile_n, sid);
  const int m0 = tile_m * TP, n0 = tile_n * BN;
  const int bz = 0;
  const bf16* Wp = p.w;
  const long a_lo_off = NS == 2 ? reinterpret_cast<const bf16*>(p.a_lo) - reinterpret_cast<const bf16*>(p.a) : 0;
  const long w_lo_off = NS == 2 ? p.w_lo - p.w : 0;
  const bf16* zpage = p.zero_page;
  const int rsub = lane >> 3;
  const int clog = (lane & 7) ^ rsub;

  // row-buffer pieces of this lane: piece jp = wave + NW i covers buffer pixels 8 jp .. 8 jp + 7, this lane pixel 8 jp + rsub
  int xsrc[XI];        // source pixel index of (image, row 0, column w >> up), valid when a mask bit is set
  int xh[XI];          // output-grid row h of the buffer pixel: kernel row kh reads source row (h + kh - 1) >> up
  unsigned xmask[XI];  // bit kh: row h + kh - 1 inside the image (and pixel inside the tile / image / problem)
  {
    const int rows_total = p.M / Wd;  // N * H image rows
    const int row0 = m0 / Wd;
    const int w0 = m0 - row0 * Wd;    // first pixel of the segment (0 unless seg)
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int b = 8 * (wave + NW * i) + rsub;
      int grow, w;
      bool ok;
      if (lin) {
        const int pb = m0 - 1 + b;  // the pixel whose (kh - 1)-th row neighbour this buffer pixel holds
        ok = b < xpix && pb >= 0 && pb < p.M;
        grow = (pb < 0 ? 0 : pb) / Wd;
        w = pb - grow * Wd;
      } else {
        const int r = b / WP, col = b - r * WP;
        grow = row0 + r;
        w = w0 + col - 1;
        ok = b < xpix && grow < rows_total && w >= 0 && w < Wd;
      }
      const int img = grow / Hd, h = grow - img * Hd;
      unsigned msk = 0;
      if (ok) msk = (h >= 1 ? 1u : 0u) | 2u | (h + 1 < Hd ? 4u : 0u);
      xmask[i] = msk;
      xh[i] = h;
      xsrc[i] = img * p.H * p.W + (w >> sh);
    }
  }
  bool wok[WL];
  long wrow_off[WL];
#pragma unroll
  for (int q = 0; q < WL; ++q) {
    const int pos = (wave + NW * q) * 8 + rsub;
    const int n = n0 + (PAIR ? unpair_col<WNT>(pos) : pos);
    wok[q] = n < p.N && (wave + NW * q) < WRG;
    wrow_off[q] = (long)n * p.ldw;
  }
  // K slice of this block in (chunk, kernel row) groups of three k-tiles (ks_q / ks_r are in groups for this kernel)
  const int g_begin = sid * p.ks_q + (sid < p.ks_r ? sid : p.ks_r);
  const int g_end = g_begin + p.ks_q + (sid < p.ks_r ? 1 : 0);

  auto stage_x = [&](int g, int slot) {  // this wave's pieces i with i % 3 == slot of group g's row buffer(s)
    const int c = g / 3, kh = g - 3 * c;
    const int cch = c * 64 + clog * 8;
    const bool second = cch >= p.C1;
    const bf16* base = reinterpret_cast<const bf16*>(second ? p.x2 : p.a);
    const int Cs = second ? p.C2 : p.C1;
    const int cs = second ? cch - p.C1 : cch;
    char* xb = xbuf + ((g - g_begin) & 1) * (NS * xbuf_bytes);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      if (i % 3 != slot) continue;
      const int jp = wave + NW * i;
      if (jp < XP) {  // wave-uniform
        const bool ok = (xmask[i] >> kh) & 1u;
        const bf16* src = ok ? base + (long)(xsrc[i] + ((xh[i] + kh - 1) >> sh) * p.W) * Cs + cs : zpage;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xb + jp * 1024), 16, 0, 0);
        if constexpr (NS == 2) {
          const bf16* src2 = ok ? src + a_lo_off : zpage;
          __builtin_amdgcn_global_load_lds((gptr_t)src2, (lptr_t)(xb + xbuf_bytes + jp * 1024), 16, 0, 0);
        }
      }
    }
  };
  auto stage_w = [&](int kt, int buf) {  // weight k-tile kt (64 K-elements) of this block's BN rows
    char* ws = wring + buf * (NS * WS_BYTES);
    const long kc = (long)kt * BK + clog * 8;
#pragma unroll
    for (int q = 0; q < WL; ++q) {
      if ((wave + NW * q) < WRG) {  // wave-uniform
        const bf16* src = wok[q] ? Wp + wrow_off[q] + kc : zpage;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + (wave + NW * q) * 1024), 16, 0, 0);
        if constexpr (NS == 2) {
          const bf16* src2 = wok[q] ? src + w_lo_off : zpage;
          __builtin_amdgcn_global_load_lds((gptr_t)src2, (lptr_t)(ws + WS_BYTES + (wave + NW * q) * 1024), 16, 0, 0);
        }
      }
    }
  };

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fq = lane >> 4;
  // buffer pixel of this lane's output pixels for tap kw = 0 (tap kw adds kw): output pixel (r, w) of the tile sits at r * WP + w
  int xb0[WMT];
#pragma unroll
  for (int j = 0; j < WMT; ++j) {
    const int ml = wm * 64 + j * 16 + frow;
    const int r = seg ? 0 : ml / Wd;
    xb0[j] = r * WP + (ml - r * Wd);
  }
  // lin: output pixels in the first / last image column must not see the wrapped neighbour (taps kw = 0 / kw = 2)
  unsigned edge = 0;  // bit j: first column, bit 8 + j: last column
  if (lin) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const int m = m0 + wm * 64 + j * 16 + frow;
      const int wcol = m % Wd;
      edge |= (wcol == 0 ? 1u : 0u) << j;
      edge |= (wcol == Wd - 1 ? 1u : 0u) << (8 + j);
    }
  }

  // residual / bias prefetch: as in gemm_glds_kernel (unconditional loads on clamped addresses)
  bf16x4 rres[PAIR ? 1 : WNT][PAIR ? 1 : WMT];
  bf16x8 r2[PAIR ? (WNT / 2 > 0 ? WNT / 2 : 1) : 1][PAIR ? WMT : 1];
  bf16x4 r1[PAIR ? WMT : 1];
  f32x4 bpre[WNT];
  const bool pre_res = sizeof(YT) == 2 && p.res && p.splits == 1 && (p.ldr & 3) == 0 && (p.N & 3) == 0;
  const bool pre_bias = !MX && p.bias_mode == CRG_BIAS_COL && p.splits == 1 && (p.N & 3) == 0;  // (MX: no registers to spare: the epilogue loads it)
  if (kg == 0) {
    const int nb = n0 + wn * (16 * WNT);
    if (pre_res) {
      const bf16* Rp = reinterpret_cast<const bf16*>(p.res);
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        const bf16* Rm = Rp + (long)(m < p.M ? m : p.M - 1) * p.ldr;
        if constexpr (PAIR) {
#pragma unroll
          for (int u = 0; u < WNT / 2; ++u) {
            const int n = nb + 32 * u + 8 * fq;
            r2[u][j] = *reinterpret_cast<const bf16x8*>(Rm + (n + 8 <= p.N ? n : p.N - 8));
          }
          if constexpr (WNT & 1) {
            const int n = nb + 16 * (WNT - 1) + 4 * fq;
            r1[j] = *reinterpret_cast<const bf16x4*>(Rm + (n + 4 <= p.N ? n : p.N - 4));
          }
        } else {
#pragma unroll
          for (int i = 0; i < WNT; ++i) {
            const int n = nb + i * 16 + fq * 4;
            rres[i][j] = *reinterpret_cast<const bf16x4*>(Rm + (n + 4 <= p.N ? n : p.N - 4));
          }
        }
      }
    }
    if (pre_bias) {
#pragma unroll
      for (int i = 0; i < WNT; ++i) {
        const int n = (PAIR && i < 2 * (WNT / 2)) ? nb + 32 * (i >> 1) + 8 * fq + 4 * (i & 1) : nb + 16 * i + 4 * fq;
        bpre[i] = *reinterpret_cast<const f32x4*>(p.bias + (n + 4 <= p.N ? n : p.N - 4));
      }
    }
  }

  if (g_begin < g_end) {
    stage_x(g_begin, 0);
    stage_x(g_begin, 1);
    stage_x(g_begin, 2);
    stage_w(3 * g_begin, 0);
  }
  // KG == 2: the two k-groups are the two waves of every SIMD (wave w and w + 4).  Run in lockstep they read their fragments together
  // and multiply together, and the matrix pipe idles during the reads.  The second group therefore runs its MFMAs ONE K-TILE LATE, from
  // fragments it keeps in registers across the barrier: per k-tile it first multiplies the previous k-tile's fragments (while its partner
  // reads) and then reads this k-tile's (while its partner multiplies) - MI355X_MICROARCH.md "Two waves per SIMD" item 9.  No extra LDS:
  // a slot is refilled one barrier after the group's reads of it have returned, as before.
#ifdef CRG_X3_NOSTAGGER
  constexpr bool STAG = false;
#else
  // (the 160-wide split-plane tile has no registers left for a loop-carried fragment set; neither has the MX form: with its cross operands
  //  loop-carried as well it spills 92 bytes per lane and measured 16 % slower than its lockstep loop)
  constexpr bool STAG = KG == 2 && WNT == 4 && !MX;
#endif
  constexpr bool LO = NS == 2 && !MX;  // a second bf16 plane multiplied on the same instruction (bf16 x 3)
  bf16x8 sxf[STAG ? WMT : 1], swf[STAG ? WNT : 1], sxl[STAG && LO ? WMT : 1], swl[STAG && LO ? WNT : 1];
  if constexpr (STAG) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) { sxf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; if constexpr (LO) sxl[j] = sxf[j]; }
#pragma unroll
    for (int i = 0; i < WNT; ++i) { swf[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; if constexpr (LO) swl[i] = swf[i]; }
  }
  // MX: cross-term operand registers (32 e4m3 bytes per lane and fragment): this k-group's weight row tiles [ci0, ci0 + CW) and all pixel tiles
  typedef int v8i __attribute__((ext_vector_type(8)));
  typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
  constexpr int CW = MX ? (WNT + 1) / 2 : 1;
  const int ci0 = kg * CW;
  v8i wc[CW], xc[MX ? WMT : 1];
  if constexpr (MX) {  // (the staggered group multiplies once before its first reads)
#pragma unroll
    for (int q = 0; q < CW; ++q) wc[q] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < WMT; ++j) xc[j] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
  }
  const int mx_sa = MX ? p.mx_scale[0][0] : 0, mx_sb = MX ? p.mx_scale[0][1] : 0;  // E8M0 scales: W lo8 and X hi8 (= the hi8 x lo8 product)
  auto mx_reads = [&](const char* ws, const char* xs, int kw) {
    if constexpr (MX) {
      const int cwq = ((fq + 2) & 3) * 2, cxq = fq * 2;  // 16-byte chunks of the 128-byte record: W [lo8 | hi8] order, X [hi8 | lo8]
#pragma unroll
      for (int q = 0; q < CW; ++q) {
        const int i = ci0 + q < WNT ? ci0 + q : WNT - 1;
        const int row = wn * (16 * WNT) + i * 16 + frow;
        const u32x4 c0 = *reinterpret_cast<const u32x4*>(ws + WS_BYTES + lds_off(row, cwq));
        const u32x4 c1 = *reinterpret_cast<const u32x4*>(ws + WS_BYTES + lds_off(row, cwq + 1));
        wc[q] = v8i{(int)c0[0], (int)c0[1], (int)c0[2], (int)c0[3], (int)c1[0], (int)c1[1], (int)c1[2], (int)c1[3]};
      }
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        const u32x4 c0 = *reinterpret_cast<const u32x4*>(xs + xbuf_bytes + lds_off(xb0[j] + kw, cxq));
        const u32x4 c1 = *reinterpret_cast<const u32x4*>(xs + xbuf_bytes + lds_off(xb0[j] + kw, cxq + 1));
        xc[j] = v8i{(int)c0[0], (int)c0[1], (int)c0[2], (int)c0[3], (int)c1[0], (int)c1[1], (int)c1[2], (int)c1[3]};
      }
      if (kw != 1 && lin) {  // block-uniform: a tap that would wrap around an image row contributes zeros
#pragma unroll
        for (int j = 0; j < WMT; ++j)
          if ((edge >> (kw == 0 ? j : 8 + j)) & 1u) xc[j] = v8i{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  };
  auto mx_cross = [&]() {
    if constexpr (MX) {
#pragma unroll
      for (int q = 0; q < CW; ++q) {
        if (ci0 + q < WNT) {  // wave-uniform (odd WNT: the second group has one row tile fewer)
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            // (acc[kg * CW + q] with a compile-time index in each arm: no dynamic register indexing)
            if (kg == 0) acc[q][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wc[q], xc[j], acc[q][j], 0, 0, 0, mx_sa, 0, mx_sb);
            else acc[(CW + q) < WNT ? CW + q : 0][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wc[q], xc[j], acc[(CW + q) < WNT ? CW + q : 0][j], 0, 0, 0, mx_sa, 0, mx_sb);
          }
        }
      }
    }
  };
  auto stag_mma = [&]() {
    if constexpr (STAG) {
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          if constexpr (LO) {
            acc[i][j] = CRG_MFMA_16x16x32(swl[i], sxf[j], acc[i][j]);
            acc[i][j] = CRG_MFMA_16x16x32(swf[i], sxl[j], acc[i][j]);
          }
          if constexpr (MX) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, swf[i]), __builtin_bit_cast(h16x8, sxf[j]), acc[i][j], 0, 0, 0);
          else acc[i][j] = CRG_MFMA_16x16x32(swf[i], sxf[j], acc[i][j]);
        }
      mx_cross();
    }
  };
  int wbuf = 0;
  auto k_loop = [&](auto GBc) {
  constexpr bool GB = decltype(GBc)::value;
  for (int g = g_begin; g < g_end; ++g) {
    const char* xs = xbuf + ((g - g_begin) & 1) * (NS * xbuf_bytes);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      wait_vmcnt<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_setprio(2);
      if (kw < 2 || g + 1 < g_end) stage_w(3 * g + kw + 1, wbuf ^ 1);
      if (g + 1 < g_end) stage_x(g + 1, kw);
      __builtin_amdgcn_s_setprio(0);
      const char* ws = wring + wbuf * (NS * WS_BYTES);
      if constexpr (STAG) {
        // one fragment set for both groups; only the ORDER differs: group 0 reads, then multiplies; group 1 multiplies (the
        // previous k-tile's fragments, zeros the first time), then reads
        auto stag_reads = [&]() {
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            const int off = lds_off(xb0[j] + kw, kg * 4 + fq);
            sxf[j] = *reinterpret_cast<const bf16x8*>(xs + off);
            if constexpr (LO) sxl[j] = *reinterpret_cast<const bf16x8*>(xs + xbuf_bytes + off);
          }
          if (kw != 1 && lin) {
#pragma unroll
            for (int j = 0; j < WMT; ++j) {
              if ((edge >> (kw == 0 ? j : 8 + j)) & 1u) {
                sxf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (LO) sxl[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
              }
            }
          }
#pragma unroll
          for (int i = 0; i < WNT; ++i) {
            const int off = lds_off(wn * (16 * WNT) + i * 16 + frow, kg * 4 + fq);
            swf[i] = *reinterpret_cast<const bf16x8*>(ws + off);
            if constexpr (LO) swl[i] = *reinterpret_cast<const bf16x8*>(ws + WS_BYTES + off);
          }
          mx_reads(ws, xs, kw);
        };
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GB) {
          stag_mma();
          __builtin_amdgcn_sched_barrier(0);
          stag_reads();
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the reads have RETURNED before the next barrier lets the slot be refilled
        } else {
          stag_reads();
          __builtin_amdgcn_sched_barrier(0);
          stag_mma();
        }
      } else {
#pragma unroll
      for (int k2 = 0; k2 < 2 / KG; ++k2) {
        const int ks = KG == 2 ? kg : k2;
        bf16x8 xf[WMT], wf[WNT], xl[NS == 2 ? WMT : 1], wl[NS == 2 ? WNT : 1];
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          const int off = lds_off(xb0[j] + kw, ks * 4 + fq);
          xf[j] = *reinterpret_cast<const bf16x8*>(xs + off);
          if constexpr (NS == 2 && !MX) xl[j] = *reinterpret_cast<const bf16x8*>(xs + xbuf_bytes + off);
        }
        if (kw != 1 && lin) {  // block-uniform branch: nothing on the hot path of the other geometries
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            if ((edge >> (kw == 0 ? j : 8 + j)) & 1u) {
              xf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
              if constexpr (NS == 2) xl[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
          }
        }
#pragma unroll
        for (int i = 0; i < WNT; ++i) {
          const int off = lds_off(wn * (16 * WNT) + i * 16 + frow, ks * 4 + fq);
          wf[i] = *reinterpret_cast<const bf16x8*>(ws + off);
          if constexpr (NS == 2 && !MX) wl[i] = *reinterpret_cast<const bf16x8*>(ws + WS_BYTES + off);
        }
        if constexpr (MX) {
          mx_reads(ws, xs, kw);
#pragma unroll
          for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int j = 0; j < WMT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wf[i]), __builtin_bit_cast(h16x8, xf[j]), acc[i][j], 0, 0, 0);
          mx_cross();
        } else {
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            if constexpr (NS == 2) {
              acc[i][j] = CRG_MFMA_16x16x32(wl[i], xf[j], acc[i][j]);
              acc[i][j] = CRG_MFMA_16x16x32(wf[i], xl[j], acc[i][j]);
            }
            acc[i][j] = CRG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
          }
        }
      }
      }
      wbuf ^= 1;
    }
  }
  if constexpr (STAG && GB) stag_mma();  // the last k-tile's fragments
  };
  if (STAG && kg == 1) k_loop(std::integral_constant<bool, true>{});
  else k_loop(std::integral_constant<bool, false>{});
  if constexpr (KG == 2) {
    // fold the second k-group's partial tile into the first: [wave & 3][i][j][lane] f32x4 in the (now idle) weight ring / row buffers
    static_assert(KG == 1 || 4 * WNT * WMT * 64 * 16 <= 2 * NS * WS_BYTES + 2 * NS * 17 * 1024, "reduction buffer must fit in the block's LDS");
    __syncthreads();  // every wave is done reading (all DMA was retired by the last vmcnt(0))
    f32x4* red = reinterpret_cast<f32x4*>(smem) + ((wave & 3) * WNT * WMT) * 64 + lane;
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) red[(i * WMT + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] += red[(i * WMT + j) * 64];
  }
  if constexpr (PAIR) {
    gemm_epilogue_pairs<WNT, WMT>(p, acc, m0, n0, wm, wn, frow, fq, bz, r2, r1, pre_res, bpre, pre_bias);
  } else {
    gemm_epilogue<WNT, YT, WMT>(p, acc, m0, n0, wm, wn, frow, fq, bz, sid, rres, pre_res, bpre, pre_bias);
  }
}

// Split-K second pass: y = epi(sum_s slab[s] + bias) + cvec + residual, 4 consecutive n per thread.
template <typename YT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p) {
  const int n4 = p.N >> 2;
  const long srows = p.M - p.slab_row0;
  const long total = srows * n4;
  const int bz = blockIdx.y;
  const float* S = p.slab + (long)bz * p.splits * srows * p.N - (long)p.slab_row0 * p.N;
  YT* Y = reinterpret_cast<YT*>(p.y) + (long)bz * p.y_bs;
  const YT* R = p.res ? reinterpret_cast<const YT*>(p.res) + (long)bz * p.r_bs : nullptr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int mr = (int)(idx / n4);
    const int m = mr + p.slab_row0;
    const int n = (int)(idx - (long)mr * n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(S + (long)m * p.N + n);
    for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(S + ((long)s * srows + m) * p.N + n);
    if (p.bias_mode == CRG_BIAS_COL) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    else if (p.bias_mode == CRG_BIAS_ROW) v += p.bias[m];
    if (p.epi == CRG_EPI_SILU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = crg_silu_f(v[e]);
    }
    if (p.cvec) v += *reinterpret_cast<const f32x4*>(p.cvec + (long)(m / p.cvec_rows) * p.cvec_ld + n);
    if (R) {
      const YT* rp = R + (long)m * p.ldr + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)rp[e];
    }
    YT* yp = Y + (long)m * p.ldy + n;
    if ((p.ldy & 3) == 0) {
      if constexpr (sizeof(YT) == 2) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16)v[e];
        *reinterpret_cast<bf16x4*>(yp) = o4;
      } else {
        *reinterpret_cast<f32x4*>(yp) = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) yp[e] = (YT)v[e];
    }
  }
}

// Split-K second pass that ALSO emits the GroupNorm statistics side channel (GemmP::gstat): same arithmetic per element as
// splitk_reduce_kernel, but a block owns a 32-row x 128-column patch - thread = (row lane 0..7: rows 4 rl .. 4 rl + 3, column group
// of 4 channels) - so that the per-channel sums of the rounded outputs over the 32 rows are a fixed-order fold of 8 row lanes
// through LDS.  grid (row blocks, column patches).
template <typename YT>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) float red[8 * 32 * 8];
  const int t = threadIdx.x, cg = t & 31, rl = t >> 5;
  const long srows = p.M - p.slab_row0;
  const float* S = p.slab - (long)p.slab_row0 * p.N;
  YT* Y = reinterpret_cast<YT*>(p.y);
  const YT* R = p.res ? reinterpret_cast<const YT*>(p.res) : nullptr;
  const int n = blockIdx.y * 128 + cg * 4;
  const int mb = p.slab_row0 + blockIdx.x * 32;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < p.N) {
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + rl * 4 + i;
      const int mc = m < p.M ? m : p.M - 1;  // clamped: the loads stay unconditional, the result of a row past M is dropped
      v[i] = *reinterpret_cast<const f32x4*>(S + (long)mc * p.N + n);
      for (int s = 1; s < p.splits; ++s) v[i] += *reinterpret_cast<const f32x4*>(S + ((long)s * srows + mc) * p.N + n);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + rl * 4 + i;
      if (m >= p.M) continue;
      f32x4 w = v[i];
      if (p.bias_mode == CRG_BIAS_COL) w += *reinterpret_cast<const f32x4*>(p.bias + n);
      else if (p.bias_mode == CRG_BIAS_ROW) w += p.bias[m];
      if (p.epi == CRG_EPI_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = crg_silu_f(w[e]);
      }
      if (p.cvec) w += *reinterpret_cast<const f32x4*>(p.cvec + (long)(m / p.cvec_rows) * p.cvec_ld + n);
      if (R) {
        const YT* rp = R + (long)m * p.ldr + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] += (float)rp[e];
      }
      YT o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (YT)w[e];
        const float f = (float)o[e];
        s1[e] += f;
        s2[e] += f * f;
      }
      YT* yp = Y + (long)m * p.ldy + n;
      if constexpr (sizeof(YT) == 2) *reinterpret_cast<uint2*>(yp) = *reinterpret_cast<const uint2*>(o);
      else *reinterpret_cast<f32x4*>(yp) = *reinterpret_cast<const f32x4*>(o);
    }
  }
  float* q = red + (rl * 32 + cg) * 8;
  *reinterpret_cast<f32x4*>(q) = f32x4{s1[0], s1[1], s1[2], s1[3]};
  *reinterpret_cast<f32x4*>(q + 4) = f32x4{s2[0], s2[1], s2[2], s2[3]};
  __syncthreads();
  if (rl == 0 && n < p.N) {
    f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
    for (int r2 = 1; r2 < 8; ++r2) {
      a += *reinterpret_cast<const f32x4*>(q + r2 * 32 * 8);
      b += *reinterpret_cast<const f32x4*>(q + r2 * 32 * 8 + 4);
    }
    const long rb = mb >> 5;
    *reinterpret_cast<f32x4*>(p.gstat + rb * p.N + n) = a;
    *reinterpret_cast<f32x4*>(p.gstat + p.gstat_plane + rb * p.N + n) = b;
  }
}

// Split-K second pass that ALSO emits the LayerNorm row statistics (GemmP::rstat) of the finished rows: a wave owns a row at a time (lanes
// stride over its 4-column groups), sums the rounded outputs and their squares, folds the 64 lanes with shuffles (fixed order) and writes
// the whole-row sums into partial 0; the other partials of the row are zeroed (the consumer folds all rstat_parts of them).
__global__ __launch_bounds__(256) void splitk_reduce_rows_kernel(GemmP p) {
  const int n4 = p.N >> 2;
  const long srows = p.M - p.slab_row0;
  const float* S = p.slab - (long)p.slab_row0 * p.N;
  bf16* Y = reinterpret_cast<bf16*>(p.y);
  const bf16* R = reinterpret_cast<const bf16*>(p.res);
  const int lane = threadIdx.x & 63;
  for (int m = p.slab_row0 + blockIdx.x * 4 + (threadIdx.x >> 6); m < p.M; m += gridDim.x * 4) {
    float q1 = 0.f, q2 = 0.f;
    for (int c = lane; c < n4; c += 64) {
      const int n = c * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(S + (long)m * p.N + n);
      for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(S + ((long)s * srows + m) * p.N + n);
      if (p.bias_mode == CRG_BIAS_COL) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      else if (p.bias_mode == CRG_BIAS_ROW) v += p.bias[m];
      if (R) {
        const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(R + (long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
      }
      bf16x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o4[e] = (bf16)v[e];
        const float f = (float)o4[e];
        q1 += f;
        q2 = __builtin_fmaf(f, f, q2);
      }
      *reinterpret_cast<bf16x4*>(Y + (long)m * p.ldy + n) = o4;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      q1 += __shfl_xor(q1, o);
      q2 += __shfl_xor(q2, o);
    }
    if (lane < p.rstat_parts) *reinterpret_cast<f32x2*>(p.rstat + ((long)m * p.rstat_parts + lane) * 2) = lane == 0 ? f32x2{q1, q2} : f32x2{0.f, 0.f};
  }
}

// Split-K second pass FUSED with the GroupNorm(+SiLU) that consumes the conv's output (openaimodel.py:208 -> :229-231 inside a ResBlock),
// for the small images of the two lowest UNet levels: one block per (sample, group) sums the K slices of its HW x gs slab in slice order,
// adds bias / per-sample vector / residual in the order of splitk_reduce_kernel, rounds to bf16 and stores the raw output, and then
// normalises those very values with the arithmetic of gn_small_kernel (shifted sums, wave shuffles, fixed-order cross-wave fp64
// combine) - bitwise what the reduce kernel followed by the single-launch GroupNorm produce, in one launch and without re-reading y.
template <int VPT>
__global__ __launch_bounds__(256) void splitk_reduce_gn_kernel(GemmP p) {
  __shared__ float red[2][4];
  __shared__ float shift_s;
  const int n = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  const int C = p.N, HW = p.gn_hw;
  const int gs = C / p.gn_groups, wv = gs >> 3;
  const int c_beg = g * gs;
  const int total = HW * wv;
  const long slice = (long)p.M * p.N;
  const float* S = p.slab;
  bf16* Y = reinterpret_cast<bf16*>(p.y);
  const bf16* R = reinterpret_cast<const bf16*>(p.res);
  crg_vec8<bf16> v[VPT];
  int row[VPT], cv[VPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = t + 256 * k;
    row[k] = i / wv;
    cv[k] = i - row[k] * wv;
    if (i < total) {
      const long m = (long)n * HW + row[k];
      const int c = c_beg + cv[k] * 8;
      const float* sp = S + m * p.N + c;
      f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
      for (int s2 = 1; s2 < p.splits; ++s2) {
        a += *reinterpret_cast<const f32x4*>(sp + s2 * slice);
        b += *reinterpret_cast<const f32x4*>(sp + s2 * slice + 4);
      }
      if (p.bias_mode == CRG_BIAS_COL) {
        a += *reinterpret_cast<const f32x4*>(p.bias + c);
        b += *reinterpret_cast<const f32x4*>(p.bias + c + 4);
      }
      if (p.cvec) {
        const float* cp = p.cvec + (m / p.cvec_rows) * p.cvec_ld + c;
        a += *reinterpret_cast<const f32x4*>(cp);
        b += *reinterpret_cast<const f32x4*>(cp + 4);
      }
      if (R) {
        const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(R + m * p.ldr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] += (float)r8[e];
          b[e] += (float)r8[4 + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[k].set(e, a[e]);
        v[k].set(4 + e, b[e]);
      }
      v[k].store(Y + m * p.ldy + c);
    }
  }
  if (t == 0) shift_s = v[0].get(0);  // element (row 0, first channel of the group) of the rounded tensor: gn_small_kernel's shift
  __syncthreads();
  const float shift = shift_s;
  float s1 = 0.f, s2 = 0.f;
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
  const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
  bf16* yb = reinterpret_cast<bf16*>(p.gn_y) + (long)n * HW * C + c_beg;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if (t + 256 * k < total) {
      const int c = c_beg + cv[k] * 8;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gn_gamma + c), g1 = *reinterpret_cast<const f32x4*>(p.gn_gamma + c + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.gn_beta + c), b1 = *reinterpret_cast<const f32x4*>(p.gn_beta + c + 4);
      crg_vec8<bf16> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
        float f = (v[k].get(e) - mean) * (rstd * ga) + be;
        if (p.gn_silu) f = crg_silu_f(f);
        o.set(e, f);
      }
      o.store(yb + (long)row[k] * C + cv[k] * 8);
    }
  }
}

// can the reduce behind this split-K launch also run the GroupNorm the caller asked for?
template <typename YT>
bool gn_slab_ok(const GemmP& p, int batch) {
  if (sizeof(YT) != 2 || !p.gn_y || !p.gn_gamma || !p.gn_beta || batch != 1 || p.slab_row0 != 0 || p.gstat || p.vt || p.rstat) return false;
  if (p.epi != CRG_EPI_NONE || (p.bias_mode != CRG_BIAS_NONE && p.bias_mode != CRG_BIAS_COL)) return false;
  if (p.gn_groups <= 0 || p.N % p.gn_groups || p.gn_hw <= 0 || p.M % p.gn_hw || p.ldy != p.N) return false;
  const int gs = p.N / p.gn_groups;
  if (gs % 8 || (p.N & 7) || (p.res && (p.ldr & 7)) || (p.cvec && (p.cvec_ld & 3))) return false;
  return (long)p.gn_hw * (gs >> 3) <= 256 * 8;
}

// the reduce launch behind a split-K launch (whole problem, or - tail split - the rows >= slab_row0)
template <typename YT>
int launch_reduce(crg_ctx* ctx, hipStream_t st, const GemmP& p, int batch) {
  const long rows = p.M - p.slab_row0;
  crg_prof_scope ps(ctx, st, CRG_K_SPLITK, (double)batch * p.splits * rows * p.N, (double)batch * rows * p.N * (4.0 * p.splits + sizeof(YT)));
  if (gn_slab_ok<YT>(p, batch)) {
    const int vpt = (int)(((long)p.gn_hw * (p.N / p.gn_groups >> 3) + 255) / 256);
    const dim3 grid(p.gn_groups, p.M / p.gn_hw);
    switch (vpt) {
      case 1: hipLaunchKernelGGL(splitk_reduce_gn_kernel<1>, grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL(splitk_reduce_gn_kernel<2>, grid, dim3(256), 0, st, p); break;
      case 3: hipLaunchKernelGGL(splitk_reduce_gn_kernel<3>, grid, dim3(256), 0, st, p); break;
      case 4: hipLaunchKernelGGL(splitk_reduce_gn_kernel<4>, grid, dim3(256), 0, st, p); break;
      case 5: hipLaunchKernelGGL(splitk_reduce_gn_kernel<5>, grid, dim3(256), 0, st, p); break;
      case 6: hipLaunchKernelGGL(splitk_reduce_gn_kernel<6>, grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL(splitk_reduce_gn_kernel<8>, grid, dim3(256), 0, st, p); break;
    }
    CRG_CHECK_LAUNCH(ctx, "splitk_reduce_gn");
    ctx->gn_fused = true;
    return 0;
  }
  if (p.rstat) {
    if (sizeof(YT) != 2 || batch != 1 || (p.N & 3) || (p.ldy & 3) || p.epi != CRG_EPI_NONE || p.cvec || p.rstat_parts > 64 || (p.res && (p.ldr & 3)))
      return crg_fail(ctx, -22, "gemm: LayerNorm row statistics behind split-K need an unbatched plain bf16 problem with 4-aligned N / ldy / ldr");
    const long blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_rows_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, p);
  } else if (p.gstat) {
    if (batch != 1 || (p.N & 3) || (p.ldy & 3) || (p.slab_row0 & 31) || (p.res && (p.ldr & 3)))
      return crg_fail(ctx, -22, "gemm: GroupNorm statistics need an unbatched problem with 4-aligned N / ldy / ldr");
    hipLaunchKernelGGL(splitk_reduce_stats_kernel<YT>, dim3((unsigned)((rows + 31) / 32), (unsigned)((p.N + 127) / 128)), dim3(256), 0, st, p);
  } else {
    const long total4 = rows * (p.N >> 2);
    const int rg = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel<YT>, dim3(rg, batch), dim3(256), 0, st, p);
  }
  CRG_CHECK_LAUNCH(ctx, "splitk_reduce");
  return 0;
}

// Split-K factor: small-M problems (8x8 / 16x16 UNet levels, token GEMMs with few rows) launch far fewer
// than 2 blocks per CU and stream long K; cut K so that ~512 blocks are in flight, keeping >= 8 k-tiles
// per slice.  Costs one fp32 slab round trip + one extra launch, so only when the tile count is low.
inline int choose_splits(const GemmP& p, int tiles, int batch) {
  if (p.epi == CRG_EPI_GEGLU || (p.N & 3) || p.ln_stat) return 1;  // (the LayerNorm correction is per whole row sum: no K slices)
  const int nk = (p.K + BK - 1) / BK;
  const long blocks = (long)tiles * batch;
  // 256..511 blocks leave one block (4 waves) on most CUs, which hides neither the barrier nor the DMA latency: cut K in
  // two when it is long enough to amortise the slab pass (measured: 8x32x32 640->640 conv 112 -> 89 us, 1280->640 203 -> 137 us;
  // K = 2560 GEMMs lose 5 %, hence the nk bound)
  // short K (nk < 24) is not split: measured slower than the 64-row / 8-wave configuration on the unsplit problem
  // (128..255 tiles with a K of 24..63 k-tiles - the 16x16-level 1x1 skip convs over a concat, 2048 x 1280 x 1920 / 2560 - are faster
  // unsplit on the 64-row / 8-wave configuration: 30.3 -> 24.0 us and 33.6 -> 30.5 us, tools/c1_probe.py, round 3)
  if (blocks >= 512 || nk < 24 || (blocks >= 128 && nk < 64)) return 1;
  int s;
  if (blocks >= 256) {
    // one to two blocks per CU and a long K: among 2..4 slices take the one whose blocks fill whole rounds of 512 best
    // (256 tiles -> 2, 288 tiles (768x768 latents at the 48x48 level) -> 3, 384 -> 4); ties go to fewer slices
    s = 2;
    double best = 1e9;
    for (int c = 2; c <= 4; ++c) {
      const double t = (double)((blocks * c + 511) / 512) / c;
      if (t < best - 1e-9) {
        best = t;
        s = c;
      }
    }
    return s;
  }
  static const int tgt = getenv("CRG_SPLIT_BLOCKS") ? atoi(getenv("CRG_SPLIT_BLOCKS")) : 512;  // dev knob: blocks aimed at by the K cut
  static const int smax = getenv("CRG_SPLIT_MAX") ? atoi(getenv("CRG_SPLIT_MAX")) : 16;          // dev knob: most K slices
  s = (int)((tgt + blocks - 1) / blocks);
  if (s > nk / 8) s = nk / 8;
  if (s > smax) s = smax;
  return s < 2 ? 1 : s;
}

struct Work {  // algorithmic work of one call (profiler) and operand footprints (XCD partition)
  double flops, bytes, a_bytes, w_bytes;
};

// (xg_m, xg_n, xg_s): see block_to_tile.  Minimise fabric traffic xg_n*|A| + xg_m*|W| over the feasible factorizations of 8.
inline void choose_xcd_partition(GemmP& p, const Work& wk) {
  p.xg_m = p.xg_n = 1;
  p.xg_s = 0;
  double best = 0.0;
  for (int gs = 8; gs >= 1; gs >>= 1) {
    if (p.splits % gs) continue;
    for (int gn = 1; gn * gs <= 8; gn <<= 1) {
      const int gm = 8 / (gs * gn);
      if (p.tiles_n % gn || p.tiles_m % gm) continue;
      const double cost = gn * wk.a_bytes + gm * wk.w_bytes;
      if (!p.xg_s || cost < best) {
        best = cost;
        p.xg_m = gm; p.xg_n = gn; p.xg_s = gs;
      }
    }
  }
}

template <int WNT, int NSPLIT, typename AT, typename YT, bool CONV, int STAGES, int WMT, int KG>
int launch_kernel(crg_ctx* ctx, hipStream_t st, GemmP& p, int batch, Work wk) {
  constexpr int BN = 32 * WNT;
  constexpr bool GLDS = (NSPLIT == 1) && (sizeof(AT) == 2);
  constexpr size_t lds = GLDS ? (size_t)STAGES * (32 * WMT + BN) * 128 + 1024 /* dummy-load target */ : (size_t)2 * NSPLIT * (BM + BN) * 128;
  {
    const int nk_total = (p.K + BK - 1) / BK;
    p.ks_q = nk_total / p.splits;
    p.ks_r = nk_total % p.splits;
  }
  p.pair = (GLDS && sizeof(YT) == 2 && p.splits == 1 && p.epi != CRG_EPI_GEGLU && (p.N & 7) == 0 && (p.ldy & 7) == 0 &&
            (p.y_bs & 7) == 0 && ((uintptr_t)p.y & 15) == 0 &&
            (!p.res || ((p.ldr & 7) == 0 && (p.r_bs & 7) == 0 && ((uintptr_t)p.res & 15) == 0)) &&
            (!p.cvec || (p.cvec_ld & 3) == 0)) ? 1 : 0;
  if (p.vt && !(p.pair && p.splits == 1 && !p.gstat && !CONV))
    return crg_fail(ctx, -22, "gemm: a transposed column range needs the paired bf16 epilogue of an unsplit, unbatched GEMM (N, ldy multiples of 8, K below the split-K rule)");
  if (p.gstat && !p.pair && p.splits == 1)
    return crg_fail(ctx, -22, "gemm/conv: GroupNorm statistics come from the paired bf16 epilogue (N, ldy, ldr multiples of 8, 16-byte aligned y / residual, no GEGLU)");
  if (p.rstat && !(!p.gstat && !p.vt && !CONV && batch == 1 && sizeof(YT) == 2 && p.rstat_parts == 2 * p.tiles_n && p.epi == CRG_EPI_NONE &&
                   ((p.splits == 1 && p.pair) || (p.splits > 1 && (p.N & 3) == 0 && (p.ldy & 3) == 0 && (!p.res || (p.ldr & 3) == 0)))))
    return crg_fail(ctx, -22, "gemm: LayerNorm row statistics come from the paired bf16 epilogue of an unbatched plain GEMM (N, ldy multiples of 8) or from its split-K reduce, with row_stats_parts = 2 * ceil(N / tile) = %d (got %d)", 2 * p.tiles_n, p.rstat_parts);
  if (p.ln_stat && !(GLDS && sizeof(YT) == 2 && !CONV && p.splits == 1 && batch == 1 && !p.res && !p.gstat && !p.rstat && !p.cvec && (p.N & 3) == 0 &&
                     (p.bias_mode == CRG_BIAS_COL || p.bias_mode == CRG_BIAS_NONE) && (p.epi == CRG_EPI_NONE || p.epi == CRG_EPI_GEGLU)))
    return crg_fail(ctx, -22, "gemm: the LayerNorm epilogue needs an unsplit, unbatched bf16 GEMM (N %% 4 == 0) without residual / statistics / SiLU");
  void (*kern)(GemmP);
  if constexpr (GLDS && sizeof(YT) == 2 && !CONV) {
    if (p.ln_stat) kern = p.pair ? gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, true, 3> : gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, false, 3>;
    else if (p.rstat && p.splits == 1) kern = gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, true, 2>;
    else
      kern = p.pair ? ((p.gstat && p.splits == 1) ? gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, true, 1> : gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, true>)
                    : gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG>;
  } else if constexpr (GLDS && sizeof(YT) == 2)
    kern = p.pair ? ((p.gstat && p.splits == 1) ? gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, true, 1> : gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG, 1, true>)
                  : gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG>;
  else if constexpr (GLDS) kern = gemm_glds_kernel<WNT, YT, CONV, STAGES, WMT, KG>;
  else kern = gemm_kernel<WNT, NSPLIT, AT, YT, CONV, NSPLIT>;  // split-bf16: 8 waves (two k-groups)
  size_t lds_bytes = lds;
  bool halo = false;
  int threads = GLDS ? 256 * KG : 256 * NSPLIT;
  if constexpr (GLDS && CONV && STAGES == 2 && WMT == 4 && KG == 1 && sizeof(YT) == 2 && (WNT == 4 || WNT == 5)) {
    static const int knob = getenv("CRG_ROWHALO") ? atoi(getenv("CRG_ROWHALO")) : 2;  // dev knob: 0 = plain implicit GEMM
    const int ngroups = (p.K / BK) / 3;
    if (knob && p.rowhalo && batch == 1 && p.splits <= ngroups) {
      halo = true;
      const int TP = p.rowhalo == 2 ? 256 : 128;
      if (p.rowhalo == 2) {
        kern = p.pair ? conv3_rowhalo_kernel<WNT, YT, true, 1, 1, 2> : conv3_rowhalo_kernel<WNT, YT, false, 1, 1, 2>;
        threads = 512;
      } else {
        kern = p.pair ? conv3_rowhalo_kernel<WNT, YT, true> : conv3_rowhalo_kernel<WNT, YT, false>;
      }
      const int XP = (p.halo_lin || p.Wo > TP) ? (TP + 2 + 7) / 8 : ((TP / p.Wo) * (p.Wo + 2) + 7) / 8;
      lds_bytes = (size_t)2 * BN * 128 + (size_t)2 * XP * 1024;
      p.ks_q = ngroups / p.splits;  // K slices in (chunk, kernel row) groups
      p.ks_r = ngroups % p.splits;
    }
  }
  {
    const size_t cap = halo ? (size_t)(p.rowhalo == 2 ? 116 : 80) * 1024 : lds;
    if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), cap, "gemm")) return rc;
  }
  dim3 grid(p.tile_count * p.splits, batch, 1);
  constexpr int slot = !GLDS ? (CONV ? CRG_K_CONV_X3 : CRG_K_GEMM_X3)
                             : (CONV ? (WNT == 5 ? CRG_K_CONV_W5 : WNT == 4 ? CRG_K_CONV_W4 : CRG_K_CONV_W1)
                                     : (WNT == 5 ? CRG_K_GEMM_W5 : WNT == 4 ? CRG_K_GEMM_W4 : CRG_K_GEMM_W1));
  crg_prof_scope ps(ctx, st, slot, wk.flops, wk.bytes);
  if constexpr (GLDS && CONV && STAGES == 2 && WMT == 4 && KG == 1 && sizeof(YT) == 2 && (WNT == 4 || WNT == 5)) {
    if (halo && p.rowhalo == 2 && p.ring) {
      // one statistics partial per 256-pixel tile and channel instead of eight (crg_groupnorm_pre then needs no finalise launch):
      // paired epilogue, no K slices, tiles that do not straddle samples
      if (p.gstat && p.gstat_tile_ok && p.pair && p.splits == 1 && (p.Ho * p.Wo) % 256 == 0) p.gstat_rows = 256;
      return launch_conv_pp(ctx, st, p, WNT);
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds_bytes, st, p);
  CRG_CHECK_LAUNCH(ctx, "gemm");
  return 0;
}

// Tile / pipeline configuration of the LDS-DMA kernel, by how many blocks the problem yields (256 CUs):
//   A  128 x BN tile, 4 waves, 2-deep ring  : >= ~1.5 blocks per CU (after split-K); two blocks share a CU and hide each
//                                             other's latencies
//   D   64 x BN tile, 4 waves, 2-deep ring  : 192..1023 tiles of 128 rows (GEGLU epilogues: 192..383) and a short K: twice the blocks
//                                             (4096x1280x1280: A 26.7, 8-wave 128-row variant 25.5, D 23.6 us).  Round 3, device time
//                                             inside a captured graph (tools/lin_probe.py): 32768x320x320 + residual 25.8 (A) -> 22.4,
//                                             8192x1920x640 41.6 -> 35.3, 2048x3840x1280 41.8 -> 34.6, 32768x320x1280 49.8 -> 45.3 us; the
//                                             GEGLU GEMMs (128-wide tiles, heavy epilogue) keep A: 8192x5120x640 82 (A) vs 89 (D)
//   C   64 x BN tile, 8 waves (two k-groups, folded through LDS), 4-deep ring : < 192 tiles (16x16 / 8x8 token GEMMs): one
//                                             block per CU at most, so the waves and the ring depth come from inside the block
//                                             (2048x1280x1280: A 24.7, C 16.7 us)
template <int WNT, int NSPLIT, typename AT, typename YT, bool CONV>
int launch(crg_ctx* ctx, hipStream_t st, GemmP& p, int batch, Work wk) {
  constexpr int BN = 32 * WNT;
  constexpr bool GLDS = (NSPLIT == 1) && (sizeof(AT) == 2);
  p.zero_page = (const bf16*)ctx->zero_page;
  p.tiles_n = (p.N + BN - 1) / BN;
  static const int force = getenv("CRG_GEMM_CFG") ? atoi(getenv("CRG_GEMM_CFG")) : 0;  // dev knob: 1 = A, 3 = C, 4 = D
  // split-K first, on 128-row tiles: a long K is the cheapest source of blocks.  Only problems that stay below ~1 block per
  // CU after that (short K) change the tile / wave configuration.
  p.tiles_m = (p.M + 127) / 128;
  p.splits = choose_splits(p, p.tiles_n * p.tiles_m, batch);
  if constexpr (GLDS && !CONV && sizeof(YT) == 2) {
    // persistent 256-row tiles (gemm_ring.hip) where every CU gets at least (almost) one: more FLOP per staged byte, no per-tile prologue
    if (ring_gemm_ok(p, batch, ctx->n_cu)) return launch_gemm_ring(ctx, st, p, wk.flops, wk.bytes);
  }
  int cfg = 1;
  if (GLDS && p.splits == 1) {
    const long blocks = (long)p.tiles_m * p.tiles_n * batch;
    static const int d_max = getenv("CRG_GEMM_D_MAX") ? atoi(getenv("CRG_GEMM_D_MAX")) : 384;      // dev knob: 64-row tiles below this many 128-row tiles (plain epilogues)
    static const int d_max_g = getenv("CRG_GEMM_D_MAXG") ? atoi(getenv("CRG_GEMM_D_MAXG")) : 384;  // ... GEGLU epilogues
    static const int d_max_c1 = getenv("CRG_CONV1_D_MAX") ? atoi(getenv("CRG_CONV1_D_MAX")) : 384;   // ... 1x1 convs (virtual-concat skip convs)
    // (3x3 convs keep 384: above it they run on the row-halo / ring kernels)
    if (blocks < 192) cfg = 3;
    else if (blocks < (CONV ? (p.ks == 1 ? d_max_c1 : d_max_g) : (p.epi == CRG_EPI_GEGLU ? d_max_g : d_max))) cfg = 4;
  }
  if (GLDS && force) cfg = force;
  if (cfg == 3 || cfg == 4) {
    p.tiles_m = (p.M + 63) / 64;
    p.splits = force ? choose_splits(p, p.tiles_n * p.tiles_m, batch) : 1;
  }
  if constexpr (GLDS && CONV && sizeof(YT) == 2 && (WNT == 4 || WNT == 5)) {
    // row-halo conv on 256-row tiles (one 8-wave block per CU): same K slices, half the m-tiles
    static const int knob = getenv("CRG_ROWHALO") ? atoi(getenv("CRG_ROWHALO")) : 2;  // dev knob: 0 plain implicit GEMM, 1 128-row tiles only
    const int ngroups = (p.K / BK) / 3;
    if (knob == 2 && cfg == 1 && p.rowhalo && batch == 1 && p.splits <= ngroups && p.M % 256 == 0 &&
        (p.halo_lin || (p.Wo <= 256 ? 256 % p.Wo == 0 : p.Wo % 256 == 0))) {
      // one block per CU: only when the grid fills its rounds of 256 blocks (a 288-block grid would run a second round 1/8 full)
      const long blocks2 = (long)(p.M / 256) * p.tiles_n * p.splits;
      const long rounds = (blocks2 + 255) / 256;
      if (blocks2 * 100 >= rounds * 256 * 85) {
        p.rowhalo = 2;
        p.tiles_m = p.M / 256;
      }
    }
  }
  p.ring = 0;
  if constexpr (GLDS && CONV && sizeof(YT) == 2 && (WNT == 4 || WNT == 5)) {
    // the staggered-wave kernel (conv_pp.hip) for the 256-row configuration, where its buffer descriptors can address the operands
    if (p.rowhalo == 2 && p.a_bytes && p.w_bytes && (p.C2 == 0 || p.x2_bytes) && (long)p.M / (p.Ho * p.Wo) * p.H * p.W < (1 << 24)) p.ring = 1;
  }
  if (p.splits > 1) {
    const size_t bytes = (size_t)batch * p.splits * p.M * p.N * sizeof(float);
    p.slab = (float*)crg_scratch(ctx, bytes);
    if (!p.slab) return crg_fail(ctx, -12, "gemm: out of scratch for %d split-K slabs", p.splits);
  }
  choose_xcd_partition(p, wk);
  p.tile_base = 0;
  p.tile_count = p.tiles_n * p.tiles_m;
  p.slab_row0 = 0;
  int rc;
  if constexpr (GLDS) {
    // Tail split: with two blocks per CU a grid runs in rounds of 512 blocks; T = 512 k + r with a small r costs a whole extra
    // round for r blocks (768x768 latents: 576 tiles -> the second round is 1/8 full and the conv runs at 56 % of its rate).
    // The first 512 k tiles go out as they are; the last r tiles - whole rows of m-tiles - are cut along K so that they
    // fill the chip once more for a fraction of a round, and are summed by the split-K reduce over just their rows.
    const int T = p.tile_count, nk = (p.K + BK - 1) / BK;
    const int r = T % 512, T1 = T - r;
    // (not with a transposed column range: the K-split tail would have to emit V^T from the reduce pass - ADVICE r3)
    if (cfg == 1 && p.rowhalo != 2 && p.splits == 1 && batch == 1 && T1 >= 512 && r > 0 && r <= 224 && nk >= 16 && r % p.tiles_n == 0 &&
        p.epi != CRG_EPI_GEGLU && !(p.N & 3) && !p.vt && !p.rstat && !p.ln_stat) {
      int s2 = 512 / r;
      if (s2 > nk / 4) s2 = nk / 4;
      if (s2 > 8) s2 = 8;
      if (s2 >= 2) {
        const int row0 = (T1 / p.tiles_n) * 128;
        float* slab = (float*)crg_scratch(ctx, (size_t)s2 * (p.M - row0) * p.N * sizeof(float));
        if (!slab) return crg_fail(ctx, -12, "gemm: out of scratch for the tail slab");
        p.xg_s = 0;  // partial ranges use the contiguous-run order
        p.tile_count = T1;
        rc = launch_kernel<WNT, NSPLIT, AT, YT, CONV, 2, 4, 1>(ctx, st, p, batch, Work{wk.flops * T1 / T, wk.bytes * T1 / T, 0, 0});
        if (rc) return rc;
        p.tile_base = T1;
        p.tile_count = r;
        p.splits = s2;
        p.slab = slab;
        p.slab_row0 = row0;
        rc = launch_kernel<WNT, NSPLIT, AT, YT, CONV, 2, 4, 1>(ctx, st, p, batch, Work{wk.flops * r / T, wk.bytes * r / T, 0, 0});
        if (rc) return rc;
        return launch_reduce<YT>(ctx, st, p, 1);
      }
    }
#ifndef CRG_C_STAGES  // dev knob (tools/build_one_variant.sh): ring depth of configuration C
#define CRG_C_STAGES 4
#endif
    if (cfg == 3) rc = launch_kernel<WNT, NSPLIT, AT, YT, CONV, CRG_C_STAGES, 2, 2>(ctx, st, p, batch, wk);
    else if (cfg == 4) rc = launch_kernel<WNT, NSPLIT, AT, YT, CONV, 2, 2, 1>(ctx, st, p, batch, wk);
    else rc = launch_kernel<WNT, NSPLIT, AT, YT, CONV, 2, 4, 1>(ctx, st, p, batch, wk);
  } else {
    rc = launch_kernel<WNT, NSPLIT, AT, YT, CONV, 2, 4, 1>(ctx, st, p, batch, wk);
  }
  if (rc) return rc;
  if (p.splits > 1) return launch_reduce<YT>(ctx, st, p, batch);
  return 0;
}

// fp32-class operands that arrive as pre-split bf16 planes: LDS-DMA kernel with NS = 2, 128 x BN tile, 8 waves in two
// k-groups, 2-deep ring (4 planes x 2 stages fill the LDS: one block per CU, so the second wave per SIMD comes from the k-groups).
template <int WNT, bool CONV>
int launch_planes(crg_ctx* ctx, hipStream_t st, GemmP& p, int batch, Work wk) {
  constexpr int BN = 32 * WNT;
  constexpr size_t lds = (size_t)2 * 2 * (128 + BN) * 128 + 1024;
  p.zero_page = (const bf16*)ctx->zero_page;
  if (p.gstat && (batch != 1 || (p.N & 3) || (p.M & 31) || (p.ldy & 3)))
    return crg_fail(ctx, -22, "gemm/conv: GroupNorm statistics need an unbatched problem with N %% 4 == 0 and M %% 32 == 0");
  p.tiles_n = (p.N + BN - 1) / BN;
  p.tiles_m = (p.M + 127) / 128;
  p.splits = p.gstat ? 1 : choose_splits(p, p.tiles_n * p.tiles_m, batch);  // the statistics come out of the epilogue: no K split
  if (p.splits > 1) {
    p.slab = (float*)crg_scratch(ctx, (size_t)batch * p.splits * p.M * p.N * sizeof(float));
    if (!p.slab) return crg_fail(ctx, -12, "gemm: out of scratch for %d split-K slabs", p.splits);
  }
  choose_xcd_partition(p, wk);
  p.tile_base = 0;
  p.tile_count = p.tiles_n * p.tiles_m;
  p.slab_row0 = 0;
  void (*kern)(GemmP) = gemm_glds_kernel<WNT, float, CONV, 2, 4, 2, 2>;
  size_t lds_bytes = lds;
  int units = (p.K + BK - 1) / BK;  // K slices in k-tiles ...
  bool halo = false;
  if constexpr (CONV && (WNT == 4 || WNT == 5)) {
    static const int knob = getenv("CRG_ROWHALO") ? atoi(getenv("CRG_ROWHALO")) : 1;  // dev knob: 0 = plain implicit GEMM
    const int ngroups = (p.K / BK) / 3;
    if (knob && p.rowhalo && batch == 1 && p.splits <= ngroups) {
      halo = true;
      kern = p.mx ? conv3_rowhalo_kernel<WNT, float, false, 2, 2, 1, true> : conv3_rowhalo_kernel<WNT, float, false, 2, 2>;
      const int XP = (p.halo_lin || p.Wo > 128) ? 17 : ((128 / p.Wo) * (p.Wo + 2) + 7) / 8;
      lds_bytes = (size_t)2 * 2 * BN * 128 + (size_t)2 * 2 * XP * 1024;
      units = ngroups;  // ... or in (chunk, kernel row) groups
    }
  }
  {
    const size_t cap = halo ? (size_t)(2 * 2 * BN * 128 + 2 * 2 * 18 * 1024) : lds;  // eligible widths need <= 18 pieces per row buffer
    if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), cap, "gemm(planes)")) return rc;
  }
  if (p.mx && !halo) return crg_fail(ctx, -22, "conv2d: the MX form takes 3x3 / stride 1 / pad 1 convs with Cin %% 64 == 0 (the row-halo kernel's domain)");
  {
    p.ks_q = units / p.splits;
    p.ks_r = units % p.splits;
    p.pair = 0;
    crg_prof_scope ps(ctx, st, CONV ? CRG_K_CONV_X3 : CRG_K_GEMM_X3, wk.flops, wk.bytes);
    hipLaunchKernelGGL(kern, dim3(p.tiles_n * p.tiles_m * p.splits, batch, 1), dim3(512), lds_bytes, st, p);
    CRG_CHECK_LAUNCH(ctx, "gemm(planes)");
  }
  if (p.splits > 1) {
    const long total4 = (long)p.M * (p.N >> 2);
    const int rg = (int)((total4 + 255) / 256 > 2048 ? 2048 : (total4 + 255) / 256);
    crg_prof_scope ps(ctx, st, CRG_K_SPLITK, (double)batch * p.splits * p.M * p.N, (double)batch * p.M * p.N * (4.0 * p.splits + 4));
    hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(rg, batch), dim3(256), 0, st, p);
    CRG_CHECK_LAUNCH(ctx, "splitk_reduce");
  }
  return 0;
}

template <int NSPLIT, typename AT, typename YT, bool CONV>
int launch_wnt(crg_ctx* ctx, hipStream_t st, GemmP& p, int batch, Work wk) {
  // 160-wide tiles when they divide N (all UNet widths are multiples of 320), else 128-wide.
  const bool geglu = p.epi == CRG_EPI_GEGLU;
  if (!geglu && p.N <= 32) return launch<1, NSPLIT, AT, YT, CONV>(ctx, st, p, batch, wk);  // conv_out-like thin outputs
  if (!geglu && p.N % 160 == 0) return launch<5, NSPLIT, AT, YT, CONV>(ctx, st, p, batch, wk);
  return launch<4, NSPLIT, AT, YT, CONV>(ctx, st, p, batch, wk);
}

template <bool CONV>
int dispatch(crg_ctx* ctx, hipStream_t st, GemmP& p, int batch, int a_dtype, int y_dtype, int prec, Work wk) {
  // GroupNorm statistics: the bf16 path (paired epilogue / split-K reduce), or the fp32-class conv on pre-split planes (its fp32 epilogue)
  const bool gstat_bf16 = prec == CRG_PREC_BF16 && a_dtype == CRG_BF16 && y_dtype == CRG_BF16 && (p.N & 7) == 0;
  const bool gstat_planes = (prec == CRG_PREC_BF16X3 || prec == CRG_PREC_F16MX) && a_dtype == CRG_BF16 && y_dtype == CRG_F32 && p.a_lo && !p.a_is_weight && (p.N & 3) == 0;
  if (p.gstat && !((gstat_bf16 || gstat_planes) && batch == 1 && p.epi == CRG_EPI_NONE))
    return crg_fail(ctx, -22, "gemm/conv: GroupNorm statistics need bf16 in / out (N %% 8 == 0) or pre-split planes in / fp32 out (N %% 4 == 0), batch 1 and a plain epilogue");
  if (prec == CRG_PREC_BF16) {
    if (a_dtype == CRG_BF16 && y_dtype == CRG_BF16) return launch_wnt<1, bf16, bf16, CONV>(ctx, st, p, batch, wk);
    if (a_dtype == CRG_BF16 && y_dtype == CRG_F32) return launch_wnt<1, bf16, float, CONV>(ctx, st, p, batch, wk);
    if (a_dtype == CRG_F32 && y_dtype == CRG_BF16) return launch_wnt<1, float, bf16, CONV>(ctx, st, p, batch, wk);
    if (a_dtype == CRG_F32 && y_dtype == CRG_F32) return launch_wnt<1, float, float, CONV>(ctx, st, p, batch, wk);
  } else if (prec == CRG_PREC_F16MX) {
    if (CONV && p.mx && y_dtype == CRG_F32 && p.a_lo && p.w_lo && p.epi == CRG_EPI_NONE) {
      if (p.N % 160 == 0) return launch_planes<5, CONV>(ctx, st, p, batch, wk);
      return launch_planes<4, CONV>(ctx, st, p, batch, wk);
    }
  } else if (prec == CRG_PREC_BF16X3) {
    if (a_dtype == CRG_BF16 && y_dtype == CRG_F32 && p.a_lo && !p.a_is_weight && p.epi == CRG_EPI_NONE) {
      if (p.N <= 32) return launch_planes<1, CONV>(ctx, st, p, batch, wk);
      if (p.N % 160 == 0) return launch_planes<5, CONV>(ctx, st, p, batch, wk);
      return launch_planes<4, CONV>(ctx, st, p, batch, wk);
    }
    if (a_dtype == CRG_F32 && y_dtype == CRG_F32) return launch_wnt<2, float, float, CONV>(ctx, st, p, batch, wk);
    if (!CONV && a_dtype == CRG_BF16 && y_dtype == CRG_F32 && p.a_is_weight)
      return launch_wnt<2, bf16, float, false>(ctx, st, p, batch, wk);
  }
  return crg_fail(ctx, -22, "gemm/conv: unsupported dtype/precision combination a=%d y=%d prec=%d", a_dtype, y_dtype, prec);
}

}  // namespace

extern "C" int crg_gemm(crg_ctx* ctx, void* stream, const crg_gemm_args* a) {
  if (!ctx || !a) return -22;
  CRG_REQUIRE(ctx, a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0, "gemm: empty problem M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  CRG_REQUIRE(ctx, a->K % 8 == 0, "gemm: K=%d must be a multiple of 8", a->K);
  const int ae = (a->a_dtype == CRG_F32) ? 4 : 8;  // elements per 16 B
  CRG_REQUIRE(ctx, a->lda % ae == 0 && a->ldw % 8 == 0, "gemm: lda=%ld / ldw=%ld must keep rows 16-byte aligned", (long)a->lda, (long)a->ldw);
  CRG_REQUIRE(ctx, ((uintptr_t)a->a & 15) == 0 && ((uintptr_t)a->w & 15) == 0, "gemm: operand pointers must be 16-byte aligned");
  CRG_REQUIRE(ctx, a->prec == CRG_PREC_BF16 || a->w_lo, "gemm: BF16X3 needs the lo weight plane");
  if (a->epilogue == CRG_EPI_GEGLU) {
    CRG_REQUIRE(ctx, a->N % 32 == 0, "gemm: GEGLU needs packed N %% 32 == 0 (got %d)", a->N);
    CRG_REQUIRE(ctx, !a->residual, "gemm: GEGLU epilogue takes no residual");
  }
  if (a->bias && a->bias_mode == CRG_BIAS_COL)
    CRG_REQUIRE(ctx, ((uintptr_t)a->bias & 15) == 0, "gemm: column bias must be 16-byte aligned");
  GemmP p{};
  p.a = a->a; p.a_lo = a->a_lo; p.lda = a->lda; p.a_bs = a->a_bstride;
  p.w = (const bf16*)a->w; p.w_lo = (const bf16*)a->w_lo; p.ldw = a->ldw; p.w_bs = a->w_bstride;
  p.bias = a->bias; p.bias_mode = a->bias ? a->bias_mode : CRG_BIAS_NONE;
  p.res = a->residual; p.ldr = a->ldr; p.r_bs = a->r_bstride;
  p.y = a->y; p.ldy = a->ldy; p.y_bs = a->y_bstride;
  p.M = a->M; p.N = a->N; p.K = a->K; p.epi = a->epilogue;
  p.cvec = nullptr; p.cvec_rows = 1; p.cvec_ld = 0; p.a_is_weight = a->a_is_weight;
  p.gstat = a->gn_stats; p.gstat_plane = (long)((a->M + 31) / 32) * a->N;
  p.gstat_rows = 32;
  if (a->gn_stats) CRG_REQUIRE(ctx, ((uintptr_t)a->gn_stats & 15) == 0, "gemm: gn_stats must be 16-byte aligned");
  if (a->vt) {
    const int bn_ = a->N % 160 == 0 ? 160 : 128;
    CRG_REQUIRE(ctx, a->prec == CRG_PREC_BF16 && a->a_dtype == CRG_BF16 && a->y_dtype == CRG_BF16 && a->batch == 1 && a->epilogue == CRG_EPI_NONE &&
                         !a->residual && !a->gn_stats && (a->bias_mode == CRG_BIAS_COL || !a->bias),
                "gemm: a transposed column range needs a plain bf16 GEMM (no batch / residual / activation / statistics)");
    CRG_REQUIRE(ctx, a->vt_n0 > 0 && a->vt_n0 < a->N && a->vt_n0 % bn_ == 0 && a->N > 32 && a->vt_tokens > 0 && a->M % a->vt_tokens == 0 && a->vt_ld >= a->vt_tokens &&
                         ((uintptr_t)a->vt & 1) == 0,
                "gemm: transposed range n0=%d (tile %d) tokens=%d ld=%ld inconsistent with M=%d N=%d", a->vt_n0, bn_, a->vt_tokens, (long)a->vt_ld, a->M, a->N);
    p.vt = (bf16*)a->vt; p.vt_n0 = a->vt_n0; p.vt_T = a->vt_tokens; p.vt_ld = a->vt_ld;
  }
  if (a->row_stats) {
    CRG_REQUIRE(ctx, a->prec == CRG_PREC_BF16 && a->a_dtype == CRG_BF16 && a->y_dtype == CRG_BF16 && a->batch == 1 && a->epilogue == CRG_EPI_NONE &&
                         !a->gn_stats && !a->vt && !a->ln_stats && (a->N & 7) == 0 && a->row_stats_parts > 0 && a->row_stats_parts <= 64 && ((uintptr_t)a->row_stats & 15) == 0,
                "gemm: row_stats need a plain unbatched bf16 GEMM with N %% 8 == 0 (no GroupNorm statistics / transposed range / LayerNorm epilogue)");
    p.rstat = a->row_stats; p.rstat_parts = a->row_stats_parts;
  }
  if (a->ln_stats) {
    CRG_REQUIRE(ctx, a->prec == CRG_PREC_BF16 && a->a_dtype == CRG_BF16 && a->y_dtype == CRG_BF16 && a->batch == 1 && !a->residual && !a->gn_stats &&
                         (a->epilogue == CRG_EPI_NONE || a->epilogue == CRG_EPI_GEGLU) && (a->bias_mode == CRG_BIAS_COL || !a->bias) && (a->N & 3) == 0,
                "gemm: the LayerNorm epilogue needs an unbatched bf16 GEMM, N %% 4 == 0, epilogue NONE / GEGLU, no residual / row bias / statistics");
    CRG_REQUIRE(ctx, a->ln_colsum && a->ln_parts >= 2 && a->ln_parts <= 16 && (a->ln_parts & 1) == 0 && a->ln_eps >= 0.f &&
                         (((uintptr_t)a->ln_colsum | (uintptr_t)a->ln_stats) & 15) == 0,
                "gemm: ln_stats needs ln_colsum, an even number of 2..16 partials per row and 16-byte aligned pointers");
    p.ln_stat = a->ln_stats; p.ln_parts = a->ln_parts; p.ln_s = a->ln_colsum; p.ln_eps = a->ln_eps;
  }
  const double flops = 2.0 * a->M * (double)a->N * a->K * a->batch;
  const double bytes = ((double)a->M * a->K * crg_dtype_size(a->a_dtype) + (double)a->N * a->K * 2 +
                        (double)a->M * a->N * crg_dtype_size(a->y_dtype) * (a->residual ? 2 : 1)) * a->batch;
  return dispatch<false>(ctx, (hipStream_t)stream, p, a->batch, a->a_dtype, a->y_dtype, a->prec,
                         Work{flops, bytes, (double)a->M * a->K * crg_dtype_size(a->a_dtype), (double)a->N * a->K * 2});
}

extern "C" int crg_conv2d(crg_ctx* ctx, void* stream, const crg_conv_args* a) {
  if (!ctx || !a) return -22;
  const int Ctot = a->C1 + a->C2;
  CRG_REQUIRE(ctx, a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0 && Ctot > 0, "conv2d: empty problem");
  CRG_REQUIRE(ctx, a->ksize == 3 || a->ksize == 1, "conv2d: ksize %d unsupported", a->ksize);
  CRG_REQUIRE(ctx, a->C1 % 8 == 0 && a->C2 % 8 == 0, "conv2d: channel counts must be multiples of 8 (C1=%d C2=%d); use crg_conv3x3_small", a->C1, a->C2);
  CRG_REQUIRE(ctx, (a->C2 == 0) == (a->x2 == nullptr), "conv2d: x2/C2 mismatch");
  CRG_REQUIRE(ctx, a->prec == CRG_PREC_BF16 || a->w_lo, "conv2d: BF16X3 / F16MX need the second weight plane");
  CRG_REQUIRE(ctx, a->Cout % 4 == 0 || !a->cvec, "conv2d: a per-sample channel vector needs Cout %% 4 == 0 (got %d)", a->Cout);
  CRG_REQUIRE(ctx, ((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->w & 15) == 0 && ((uintptr_t)a->bias & 15) == 0 && ((uintptr_t)a->cvec & 15) == 0,
              "conv2d: pointers must be 16-byte aligned");
  const int Hv = a->upsample2x ? 2 * a->H : a->H, Wv = a->upsample2x ? 2 * a->W : a->W;
  CRG_REQUIRE(ctx, a->Ho > 0 && a->Wo > 0 && (a->Ho - 1) * a->stride - a->pad_t < Hv && (a->Wo - 1) * a->stride - a->pad_l < Wv,
              "conv2d: output %dx%d inconsistent with input %dx%d stride %d", a->Ho, a->Wo, Hv, Wv, a->stride);
  if (a->x_lo) {
    CRG_REQUIRE(ctx, (a->prec == CRG_PREC_BF16X3 || a->prec == CRG_PREC_F16MX) && a->x_dtype == CRG_BF16 && a->y_dtype == CRG_F32 && !a->x2 && ((uintptr_t)a->x_lo & 15) == 0,
                "conv2d: pre-split activations (x_lo) need prec BF16X3 / F16MX, 16-bit planes, fp32 output, no second input");
  }
  if (a->prec == CRG_PREC_F16MX) {
    CRG_REQUIRE(ctx, a->x_lo && a->w_lo && a->ksize == 3 && a->stride == 1 && Ctot % 64 == 0 && !a->upsample2x,
                "conv2d: F16MX takes 3x3 / stride 1 convs on MX planes (x = fp16 plane, x_lo = e4m3 pair plane, same for w / w_lo), Cin %% 64 == 0");
    for (int i = 0; i < 4; ++i) CRG_REQUIRE(ctx, a->mx_log2[i] >= -100 && a->mx_log2[i] <= 100, "conv2d: mx_log2[%d] = %d out of range", i, a->mx_log2[i]);
    CRG_REQUIRE(ctx, a->mx_log2[1] - a->mx_log2[0] == a->mx_log2[3] - a->mx_log2[2],
                "conv2d: F16MX computes both cross terms in one instruction: the lo8 / hi8 scale ratio must be the same for w (%d) and x (%d)",
                a->mx_log2[1] - a->mx_log2[0], a->mx_log2[3] - a->mx_log2[2]);
  }
  GemmP p{};
  p.a = a->x; p.a_lo = a->x_lo; p.x2 = a->x2; p.C1 = a->C1; p.C2 = a->C2; p.Ctot = Ctot;
  p.w = (const bf16*)a->w; p.w_lo = (const bf16*)a->w_lo; p.ldw = (long)a->ksize * a->ksize * Ctot; p.w_bs = 0;
  p.bias = a->bias; p.bias_mode = a->bias ? CRG_BIAS_COL : CRG_BIAS_NONE;
  p.res = a->residual; p.ldr = a->Cout; p.r_bs = 0;
  p.y = a->y; p.ldy = a->Cout; p.y_bs = 0;
  p.M = a->N * a->Ho * a->Wo; p.N = a->Cout; p.K = a->ksize * a->ksize * Ctot; p.epi = CRG_EPI_NONE;
  p.cvec = a->cvec; p.cvec_rows = a->Ho * a->Wo; p.cvec_ld = a->cvec_ld ? a->cvec_ld : a->Cout;
  p.H = a->H; p.W = a->W; p.Ho = a->Ho; p.Wo = a->Wo; p.ks = a->ksize; p.stride = a->stride;
  p.pad_t = a->pad_t; p.pad_l = a->pad_l; p.up = a->upsample2x;
  if (a->prec == CRG_PREC_F16MX) {
    // mx_log2 = {w hi8, w lo8, x hi8, x lo8}: each plane stores value * 2^s, the instruction's E8M0 scale undoes it (127 - s).
    // k-group 0 multiplies w lo8 by x hi8, k-group 1 w hi8 by x lo8.
    p.mx = 1;
    p.mx_scale[0][0] = 127 - a->mx_log2[1]; p.mx_scale[0][1] = 127 - a->mx_log2[2];
    p.mx_scale[1][0] = 127 - a->mx_log2[0]; p.mx_scale[1][1] = 127 - a->mx_log2[3];
  }
  p.gstat = a->gn_stats; p.gstat_plane = (long)((p.M + 31) / 32) * p.N;
  p.gstat_rows = 32;
  p.gstat_tile_ok = a->gn_stats != nullptr && a->gn_stats_rows != nullptr;
  if (a->gn_stats) CRG_REQUIRE(ctx, ((uintptr_t)a->gn_stats & 15) == 0 && (a->Ho * a->Wo) % 32 == 0, "conv2d: gn_stats must be 16-byte aligned and Ho * Wo a multiple of 32");
  p.cm = (a->ksize == 3 && Ctot % 64 == 0) ? 1 : 0;  // must match crg_pack_weight's layout rule
  p.rowhalo = (p.cm && a->stride == 1 && a->pad_t == 1 && a->pad_l == 1 && a->Ho == Hv && a->Wo == Wv &&
               a->C1 % 8 == 0 && a->x_dtype == CRG_BF16 &&
               ((a->y_dtype == CRG_BF16 && a->prec == CRG_PREC_BF16) || (a->y_dtype == CRG_F32 && (a->prec == CRG_PREC_BF16X3 || a->prec == CRG_PREC_F16MX) && a->x_lo)))
                  ? 1 : 0;
  p.halo_lin = (a->Wo >= 16 && (a->Wo <= 128 ? 128 % a->Wo == 0 : a->Wo % 128 == 0)) ? 0 : 1;  // geometry of the output grid
  {
    const double ab = (double)a->N * a->H * a->W * a->C1 * 2.0, xb = (double)a->N * a->H * a->W * a->C2 * 2.0, wb = (double)p.N * p.K * 2.0;
    const double lim = 2147483648.0;
    p.a_bytes = ab < lim ? (unsigned)ab : 0;
    p.x2_bytes = xb < lim ? (unsigned)xb : 0;
    p.w_bytes = wb < lim ? (unsigned)wb : 0;
  }
  if (a->gn_y) {
    CRG_REQUIRE(ctx, a->gn_gamma && a->gn_beta && a->gn_groups > 0 && a->Cout % a->gn_groups == 0 && a->y,
                "conv2d: the fused GroupNorm needs gamma, beta, a group count that divides Cout=%d and the raw output y", a->Cout);
    CRG_REQUIRE(ctx, (((uintptr_t)a->gn_gamma | (uintptr_t)a->gn_beta | (uintptr_t)a->gn_y) & 15) == 0, "conv2d: GroupNorm pointers must be 16-byte aligned");
    p.gn_gamma = a->gn_gamma; p.gn_beta = a->gn_beta; p.gn_eps = a->gn_eps; p.gn_groups = a->gn_groups; p.gn_silu = a->gn_silu;
    p.gn_hw = a->Ho * a->Wo; p.gn_y = a->gn_y;
  }
  const double flops = 2.0 * p.M * (double)p.N * p.K;
  const double bytes = (double)a->N * a->H * a->W * Ctot * crg_dtype_size(a->x_dtype) + (double)p.N * p.K * 2 +
                       (double)p.M * p.N * crg_dtype_size(a->y_dtype) * (a->residual ? 2 : 1);
  ctx->gn_fused = false;
  const int rc = dispatch<true>(ctx, (hipStream_t)stream, p, 1, a->x_dtype, a->y_dtype, a->prec,
                                Work{flops, bytes, (double)a->N * a->H * a->W * Ctot * crg_dtype_size(a->x_dtype), (double)p.N * p.K * 2});
  if (a->gn_stats_rows) *a->gn_stats_rows = p.gstat_rows;
  if (rc || !a->gn_y || ctx->gn_fused) return rc;
  // not split along K (or a shape the fused kernel does not take): the GroupNorm runs as its own launch(es) on the finished y
  return crg_groupnorm(ctx, stream, a->y, nullptr, a->Cout, a->gn_gamma, a->gn_beta, a->gn_y, a->N, a->Ho * a->Wo, a->Cout, a->gn_groups,
                       a->gn_eps, a->gn_silu, a->y_dtype);
}
