// Persistent 256-row GEMM with a 3-slot LDS-DMA ring (gfx950), bf16 in / out:  Y[m][n] = epi( sum_k A[m][k] W[n][k] + bias[n] ) (+ residual)
//
// Why it exists (round 3).  The 128 x 128 / 128 x 160 tiles of gemm_glds_kernel stage (128 + BN) x 128 B per 64-deep k-tile for
// 2 x 128 x BN x 64 FLOP: 64 / 71 FLOP per staged byte.  A CU takes in about 50 GB/s through the LDS-DMA path when every CU streams
// (measured, DESIGN 4: "the 128 x 160 tile moves 12.8 TB/s L2 -> LDS"), i.e. those tiles are capped near 0.8 - 0.9 PFLOP/s by operand
// delivery however well the loop is scheduled - and the short-K token GEMMs of the UNet (K = 640 / 1280: 10 / 20 k-tiles behind a
// ~2 us first-DMA latency and a 2 - 3 us epilogue per tile) sit at 0.55 - 0.75.  The 3x3 convs reach 0.9 - 1.2 because the row-halo
// staging gives them the operand economy of a much larger tile.  This kernel gives the GEMMs the same economy directly:
//   * tile 256 x BN (BN = 160, or 128 for the GEGLU pairing) on 8 waves (4 x 2, wave tile 64 x BN/2): 98 / 85 FLOP per staged byte;
//   * one block per CU walks SEVERAL tiles; the ring never drains at a tile boundary - the first two k-tiles of the next tile are
//     requested while the current tile's last k-tiles are multiplied, and they land under its epilogue;
//   * ring of three slots (A 32 KB + W 16 / 20 KB each, 144 / 156 KB): k-tile u + 2 is requested in k-tile u, behind its first MFMA
//     block (its slot held k-tile u - 1, which every wave finished reading before the barrier of k-tile u) and has about one and a
//     half k-tile periods to land; waits are counted (s_waitcnt vmcnt(n)), one raw s_barrier per k-tile, nothing in the loop drains to zero;
//   * every piece goes through a buffer descriptor (rows >= M, columns >= N: zeros from the range check), the epilogue stores too
//     (out-of-range lanes are dropped), so that the number of vector-memory instructions a wave issues per tile is a compile-time
//     constant - that is what lets the first two waits of a tile leave the previous tile's stores in flight.
// Same MFMA layout as the other kernels (weights = A operand: a lane owns 4 (paired: 8) consecutive output channels of one row).
// Launched from gemm_conv.hip's `launch` where it measured faster than the two-blocks-per-CU kernel (see ring_gemm_ok below).
#include "gemm_shared.h"
#include <stdlib.h>

namespace crg_mm {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ void rg_dma16(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
#endif
}
static __device__ __forceinline__ void rg_store16(void* base, unsigned bytes, int voff, const bf16x8& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  u32x4 r;
  __builtin_memcpy(&r, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(r, rs, voff, 0, 0);
#endif
}
static __device__ __forceinline__ void rg_store8(void* base, unsigned bytes, int voff, const bf16x4& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  u32x2 r;
  __builtin_memcpy(&r, &v, 8);
#ifdef CRG_RG_ABL_NOSTORE  // timing ablation only: the store goes out of range (dropped by the range check, still issued and counted)
  voff = (int)0x80000000;
#endif
  __builtin_amdgcn_raw_buffer_store_b64(r, rs, voff, 0, 0);
#endif
}
static __device__ __forceinline__ bf16x8 rg_load16(const void* base, unsigned bytes, int voff) {
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
  __builtin_memcpy(&v, &r, 16);
#endif
  return v;
}
static __device__ __forceinline__ bf16x4 rg_load8(const void* base, unsigned bytes, int voff) {
  bf16x4 v = {0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
  __builtin_memcpy(&v, &r, 8);
#endif
  return v;
}
static __device__ __forceinline__ void rg_wait(int n) {  // wave-uniform s_waitcnt vmcnt(n)
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    case 12: wait_vmcnt<12>(); break;
    case 13: wait_vmcnt<13>(); break;
    case 14: wait_vmcnt<14>(); break;
    case 15: wait_vmcnt<15>(); break;
    case 16: wait_vmcnt<16>(); break;
    case 17: wait_vmcnt<17>(); break;
    case 18: wait_vmcnt<18>(); break;
    case 19: wait_vmcnt<19>(); break;
    case 20: wait_vmcnt<20>(); break;
    case 21: wait_vmcnt<21>(); break;
    case 22: wait_vmcnt<22>(); break;
    case 23: wait_vmcnt<23>(); break;
    case 24: wait_vmcnt<24>(); break;
    case 25: wait_vmcnt<25>(); break;
    case 26: wait_vmcnt<26>(); break;
    case 27: wait_vmcnt<27>(); break;
    case 28: wait_vmcnt<28>(); break;
    case 29: wait_vmcnt<29>(); break;
    case 30: wait_vmcnt<30>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

struct RingP {
  const bf16* a; long lda; unsigned a_bytes;
  const bf16* w; long ldw; unsigned w_bytes;   // packed [N][K] (GEGLU: value / gate rows interleaved in 16-row groups)
  const float* bias;                           // fp32 [N] (GEGLU: packed like the rows) or null
  const bf16* res; long ldr; unsigned res_bytes;
  bf16* y; long ldy; unsigned y_bytes;
  int M, N, K, nk;
  int tiles_m, tiles_n, tile_count;
  int xg_m, xg_n;                              // XCD partition of the (m-tile, n-tile) grid (product 8) or 0: contiguous runs
  int st_m, st_n;                              // supertile of the blocks co-resident on one XCD (st_m * st_n = grid / 8), see rg_tile; 0 = n-fastest order
  // LayerNorm as an epilogue correction (GemmP::ln_stat): y = rstd_m * (acc - mean_m * ln_s[n]) + bias'[n] on the raw rows
  const float* ln_stat; int ln_parts; const float* ln_s; float ln_eps;
};

// linear tile id -> (tile_m, tile_n): ids that are equal mod 8 run on one XCD (blocks are dealt round-robin and a block's ids are
// b, b + G, b + 2G, ... with G % 8 == 0), n fastest inside an XCD's share so that co-resident blocks share A rows / W panels
static __device__ __forceinline__ void rg_tile(const RingP& p, int L, int& tile_m, int& tile_n) {
  if (p.xg_m) {
    const int xcd = L & 7, idx = L >> 3;
    const int xn = xcd % p.xg_n, xm = xcd / p.xg_n;
    const int nnl = p.tiles_n / p.xg_n, nml = p.tiles_m / p.xg_m;
    int tn, tm;
    if (p.st_m) {
      // The grid / 8 blocks an XCD hosts at a time work on ONE st_m x st_n supertile (m fastest inside it), and the supertiles follow
      // each other along n: the st_m row blocks of A stay in the XCD's L2 for the whole n sweep and every weight panel is fetched once
      // per supertile row.  (n-fastest over the whole share - round 3 - kept only two row blocks in flight and walked ALL the share's
      // weight panels for them, 20 tiles' worth of W per pair of row blocks, while the output stream evicted them: PMC FETCH_SIZE
      // 147 MB per launch against 17 MB of operands at 8192 x 5120 x 640.)
      const int per = p.st_m * p.st_n, q = idx / per, r = idx - q * per;
      const int nsn = nnl / p.st_n, ms = q / nsn, ns = q - ms * nsn;
      tm = ms * p.st_m + r % p.st_m;
      tn = ns * p.st_n + r / p.st_m;
    } else {
      tn = idx % nnl;
      tm = idx / nnl;
    }
    tile_n = xn * nnl + tn;
    tile_m = xm * nml + tm;
  } else {
    const int T = p.tile_count;
    const int q = T >> 3, r = T & 7, xcd = L & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    tile_n = bid % p.tiles_n;
    tile_m = bid / p.tiles_n;
  }
}

// STAG (default): the two waves of a SIMD run a k-tile apart - waves 4-7 multiply k-tile u - 1 from fragments they kept in registers
// across the barrier while waves 0-3 read k-tile u, and read k-tile u while waves 0-3 multiply it - and BOTH defer a tile's epilogue
// (bias, GELU, stores: as many vector-ALU cycles as the tile's MFMAs for GEGLU) to behind the next barrier, where it runs beside the
// partner's MFMAs instead of beside the partner's epilogue (MI355X_MICROARCH.md "Two waves per SIMD" item 9).  Same ring, same slots,
// same results bit for bit.
// LNE: the A rows are RAW LayerNorm inputs, W is W o gamma; every tile's epilogue folds the producer's row-statistics partials of its
// 256 rows (lane l of a wave: row l of the wave's 64, coalesced; the rows a lane owns come over by ds_bpermute) and corrects the
// accumulators before bias / GELU.  The loads sit in the epilogue like the bias loads (complete before the tile's stores go out, so the
// counted waits see the same number of instructions in flight).
template <int WNT, bool GEGLU, bool STAG = true, bool LNE = false>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(RingP p) {
  constexpr bool PAIR = !GEGLU;
  constexpr int WMT = 4, NW = 8, TM = 256, NSLOT = 3;
  constexpr int BN = 32 * WNT;
  constexpr int AS_BYTES = TM * 128;
  constexpr int WS_BYTES = BN * 128;
  constexpr int SLOT = AS_BYTES + WS_BYTES;
  constexpr int ARG = TM / 8, WRG = BN / 8;        // 1 KiB pieces per k-tile
  constexpr int AL = ARG / NW;                     // A pieces per wave (4)
  constexpr int WL = (WRG + NW - 1) / NW;          // W pieces per wave (rounded up; waves past WRG issue none)
  constexpr int OOB = (int)0x80000000;
  // stores per wave and tile (constant: out-of-range lanes are dropped by the range check, the instruction is still issued)
  constexpr int NST = GEGLU ? WMT * (WNT / 2) : WMT * (WNT / 2 + (WNT & 1));
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int rsub = lane >> 3;
  const int clog = (lane & 7) ^ rsub;
  const int frow = lane & 15, fq = lane >> 4;
  const int nWp = (WRG - wave + NW - 1) / NW;      // W pieces this wave issues per k-tile (wave-uniform)
  const int nload = AL + nWp;

  const int G = gridDim.x;
  const int my_tiles = (p.tile_count - (int)blockIdx.x + G - 1) / G;
  const int S_total = my_tiles * p.nk;

  // ---- issue side: the (tile, k-tile) whose pieces are requested next ----
  int is_i = 0, is_kt = 0;                         // tile ordinal of this block, k-tile inside it
  int a_vo[AL], w_vo[WL];
  auto set_issue_tile = [&](int i) {
    int tm, tn;
    rg_tile(p, (int)blockIdx.x + i * G, tm, tn);
#pragma unroll
    for (int q = 0; q < AL; ++q) {
      const int row = tm * TM + (wave + NW * q) * 8 + rsub;
      a_vo[q] = row < p.M ? (int)((long)row * p.lda * 2) + clog * 16 : OOB;
    }
#pragma unroll
    for (int q = 0; q < WL; ++q) {
      const int pos = (wave + NW * q) * 8 + rsub;
      const int n = tn * BN + (PAIR ? unpair_col<WNT>(pos) : pos);
      w_vo[q] = ((wave + NW * q) < WRG && n < p.N) ? (int)((long)n * p.ldw * 2) + clog * 16 : OOB;
    }
  };
  auto issue = [&](int s) {  // stage s of this block's sequence into slot s % 3
    char* as = smem + (s % NSLOT) * SLOT;
    char* ws = as + AS_BYTES;
    const int soff = is_kt * 128;
#pragma unroll
    for (int q = 0; q < AL; ++q) rg_dma16(p.a, p.a_bytes, as + (wave + NW * q) * 1024, a_vo[q], soff);
#pragma unroll
    for (int q = 0; q < WL; ++q)
      if ((wave + NW * q) < WRG) rg_dma16(p.w, p.w_bytes, ws + (wave + NW * q) * 1024, w_vo[q], soff);
    if (++is_kt == p.nk) {
      is_kt = 0;
      ++is_i;
      if (is_i < my_tiles) set_issue_tile(is_i);
    }
  };

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a slot (second k-step: ^ 64, the k-step is bit 2 of the XOR-swizzled 16-byte chunk index)
  int xoff[WMT];
#pragma unroll
  for (int j = 0; j < WMT; ++j) xoff[j] = lds_off(wm * 64 + j * 16 + frow, fq);
  const int wb0 = AS_BYTES + (wn * (16 * WNT) + frow) * 128 + ((fq ^ (frow & 7)) << 4);  // rows 16 i + frow share (row & 7): + i * 2048

  if (S_total > 0) {
    set_issue_tile(0);
    issue(0);
    if (S_total > 1) issue(1);
  }
  // ---- pieces of the loop ----
  bf16x8 xf0[WMT], wf0[WNT], xf1[WMT], wf1[WNT];
  auto reads = [&](int u) {
    const char* sl = smem + (u % NSLOT) * SLOT;
#pragma unroll
    for (int j = 0; j < WMT; ++j) xf0[j] = *reinterpret_cast<const bf16x8*>(sl + xoff[j]);
#pragma unroll
    for (int q = 0; q < WNT; ++q) wf0[q] = *reinterpret_cast<const bf16x8*>(sl + wb0 + q * 2048);
#pragma unroll
    for (int j = 0; j < WMT; ++j) xf1[j] = *reinterpret_cast<const bf16x8*>(sl + (xoff[j] ^ 64));
#pragma unroll
    for (int q = 0; q < WNT; ++q) wf1[q] = *reinterpret_cast<const bf16x8*>(sl + ((wb0 ^ 64) + q * 2048));
  };
  auto mma0 = [&]() {
#pragma unroll
    for (int q = 0; q < WNT; ++q)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[q][j] = CRG_MFMA_16x16x32(wf0[q], xf0[j], acc[q][j]);
  };
  auto mma1 = [&]() {
#pragma unroll
    for (int q = 0; q < WNT; ++q)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[q][j] = CRG_MFMA_16x16x32(wf1[q], xf1[j], acc[q][j]);
  };
  // LNE: (rstd, mean * rstd) of the rows this lane's accumulators belong to, for the row block ln_tm - kept across the block's tiles: the
  // partials come from another XCD's writes, i.e. from the fabric (1 - 2 us in front of an epilogue's first store)
  float rj[LNE ? WMT : 1], mj[LNE ? WMT : 1];
  int ln_tm = -1;
  // epilogue of this block's i-th tile (exactly NST store instructions per wave), then the accumulators start over
  auto epilogue = [&](int i) {
    int tile_m, tile_n;
    rg_tile(p, (int)blockIdx.x + i * G, tile_m, tile_n);
    const int m0 = tile_m * TM, n0 = tile_n * BN;
    const int nb = n0 + wn * (16 * WNT);
    if (LNE && tile_m != ln_tm) {  // (consecutive tiles of a block share their rows in both tile orders: one fetch per block, not per tile)
      ln_tm = tile_m;
      const int mr = m0 + wm * 64 + lane;
      const f32x2 ab = ln_fold_row(p.ln_stat, mr < p.M ? mr : p.M - 1, p.ln_parts);
      const float a = ab[0], b = ab[1];
      const float invk = 1.0f / (float)p.K;
      const float mean = a * invk;
      float var = __builtin_fmaf(-mean, mean, b * invk);
      var = var > 0.f ? var : 0.f;
      const float rstd = __builtin_amdgcn_rsqf(var + p.ln_eps);
      const float mrs = mean * rstd;
#pragma unroll
      for (int j = 0; j < WMT; ++j) {
        rj[j] = __shfl(rstd, j * 16 + frow);
        mj[j] = __shfl(mrs, j * 16 + frow);
      }
    }
    auto lncorr = [&](const f32x4& a4, int j, const f32x4& s4) -> f32x4 {
      if constexpr (LNE) return rj[j] * a4 - mj[j] * s4;
      else return a4;
    };
    auto lns = [&](int col, bool ok) -> f32x4 {
      if constexpr (LNE) return *reinterpret_cast<const f32x4*>(p.ln_s + (ok ? col : 0));
      else return f32x4{0.f, 0.f, 0.f, 0.f};
    };
    if constexpr (GEGLU) {
#pragma unroll
      for (int u2 = 0; u2 < WNT / 2; ++u2) {
        const int pn = nb + u2 * 32 + fq * 4;       // packed column of the value tile; its gate group sits 16 packed columns further
        const int jn = nb / 2 + u2 * 16 + fq * 4;   // output column
        const bool nok = pn + 20 <= p.N;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = bv;
        if (p.bias) {
          const int pc = nok ? pn : 0;
          bv = *reinterpret_cast<const f32x4*>(p.bias + pc);
          bg = *reinterpret_cast<const f32x4*>(p.bias + pc + 16);
        }
        const f32x4 sv = lns(pn, nok), sg = lns(pn + 16, nok);
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          const int m = m0 + wm * 64 + j * 16 + frow;
          const f32x4 v = lncorr(acc[2 * u2][j], j, sv) + bv, g = lncorr(acc[2 * u2 + 1][j], j, sg) + bg;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] * crg_gelu_erf_f(g[e]));
          rg_store8(p.y, p.y_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldy + jn) * 2) : OOB, o);
        }
      }
    } else {
#pragma unroll
      for (int u2 = 0; u2 < WNT / 2; ++u2) {
        const int n = nb + 32 * u2 + 8 * fq;        // paired mapping: this lane's 8 consecutive columns of tiles (2u, 2u + 1)
        const bool nok = n + 8 <= p.N;
        f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bb = ba;
        if (p.bias) {
          const int pc = nok ? n : 0;
          ba = *reinterpret_cast<const f32x4*>(p.bias + pc);
          bb = *reinterpret_cast<const f32x4*>(p.bias + pc + 4);
        }
        const f32x4 sa = lns(n, nok), sb = lns(n + 4, nok);
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          const int m = m0 + wm * 64 + j * 16 + frow;
          f32x4 a4 = lncorr(acc[2 * u2][j], j, sa) + ba, b4 = lncorr(acc[2 * u2 + 1][j], j, sb) + bb;
          if (p.res) {
            const bf16x8 r8 = rg_load16(p.res, p.res_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldr + n) * 2) : OOB);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a4[e] += (float)r8[e];
              b4[e] += (float)r8[4 + e];
            }
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (bf16)a4[e];
            o[4 + e] = (bf16)b4[e];
          }
          rg_store16(p.y, p.y_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldy + n) * 2) : OOB, o);
        }
      }
      if constexpr (WNT & 1) {
        const int n = nb + 16 * (WNT - 1) + 4 * fq;
        const bool nok = n + 4 <= p.N;
        f32x4 ba = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) ba = *reinterpret_cast<const f32x4*>(p.bias + (nok ? n : 0));
        const f32x4 sa = lns(n, nok);
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          const int m = m0 + wm * 64 + j * 16 + frow;
          f32x4 a4 = lncorr(acc[WNT - 1][j], j, sa) + ba;
          if (p.res) {
            const bf16x4 r4 = rg_load8(p.res, p.res_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldr + n) * 2) : OOB);
#pragma unroll
            for (int e = 0; e < 4; ++e) a4[e] += (float)r4[e];
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)a4[e];
          rg_store8(p.y, p.y_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldy + n) * 2) : OOB, o);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < WNT; ++q)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto top = [&](int u, int& ttl) {  // k-tile u landed (the batch requested after it and, behind an epilogue, its stores stay in flight); barrier
    rg_wait((u + 1 < S_total ? nload : 0) + (ttl > 0 ? NST : 0));
    if (ttl > 0) --ttl;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  if constexpr (!STAG) {
    int u = 0, ttl = 0;
    for (int i = 0; i < my_tiles; ++i) {
      for (int kt = 0; kt < p.nk; ++kt, ++u) {
        top(u, ttl);
        reads(u);
        __builtin_amdgcn_sched_barrier(0);
        mma0();
        __builtin_amdgcn_sched_barrier(0);
        // the DMA batch of k-tile u + 2 goes out BEHIND the first MFMA block; its slot held k-tile u - 1, which every wave finished
        // reading before this k-tile's barrier
        if (u + 2 < S_total) {
          __builtin_amdgcn_s_setprio(2);
          issue(u + 2);
          __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma1();
      }
      epilogue(i);  // the next tile's first k-tiles are landing underneath
      ttl = 2;
    }
  } else if (wave < 4) {
    // waves 0-3: barrier | epilogue of the previous tile (first k-tile of a tile) | read u | multiply u (issue u + 2 in between)
    int u = 0, ttl = 0;
    for (int i = 0; i < my_tiles; ++i) {
      for (int kt = 0; kt < p.nk; ++kt, ++u) {
        top(u, ttl);
        if (kt == 0 && i > 0) {
          epilogue(i - 1);  // its stores precede the batch of k-tile u + 2: only the next wait has to leave them in flight
          ttl = 1;
        }
        reads(u);
        __builtin_amdgcn_sched_barrier(0);
        mma0();
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < S_total) issue(u + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma1();
      }
    }
    if (my_tiles > 0) epilogue(my_tiles - 1);
  } else {
    // waves 4-7: barrier | multiply u - 1 (fragments in registers) | issue u + 2 | epilogue if u - 1 closed a tile | read u
    int u = 0, ttl = 0;
    for (int i = 0; i < my_tiles; ++i) {
      for (int kt = 0; kt < p.nk; ++kt, ++u) {
        top(u, ttl);
        if (u > 0) {
          mma0();
          mma1();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < S_total) issue(u + 2);
        if (kt == 0 && i > 0) {
          epilogue(i - 1);  // behind the batch of k-tile u + 2: the next TWO waits leave the stores in flight
          ttl = 2;
        }
        __builtin_amdgcn_sched_barrier(0);
        reads(u);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the reads have returned before the next barrier lets the slot be refilled
      }
    }
    if (my_tiles > 0) {
      mma0();
      mma1();
      epilogue(my_tiles - 1);
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
// Shapes the ring kernel takes in place of gemm_glds_kernel: bf16 in / out, one problem (no batch), K a multiple of 64, plain or
// GEGLU epilogue (no SiLU, no per-row bias, no timestep vector, no GroupNorm statistics), every extent addressable through a 32-bit
// buffer descriptor - and, by measurement (tools/lin_probe.py / lin_probe_xl.py, device time inside a captured graph, round 3):
//   GEGLU epilogue, >= 3 tiles per CU : 8192 x 5120 x 640   82 -> 71 us,  16384 x 5120 x 640  164 -> 145 us,  4096 x 10240 x 1280  139 -> 127 us
//   GEGLU, 2.5 tiles per CU           : 2048 x 10240 x 1280  69 -> 69 us (equal)
//   plain epilogue (any tile count)   : 32768 x 320 x 320 + residual 18.6 -> 21.9 us, 8192 x 1920 x 640 30.5 -> 34.4 us, 32768 x 320 x 1280 36 -> 40 us,
//                                       16384 x 1920 x 640 58.5 -> 57 us: the two-blocks-per-CU kernel hides its short epilogue behind the
//                                       other block's MFMAs already; this kernel's 52 one-KiB DMA pieces per k-tile (the row-halo conv needs
//                                       32) then cost more issue time than the larger tile saves
// so it is routed for GEGLU GEMMs with at least three 256-row tiles per CU.  CRG_GEMM_RING: 0 = never, 1 (default) = that rule,
// 2 = plain epilogues as well (from CRG_GEMM_RING_MIN percent of a tile per CU, default 300).
bool ring_gemm_ok(const GemmP& p, int batch, int n_cu) {
  static const int on = getenv("CRG_GEMM_RING") ? atoi(getenv("CRG_GEMM_RING")) : 1;
  static const int min_pct = getenv("CRG_GEMM_RING_MIN") ? atoi(getenv("CRG_GEMM_RING_MIN")) : 300;  // dev knob: least tiles, % of the CU count
  if (!on || batch != 1 || p.K % 64 || p.K < 128 || p.splits != 1) return false;
  if (p.epi != CRG_EPI_GEGLU && !(on >= 2 && p.epi == CRG_EPI_NONE)) return false;
  if (p.bias_mode == CRG_BIAS_ROW || p.cvec || p.gstat || p.vt || p.rstat) return false;
  if (p.ln_stat && (p.res || ((uintptr_t)p.ln_s & 15))) return false;
  const bool geglu = p.epi == CRG_EPI_GEGLU;
  if (geglu ? (p.N % 32 || (p.ldy & 3)) : (p.N % 8 || (p.ldy & 7) || ((uintptr_t)p.y & 15))) return false;
  if (p.res && ((p.ldr & 7) || ((uintptr_t)p.res & 15))) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  const double lim = 2147483648.0;
  if ((double)p.M * p.lda * 2 >= lim || (double)p.N * p.ldw * 2 >= lim || (double)p.M * p.ldy * 2 >= lim || (p.res && (double)p.M * p.ldr * 2 >= lim)) return false;
  const int bn = (!geglu && p.N % 160 == 0) ? 160 : 128;
  const long tiles = (long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn);
  return tiles * 100 >= (long)(n_cu > 0 ? n_cu : 256) * min_pct;
}

int launch_gemm_ring(crg_ctx* ctx, hipStream_t st, const GemmP& g, double flops, double bytes) {
  const bool geglu = g.epi == CRG_EPI_GEGLU;
  const int wnt = (!geglu && g.N % 160 == 0) ? 5 : 4;
  const int bn = 32 * wnt;
  RingP p{};
  p.a = (const bf16*)g.a; p.lda = g.lda; p.a_bytes = (unsigned)((double)g.M * g.lda * 2);
  p.w = g.w; p.ldw = g.ldw; p.w_bytes = (unsigned)((double)g.N * g.ldw * 2);
  p.bias = g.bias_mode == CRG_BIAS_COL ? g.bias : nullptr;
  p.res = (const bf16*)g.res; p.ldr = g.ldr; p.res_bytes = g.res ? (unsigned)((double)g.M * g.ldr * 2) : 0;
  p.y = (bf16*)g.y; p.ldy = g.ldy; p.y_bytes = (unsigned)((double)g.M * g.ldy * 2);
  p.M = g.M; p.N = g.N; p.K = g.K; p.nk = g.K / 64;
  p.ln_stat = g.ln_stat; p.ln_parts = g.ln_parts; p.ln_s = g.ln_s; p.ln_eps = g.ln_eps;
  p.tiles_m = (g.M + 255) / 256;
  p.tiles_n = (g.N + bn - 1) / bn;
  p.tile_count = p.tiles_m * p.tiles_n;
  // XCD partition minimising xg_n * |A| + xg_m * |W| over the factorizations of 8 that divide the tile grid (else contiguous runs)
  p.xg_m = p.xg_n = 0;
  {
    const double ab = (double)g.M * g.K * 2, wb = (double)g.N * g.K * 2;
    double best = 0.0;
    for (int gn = 1; gn <= 8; gn <<= 1) {
      const int gm = 8 / gn;
      if (p.tiles_n % gn || p.tiles_m % gm) continue;
      const double cost = gn * ab + gm * wb;
      if (!p.xg_m || cost < best) {
        best = cost;
        p.xg_m = gm; p.xg_n = gn;
      }
    }
  }
  int grid = ctx->n_cu > 0 ? ctx->n_cu : 256;
  grid &= ~7;  // a block's tile ids must stay on one XCD label
  if (grid > p.tile_count) grid = p.tile_count;
  p.st_m = p.st_n = 0;
  {
    static const int super = getenv("CRG_GEMM_RING_SUPER") ? atoi(getenv("CRG_GEMM_RING_SUPER")) : 1;  // dev knob (A/B): 0 = n-fastest tile order inside an XCD's share
    const int per = grid / 8;  // blocks of one XCD, i.e. tiles it works on at a time
    if (super && p.xg_m && grid % 8 == 0 && per > 0) {
      const int nnl = p.tiles_n / p.xg_n, nml = p.tiles_m / p.xg_m;
      // ... where the XCD's whole row share is ONE supertile row (its A rows then stay in L2 for the entire launch).  Measured, device time in
      // a graph + PMC (tools/ring_pmc.sh, round 4): 8192 x 5120 x 640 FETCH_SIZE 147 -> 87 MB per launch, 86.7 -> 82.4 us; 4096 x 10240 x 1280
      // 144.3 -> 141.5 us; with two supertile rows per XCD (16384 x 5120 x 640) 158.7 -> 161.2 us, with sixteen (K = 320) +4 %: those keep
      // the n-fastest order
      for (int sm = 8; sm >= 1; sm >>= 1) {
        if (per % sm || nml != sm) continue;
        const int sn = per / sm;
        if (nnl % sn) continue;
        p.st_m = sm; p.st_n = sn;
        break;
      }
    }
  }
  static const int stag = getenv("CRG_GEMM_RING_STAG") ? atoi(getenv("CRG_GEMM_RING_STAG")) : 1;  // dev knob: 0 = all eight waves in lockstep (round 3a)
  void (*kern)(RingP) = stag ? (geglu ? gemm_ring_kernel<4, true> : (wnt == 5 ? gemm_ring_kernel<5, false> : gemm_ring_kernel<4, false>))
                             : (geglu ? gemm_ring_kernel<4, true, false> : (wnt == 5 ? gemm_ring_kernel<5, false, false> : gemm_ring_kernel<4, false, false>));
  if (g.ln_stat)  // LayerNorm epilogue: staggered schedule only
    kern = geglu ? gemm_ring_kernel<4, true, true, true> : (wnt == 5 ? gemm_ring_kernel<5, false, true, true> : gemm_ring_kernel<4, false, true, true>);
  const size_t lds = (size_t)3 * (256 * 128 + bn * 128);
  if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), 160 * 1024, "gemm ring")) return rc;
  crg_prof_scope ps(ctx, st, wnt == 5 ? CRG_K_GEMM_W5 : CRG_K_GEMM_W4, flops, bytes);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p);
  CRG_CHECK_LAUNCH(ctx, "gemm_ring");
  return 0;
}

}  // namespace crg_mm
