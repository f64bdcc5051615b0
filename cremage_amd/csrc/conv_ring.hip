// 3x3 / stride 1 / pad 1 conv on 256-pixel tiles with a DEEP LDS-DMA ring (gfx950).
//
// Same math, operand layout and row-halo staging as conv3_rowhalo_kernel<.., MT = 2> (gemm_conv.hip): tile = 256 output
// pixels x BN output channels on 8 waves (4 x 2, each 64 x 16*WNT), K in (64-channel chunk, kernel row) GROUPS of three
// k-tiles (kw = 0..2) that share ONE row buffer of R x (W + 2) pixels; weights [Cout][chunk-major K].
//
// What is different is the pipeline.  The 2-stage kernel issues the weight k-tile t + 1 at the top of k-tile t and drains
// vmcnt to 0 before every k-tile: with one block per CU its k-tile period is the global -> LDS round trip (~3000 cycles
// measured) although the MFMAs of a k-tile take 1280.  Here
//   * the weight ring has FOUR slots: W(t + 3) is issued in k-tile t and only has to be in LDS at the top of k-tile t + 2;
//   * the row buffer of group g + 1 is issued whole in the first k-tile of group g (its slot was last read in group g - 1)
//     and has to be there at the top of the LAST k-tile of group g;
//   * every wait is a counted s_waitcnt vmcnt(n) that leaves the newest issue batch in flight (n = that batch's size for
//     this wave), followed by ONE raw s_barrier per k-tile; nothing in the loop drains to 0;
//   * because everything a k-tile reads is visible one barrier EARLY, the A / W fragments of the first 32-wide k-step of
//     k-tile t + 1 are read during the second k-step's MFMAs of k-tile t: the matrix pipe does not drain at the barrier.
// LDS: 4 x BN x 128 B (80 KB at BN = 160) + 2 x <= 36 KB row buffers = 152 KB: one block per CU, two waves per SIMD.
//
// Split-K (K slices in groups, 16x16 / 8x8 levels): each slice writes its fp32 accumulators as a register image
// slab[tile][slice][wave][i][j][lane] (f32x4 per lane: 1 KiB coalesced wave-stores), then takes a ticket on the tile's
// arrival counter (release fence before, guide 5 "In-launch split-K reduction"); the block that draws the last ticket
// sums the slices IN SLICE ORDER (its own from registers, the others from their slabs behind an acquire fence) - the
// result does not depend on which block came last - and runs the normal epilogue.  No second launch, no float atomics.
#include "gemm_shared.h"
#include <type_traits>

namespace crg_mm {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

static __device__ __forceinline__ void wait_vm_n(int n) {  // wave-uniform n: s_waitcnt vmcnt(n), n <= 10
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
    default: wait_vmcnt<10>(); break;
  }
}

// One LDS-DMA piece through a raw buffer descriptor: lane -> 16 bytes at base + voff + soff, zeros when out of [0, bytes).
// (Device pass only: used inside a lambda, the builtin makes the HOST pass drop the kernel's stub without a diagnostic.)
static __device__ __forceinline__ void dma16(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
#endif
}

#ifdef CRG_RING_STAMPS
// dev instrumentation (tools/ring_stamp_probe.py; build with tools/build_variant.sh stamps -DCRG_RING_STAMPS): wall_clock64 stamps of
// wave 0 of every block: 0 entry, 1 first operands landed (prologue barrier passed), 2 K loop done, 3 epilogue stores issued
__device__ unsigned long long crg_ring_stamps[1024 * 4];
#define RING_STAMP(i) do { if (t == 0 && blockIdx.x < 1024) crg_ring_stamps[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#else
#define RING_STAMP(i)
#endif

template <int WNT, bool PAIR, int SPREAD, bool LIN>
__global__ __launch_bounds__(512, 2) void conv3_ring_kernel(GemmP p) {
  constexpr int WMT = 4, NW = 8, TP = 256, WST = 4;
  constexpr int BN = 32 * WNT;
  constexpr int WS_BYTES = BN * 128;
  constexpr int WRG = BN / 8;
  constexpr int WL = (WRG + NW - 1) / NW;
  constexpr int XI = 5;  // row-buffer pieces (8 pixels = 1 KiB) per wave and group: <= 36 pieces (TP + 2 R pixels, W >= 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  RING_STAMP(0);
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Wd = p.Wo, Hd = p.Ho, sh = p.up ? 1 : 0;
  constexpr bool lin = LIN;  // compile time: a runtime test put a branch (and a conservative lgkmcnt(0)) in front of every MFMA block
  const bool seg = lin || Wd > TP;
  const int WP = seg ? TP + 2 : Wd + 2;
  const int R = seg ? 1 : TP / Wd;
  const int xpix = R * WP;
  const int XP = (xpix + 7) >> 3;
  const int xbuf_bytes = XP * 1024;
  char* const wring = smem;                   // [slot][BN x 128 B]
  char* const xbuf = smem + WST * WS_BYTES;   // [parity][xbuf_bytes]

  int tile_m, tile_n, sid;
  block_to_tile(p, tile_m, tile_n, sid);
  const int m0 = tile_m * TP, n0 = tile_n * BN;
  const int rsub = lane >> 3;
  const int clog = (lane & 7) ^ rsub;

  // number of DMA pieces this wave issues per weight k-tile / per row buffer (wave-uniform)
  const int nW = (WRG - wave + NW - 1) / NW;
  const int nX = (XP - wave + NW - 1) / NW;

  // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): a lane's source is a 32-bit byte offset, the
  // 64-channel chunk / kernel row / k-tile term is a SCALAR offset, and a lane whose offset is out of range gets ZEROS written to
  // its LDS slot by the range check (tools/probes/oob_lds_dma.hip) - image borders, tile tails and rows >= N need no zero page,
  // no 64-bit address arithmetic and no select.  xp0 = pixel index of the kernel-row-0 source (24 bits), xmask bits 0-2 =
  // kernel row kh readable, bit 3 (nearest-2x upsample only) = source row of kh = 1 differs from that of kh = 0.
  // (the descriptors are built where they are used: a lambda cannot capture the descriptor type on the host pass)
  constexpr int OOB = (int)0x80000000;
  int xp0[XI];
  unsigned xmask[XI];
  {
    const int rows_total = p.M / Wd;
    const int row0 = m0 / Wd;
    const int w0 = m0 - row0 * Wd;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int b = 8 * (wave + NW * i) + rsub;
      int grow, w;
      bool ok;
      if (lin) {
        const int pb = m0 - 1 + b;
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
      if (sh && ((h >> 1) != ((h - 1) >> 1))) msk |= 8u;
      xmask[i] = msk;
      xp0[i] = img * p.H * p.W + (w >> sh) + ((h - 1) >> sh) * p.W;  // may be "negative" for h = 0: only used when bit 0 is set
    }
  }
  int wvo[WL];  // byte offset of this lane's 16 bytes in k-tile 0 of its weight row
#pragma unroll
  for (int q = 0; q < WL; ++q) {
    const int pos = (wave + NW * q) * 8 + rsub;
    const int n = n0 + (PAIR ? unpair_col<WNT>(pos) : pos);
    wvo[q] = (n < p.N && (wave + NW * q) < WRG) ? (int)((long)n * p.ldw * 2) + clog * 16 : OOB;
  }
  const int g_begin = sid * p.ks_q + (sid < p.ks_r ? sid : p.ks_r);
  const int g_end = g_begin + p.ks_q + (sid < p.ks_r ? 1 : 0);
  const int NT = 3 * (g_end - g_begin);

  auto stage_x = [&](int g) {  // the whole row buffer of group g
    const int c = g / 3, kh = g - 3 * c;
    const int cch = c * 64;
    const bool second = cch >= p.C1;
    const int Cs2 = 2 * (second ? p.C2 : p.C1);
    const int soff = 2 * (second ? cch - p.C1 : cch);
    // source row of kernel row kh relative to that of kernel row 0: kh rows, or (upsample) 0 / bit 3 / 1
    const int drow = sh ? (kh == 2 ? p.W : 0) : kh * p.W;
    const unsigned bit = 1u << kh;
    char* xb = xbuf + ((g - g_begin) & 1) * xbuf_bytes;
    const void* xbase = second ? p.x2 : p.a;
    const unsigned xbytes = second ? p.x2_bytes : p.a_bytes;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int jp = wave + NW * i;
      if (jp < XP) {  // wave-uniform
        int px = xp0[i] + drow;
        if (sh && kh == 1) px += (xmask[i] & 8u) ? p.W : 0;
        const int vo = (xmask[i] & bit) ? px * Cs2 + clog * 16 : OOB;
        dma16(xbase, xbytes, xb + jp * 1024, vo, soff);
      }
    }
  };
  auto stage_w = [&](int kt, int slot) {  // weight k-tile kt (absolute index: 64 K-elements) into ring slot
    char* ws = wring + slot * WS_BYTES;
#pragma unroll
    for (int q = 0; q < WL; ++q) {
      if ((wave + NW * q) < WRG)  // wave-uniform
        dma16(p.w, p.w_bytes, ws + (wave + NW * q) * 1024, wvo[q], kt * 128);
    }
  };

  // piece-granular forms of the two stagers (SPREAD == 3 threads single pieces through the MFMA groups)
  struct XGroup { const void* base; unsigned bytes; int Cs2, soff, drow; unsigned bit; char* xb; bool mid; };
  auto x_group = [&](int g) {
    const int c = g / 3, kh = g - 3 * c;
    const int cch = c * 64;
    const bool second = cch >= p.C1;
    XGroup r;
    r.base = second ? p.x2 : p.a;
    r.bytes = second ? p.x2_bytes : p.a_bytes;
    r.Cs2 = 2 * (second ? p.C2 : p.C1);
    r.soff = 2 * (second ? cch - p.C1 : cch);
    r.drow = sh ? (kh == 2 ? p.W : 0) : kh * p.W;
    r.bit = 1u << kh;
    r.xb = xbuf + ((g - g_begin) & 1) * xbuf_bytes;
    r.mid = sh && kh == 1;
    return r;
  };
  auto x_piece = [&](int i, const XGroup& r) {
    const int jp = wave + NW * i;
    if (jp < XP) {  // wave-uniform
      int px = xp0[i] + r.drow;
      if (r.mid) px += (xmask[i] & 8u) ? p.W : 0;
      const int vo = (xmask[i] & r.bit) ? px * r.Cs2 + clog * 16 : OOB;
      dma16(r.base, r.bytes, r.xb + jp * 1024, vo, r.soff);
    }
  };
  auto w_piece = [&](int q, int kt, int slot) {
    if ((wave + NW * q) < WRG) dma16(p.w, p.w_bytes, wring + slot * WS_BYTES + (wave + NW * q) * 1024, wvo[q], kt * 128);
  };

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fq = lane >> 4;
  int xb0[WMT];
#pragma unroll
  for (int j = 0; j < WMT; ++j) {
    const int ml = wm * 64 + j * 16 + frow;
    const int r = seg ? 0 : ml / Wd;
    xb0[j] = r * WP + (ml - r * Wd);
  }
  unsigned edge = 0;
  if (lin) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const int m = m0 + wm * 64 + j * 16 + frow;
      const int wcol = m % Wd;
      edge |= (wcol == 0 ? 1u : 0u) << j;
      edge |= (wcol == Wd - 1 ? 1u : 0u) << (8 + j);
    }
  }

  // fragment reads.  The byte offsets are NOT left to the compiler to hoist: the 24 (j, kw, ks) activation offsets plus the
  // weight offsets cost ~45 loop-invariant VGPRs, and with 152 registers of accumulators and two fragment sets the loop
  // spilled (scratch traffic counts in vmcnt: it would drain the DMA ring).  xoff[j] = offset of the CURRENT k-tile's first
  // k-step fragment, recomputed (5 VALU) when the next k-tile's fragments are requested; the second k-step is xoff ^ 64
  // (k-step = bit 2 of the 16-byte chunk index, XOR-swizzled).
  int xoff[WMT];
  auto set_xoff = [&](int kw) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      int u = xb0[j];
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass rejects the "v" constraint and then silently drops the kernel's stub)
      asm volatile("" : "+v"(u));  // opaque: keeps this arithmetic inside the loop
#endif
      u += kw;
      xoff[j] = (u << 7) + (((u & 7) ^ fq) << 4);
    }
  };
  auto read_x = [&](bf16x8 (&xf)[WMT], const char* xs, int kw, int ks) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xs + (xoff[j] ^ (ks << 6)));
    if constexpr (LIN) if (kw != 1) {
#pragma unroll
      for (int j = 0; j < WMT; ++j)
        if ((edge >> (kw == 0 ? j : 8 + j)) & 1u) xf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  const int wb0 = (wn * (16 * WNT) + frow) * 128 + ((fq ^ (frow & 7)) << 4);  // rows 16 i + frow share (row & 7): + i * 2048
  auto read_w = [&](bf16x8 (&wf)[WNT], const char* ws, int ks) {
    const char* base = ws + (wb0 ^ (ks << 6));
#pragma unroll
    for (int i = 0; i < WNT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
  };
  auto mma = [&](const bf16x8 (&xf)[WMT], const bf16x8 (&wf)[WNT]) {
#ifdef CRG_ABL_NOMMA  // timing-only ablation: keep the fragments live, issue no MFMA
#pragma unroll
    for (int i = 0; i < WNT; ++i) asm volatile("" ::"v"(wf[i]));
#pragma unroll
    for (int j = 0; j < WMT; ++j) asm volatile("" ::"v"(xf[j]));
#else
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
#endif
  };

  auto mma_i = [&](const bf16x8 (&xf)[WMT], const bf16x8 (&wf)[WNT], int i) {  // the four MFMAs of weight fragment i
#ifdef CRG_ABL_NOMMA
    asm volatile("" ::"v"(wf[i]));
#pragma unroll
    for (int j = 0; j < WMT; ++j) asm volatile("" ::"v"(xf[j]));
#else
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
#endif
  };
  bf16x8 xf0[WMT], wf0[WNT], xf1[WMT], wf1[WNT];
  if (NT > 0) {
    // prologue: X(g0), W(0), W(1), W(2); first wait leaves W(1), W(2) in flight
    stage_x(g_begin);
    stage_w(3 * g_begin, 0);
    if (NT > 1) stage_w(3 * g_begin + 1, 1);
    if (NT > 2) stage_w(3 * g_begin + 2, 2);
    wait_vm_n(nW * (NT > 2 ? 2 : NT - 1));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RING_STAMP(1);
    set_xoff(0);
    read_x(xf0, xbuf, 0, 0);
    read_w(wf0, wring, 0);
  }
  // residual (16 bytes per tile pair, 8 for an odd last tile) and bias of this lane: NOT held across the K loop (that cost
  // 60 VGPRs and spilled); they are requested in the FINAL k-tile, behind its first MFMA block, when the first fragment set
  // is dead, and arrive under the last 20 MFMAs of both waves of the SIMD.
  bf16x4 rres[PAIR ? 1 : WNT][PAIR ? 1 : WMT];
  bf16x8 r2[PAIR ? (WNT / 2 > 0 ? WNT / 2 : 1) : 1][PAIR ? WMT : 1];
  bf16x4 r1[PAIR ? WMT : 1];
  f32x4 bpre[WNT];
  const bool whole = p.splits == 1 || p.inred;  // this launch writes the finished output (no reduce kernel behind it)
  const bool pre_res = p.res && whole && (p.ldr & 3) == 0 && (p.N & 3) == 0;
  const bool pre_bias = p.bias_mode == CRG_BIAS_COL && whole && (p.N & 3) == 0;
  auto fetch_res = [&]() {
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
  };

  // one k-tile: KW = tap column (compile time), FINAL = last k-tile of the slice (no issue, no prefetch; residual request)
  // STEADY = a k-tile of a group that is not the slice's last one: every issue below happens, so the waits are the two / three
  // compile-time counts selected by this wave's piece counts (two scalar branches instead of the generic switch)
  auto ktile = [&](int g, int tt, auto KWc, auto FINALc, auto STEADYc) {
    constexpr int kw = decltype(KWc)::value;
    constexpr bool FINAL = decltype(FINALc)::value;
    constexpr bool STEADY = decltype(STEADYc)::value;
    const char* xs = xbuf + ((g - g_begin) & 1) * xbuf_bytes;
    const bool next_g = STEADY || g + 1 < g_end;
    // top(tt): everything older than the batch issued in k-tile tt - 1 has landed
    if constexpr (STEADY) {
      if (nW == 3) {
        if constexpr (kw == 1) { if (nX == 5) wait_vmcnt<8>(); else wait_vmcnt<7>(); }
        else wait_vmcnt<3>();
      } else {
        if constexpr (kw == 1) { if (nX == 5) wait_vmcnt<7>(); else wait_vmcnt<6>(); }
        else wait_vmcnt<2>();
      }
    } else {
      int allow = (tt + 2 < NT) ? nW : 0;
      if (kw == 1 && next_g) allow += nX;
      wait_vm_n(allow);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* ws = wring + (tt & 3) * WS_BYTES;
    // Phase order pinned with sched_barrier: left alone, hipcc threads the ds_reads through the MFMAs and fences them with
    // s_waitcnt lgkmcnt(0) (the matrix pipe then waits out an LDS round trip four or five times per k-tile).  Pinned, every
    // MFMA block consumes fragments requested one block earlier and the waits are exact counts.
    // SPREAD: where this wave issues its DMA batch.  0 = at the top; 1 = behind the first MFMA block; 2 = waves 0-3 at the
    // top, their SIMD partners (waves 4-7) behind the first MFMA block, so that one wave of a SIMD multiplies while the
    // other issues (guide: microarch "Two waves per SIMD" item 9).
    auto issue = [&]() {
#ifndef CRG_ABL_NODMA
      if constexpr (!FINAL) {
        __builtin_amdgcn_s_setprio(2);
        if (kw == 0 && next_g) stage_x(g + 1);
        if (tt + 3 < NT) stage_w(3 * g_begin + tt + 3, (tt + 3) & 3);
        __builtin_amdgcn_s_setprio(0);
      }
#endif
    };
    if constexpr (SPREAD == 3) {
      // Fine interleave: every group of four MFMAs is followed by ONE other item - a fragment-read burst or one or two DMA
      // pieces - so that no wave ever sits in a long non-MFMA phase (in lockstep the two waves of a SIMD sat in theirs together
      // and the matrix pipe idled: ablation, 0.35 us of LDS / barrier work + 0.19 us of DMA issue per k-tile on top of the MFMAs).
#define SB __builtin_amdgcn_sched_barrier(0)
#ifndef CRG_ABL_NODMA
      const bool do_x = !FINAL && kw == 0 && next_g;
      const bool do_w = !FINAL && tt + 3 < NT;
#else
      const bool do_x = false, do_w = false;
#endif
      XGroup xg{};
      if (do_x) xg = x_group(g + 1);
      const int wkt = 3 * g_begin + tt + 3, wslot = (tt + 3) & 3;
      auto gap = [&](auto Kc) {  // DMA slot k: weight piece k (k < WL), then row-buffer piece k - WL
        constexpr int k = decltype(Kc)::value;
        if constexpr (k < WL) { if (do_w) w_piece(k, wkt, wslot); }
        else if constexpr (k - WL < XI) { if (do_x) x_piece(k - WL, xg); }
      };
      using std::integral_constant;
      mma_i(xf0, wf0, 0); SB;
      read_x(xf1, xs, kw, 1); SB;
      mma_i(xf0, wf0, 1); SB;
      read_w(wf1, ws, 1); SB;
      mma_i(xf0, wf0, 2); SB;
      gap(integral_constant<int, 0>{}); gap(integral_constant<int, 1>{}); SB;
      mma_i(xf0, wf0, 3); SB;
      gap(integral_constant<int, 2>{}); gap(integral_constant<int, 3>{}); SB;
      if constexpr (WNT > 4) { mma_i(xf0, wf0, 4); SB; }
      if constexpr (!FINAL) {
        const char* xs2 = (kw == 2) ? xbuf + ((g + 1 - g_begin) & 1) * xbuf_bytes : xs;
        set_xoff(kw == 2 ? 0 : kw + 1);
        read_x(xf0, xs2, kw == 2 ? 0 : kw + 1, 0);
      } else {
        fetch_res();
      }
      SB;
      mma_i(xf1, wf1, 0); SB;
      if constexpr (!FINAL) { read_w(wf0, wring + ((tt + 1) & 3) * WS_BYTES, 0); SB; }
      mma_i(xf1, wf1, 1); SB;
      gap(integral_constant<int, 4>{}); gap(integral_constant<int, 5>{}); SB;
      mma_i(xf1, wf1, 2); SB;
      gap(integral_constant<int, 6>{}); gap(integral_constant<int, 7>{}); SB;
      mma_i(xf1, wf1, 3); SB;
      if constexpr (WNT > 4) mma_i(xf1, wf1, 4);
#undef SB
      return;
    }
    const bool early = SPREAD == 0 || (SPREAD == 2 && wave < 4);
    if (early) issue();
    read_x(xf1, xs, kw, 1);
    read_w(wf1, ws, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(xf0, wf0);
    __builtin_amdgcn_sched_barrier(0);
    if (!early) issue();
    if constexpr (!FINAL) {
      // first k-step of the next k-tile: visible since this k-tile's barrier
      const char* xs2 = (kw == 2) ? xbuf + ((g + 1 - g_begin) & 1) * xbuf_bytes : xs;
      set_xoff(kw == 2 ? 0 : kw + 1);
      read_x(xf0, xs2, kw == 2 ? 0 : kw + 1, 0);
      read_w(wf0, wring + ((tt + 1) & 3) * WS_BYTES, 0);
    } else {
      fetch_res();
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(xf1, wf1);
  };
  using std::integral_constant;
  {
    int tt = 0;
    using KT0 = integral_constant<int, 0>; using KT1 = integral_constant<int, 1>; using KT2 = integral_constant<int, 2>;
    using T_ = integral_constant<bool, true>; using F_ = integral_constant<bool, false>;
    for (int g = g_begin; g + 1 < g_end; ++g, tt += 3) {
      ktile(g, tt, KT0{}, F_{}, T_{});
      ktile(g, tt + 1, KT1{}, F_{}, T_{});
      ktile(g, tt + 2, KT2{}, F_{}, T_{});
    }
    if (NT > 0) {
      ktile(g_end - 1, tt, KT0{}, F_{}, F_{});
      ktile(g_end - 1, tt + 1, KT1{}, F_{}, F_{});
      ktile(g_end - 1, tt + 2, KT2{}, T_{}, F_{});
    } else {
      fetch_res();
    }
  }
  RING_STAMP(2);
  if (p.splits > 1 && p.inred) {
    // ---- in-launch split-K sum (guide 5, "In-launch split-K reduction"; placement-independent, no spin, no float atomics) ----
    // every slice stores its accumulators as a register image (f32x4 per lane, 1 KiB coalesced wave-stores), drains, and ONE lane
    // releases (agent scope) and takes a ticket; the block that draws the last ticket acquires, sums the slices IN SLICE ORDER
    // (its own from registers) - bitwise the same whichever block comes last - and writes the output.
    const long tile_lin = (long)tile_m * p.tiles_n + tile_n;
    constexpr int IMG = WNT * WMT * 64;  // f32x4 per wave image
    f32x4* const slab4 = reinterpret_cast<f32x4*>(p.slab);
    f32x4* img = slab4 + ((tile_lin * p.splits + sid) * NW + wave) * IMG + lane;
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) img[(i * WMT + j) * 64] = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // all slab stores of the block issued and acknowledged; the LDS is idle from here on
    unsigned* const flag = reinterpret_cast<unsigned*>(smem);
    if (t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // keep: the compiler may drop the fence's own wait (guide G16 pitfall 12)
      *flag = __hip_atomic_fetch_add(p.tile_cnt + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned ticket = *flag;
    if (ticket != (unsigned)(p.splits - 1)) return;
    if (t == 0) {
      __hip_atomic_store(p.tile_cnt + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // sum = ((p0 + p1) + p2) + ... over ALL slices' images in slice order, this block's own re-read like the others (it is the
    // same fp32 values; keeping it in registers beside a running sum and a slice in flight needs 240 VGPRs and spilled).
    // Each slice is requested whole - 20 independent 16-byte loads in flight - before it is added.
    f32x4 tmp[WNT][WMT];
    for (int s2 = 0; s2 < p.splits; ++s2) {
      const f32x4* src = slab4 + ((tile_lin * p.splits + s2) * NW + wave) * IMG + lane;
#pragma unroll
      for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WMT; ++j) tmp[i][j] = src[(i * WMT + j) * 64];
      if (s2 == 0) {
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
          for (int j = 0; j < WMT; ++j) acc[i][j] = tmp[i][j];
      } else {
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
          for (int j = 0; j < WMT; ++j) acc[i][j] += tmp[i][j];
      }
    }
  }
  if constexpr (PAIR) {
    gemm_epilogue_pairs<WNT, WMT>(p, acc, m0, n0, wm, wn, frow, fq, 0, r2, r1, pre_res, bpre, pre_bias);
  } else {
    gemm_epilogue<WNT, bf16, WMT>(p, acc, m0, n0, wm, wn, frow, fq, 0, sid, rres, pre_res, bpre, pre_bias);
  }
  RING_STAMP(3);
}

#ifdef CRG_RING_STAMPS
}  // namespace crg_mm
extern "C" int crg_debug_read_ring(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(crg_mm::crg_ring_stamps), sizeof(unsigned long long) * (size_t)n);
}
namespace crg_mm {
#endif

// Host entry (called from gemm_conv.hip's launch_kernel in place of the 2-stage 256-row kernel): bf16 in / bf16 out.
int launch_conv_ring(crg_ctx* ctx, hipStream_t st, const GemmP& p, int wnt, int spread) {
  void (*kern)(GemmP) = nullptr;
  const bool lin = p.halo_lin != 0;
#define CRG_RING_PICK(W, S) (p.pair ? (lin ? conv3_ring_kernel<W, true, S, true> : conv3_ring_kernel<W, true, S, false>) \
                                    : (lin ? conv3_ring_kernel<W, false, S, true> : conv3_ring_kernel<W, false, S, false>))
  // (the DMA-issue placements 2 and 3 measured no gain over 1 - DESIGN 6, round 2 - and are no longer instantiated; they stay in the
  //  template for the record)
  if (spread >= 4 && !p.inred) return launch_conv_pp(ctx, st, p, wnt, spread - 4);
  if (wnt == 5) kern = spread ? CRG_RING_PICK(5, 1) : CRG_RING_PICK(5, 0);
  else if (wnt == 4) kern = spread ? CRG_RING_PICK(4, 1) : CRG_RING_PICK(4, 0);
  else return crg_fail(ctx, -22, "conv ring: unsupported tile width %d", wnt);
#undef CRG_RING_PICK
  const int BN = 32 * wnt, TP = 256;
  const int XP = (p.halo_lin || p.Wo > TP) ? (TP + 2 + 7) / 8 : ((TP / p.Wo) * (p.Wo + 2) + 7) / 8;
  if (XP > 40) return crg_fail(ctx, -22, "conv ring: row buffer of %d pieces unsupported", XP);
  const size_t lds = (size_t)4 * BN * 128 + (size_t)2 * XP * 1024;
  if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), 160 * 1024, "conv ring")) return rc;
  hipLaunchKernelGGL(kern, dim3(p.tile_count * p.splits, 1, 1), dim3(512), lds, st, p);
  CRG_CHECK_LAUNCH(ctx, "conv_ring");
  return 0;
}

}  // namespace crg_mm
