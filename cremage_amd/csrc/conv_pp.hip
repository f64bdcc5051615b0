// 3x3 / stride 1 / pad 1 conv on 256-pixel tiles with the two waves of every SIMD out of step (gfx950).
//
// (Round 2's conv3_ring_kernel - the same tile, LDS images and DMA pieces with all eight waves in lockstep - was removed in round 4.)
// Same math, operand layout and row-halo staging as conv3_rowhalo_kernel<.., MT = 2> (gemm_conv.hip); the pipeline:
//   * the weight ring has FOUR slots: W(t + 3) is issued in k-tile t and only has to be in LDS at the top of k-tile t + 2;
//   * the row buffer of group g + 1 (R x (W + 2) pixels of one 64-channel chunk and kernel row, shared by the three taps kw = 0..2) is
//     issued whole in the first k-tile of group g (its slot was last read in group g - 1);
//   * every wait is a counted s_waitcnt vmcnt(n) that leaves the newest issue batch in flight, followed by ONE raw s_barrier per
//     k-tile; nothing in the loop drains to 0; every piece goes through a buffer descriptor whose range check supplies the zeros of
//     the padding and of the tile tails.
// LDS: 4 x BN x 128 B (80 KB at BN = 160) + 2 x <= 36 KB row buffers = 152 KB: one block per CU, two waves per SIMD.
// Tile = 256 output pixels x BN
// output channels on 8 waves (4 x 2, each 64 x 16 WNT), K in (64-channel chunk, kernel row) groups of three k-tiles that share
// one row buffer, weight k-tiles through a 4-slot ring, everything by buffer_load ... lds behind counted s_waitcnt vmcnt(n).
// The STAGGER: one barrier per k-tile, waves 0-3 multiply k-tile t and then read k-tile t + 1 while waves 4-7 read k-tile t and then
// multiply it (described in front of its code further down): 8 % faster than the ring kernel, whose eight waves do each of the three
// things in lockstep so that the three costs ADD (0.60 us of MFMAs + 0.25 us of LDS reads + 0.19 us of DMA issue = the 1.0 - 1.1 us of
// a k-tile).  Two other schedules on the same tile were built and measured in round 3 and removed in round 4: a 4-barrier
// ping-pong (load section -> barrier -> 20 MFMAs -> barrier, waves 4-7 one barrier behind: as fast as the ring kernel, no faster - a
// barrier hand-off idles the matrix pipe for ~180 cycles) and the stagger with its DMA pieces threaded between the MFMA groups (3 % slower).
#include "gemm_shared.h"
#include <type_traits>

namespace crg_mm {

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void pp_wait_vm_n(int n) {  // wave-uniform n: s_waitcnt vmcnt(n), n <= 12
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
    default: wait_vmcnt<12>(); break;
  }
}

// One LDS-DMA piece through a raw buffer descriptor: lane -> 16 bytes at base + voff + soff, zeros when out of [0, bytes).
__device__ __forceinline__ void pp_dma16(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
#endif
}

#define PP_BARRIER()                       \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
  } while (0)
#define PP_SB __builtin_amdgcn_sched_barrier(0)

#ifdef CRG_PP_STAMPS
// dev instrumentation (tools/pp_stamp_probe.py; tools/build_one_variant.sh stamps conv_pp -DCRG_PP_STAMPS): wall_clock64 stamps of wave 0
// of every block: 0 entry, 1 first DMA piece about to be issued, 2 first operands landed (prologue barrier passed), 3 K loop done, 4 epilogue
// stores issued
__device__ unsigned long long crg_pp_stamps[1024 * 8];
#define PP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) crg_pp_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define PP_STAMP(i)
#endif

}  // namespace

template <int WNT, bool PAIR, bool LIN>
__global__ __launch_bounds__(512, 2) void conv3_pp_kernel(GemmP p) {
  constexpr int WMT = 4, NW = 8, TP = 256, WST = 4;
  constexpr int BN = 32 * WNT;
  constexpr int WS_BYTES = BN * 128;
  constexpr int WRG = BN / 8;
  constexpr int WL = (WRG + NW - 1) / NW;
  constexpr int XI = 5;  // row-buffer pieces (8 pixels = 1 KiB) per wave and group: <= 36 pieces (TP + 2 R pixels, W >= 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  PP_STAMP(0);
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool grp_b = wave >= 4;  // the second-dispatched half: SIMD partners of waves 0-3
  const int wm = wave >> 1, wn = wave & 1;
  const int Wd = p.Wo, Hd = p.Ho, sh = p.up ? 1 : 0;
  constexpr bool lin = LIN;
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

  // DMA pieces this wave issues per weight k-tile / per row buffer (wave-uniform)
  const int nW = (WRG - wave + NW - 1) / NW;
  const int nX = (XP - wave + NW - 1) / NW;

  // per-lane source descriptors of the DMA pieces (pixel index of the buffer position this lane fills + validity of its three kernel rows)
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
      xp0[i] = img * p.H * p.W + (w >> sh) + ((h - 1) >> sh) * p.W;
    }
  }
  int wvo[WL];
#pragma unroll
  for (int q = 0; q < WL; ++q) {
    const int pos = (wave + NW * q) * 8 + rsub;
    const int n = n0 + (PAIR ? unpair_col<WNT>(pos) : pos);
    wvo[q] = (n < p.N && (wave + NW * q) < WRG) ? (int)((long)n * p.ldw * 2) + clog * 16 : OOB;
  }
  const int g_begin = sid * p.ks_q + (sid < p.ks_r ? sid : p.ks_r);
  const int g_end = g_begin + p.ks_q + (sid < p.ks_r ? 1 : 0);
  const int NT = 3 * (g_end - g_begin);

  struct XGroup { const void* base; unsigned bytes; int Cs2, soff, drow; unsigned bit; char* xb; bool mid; };
  auto x_group = [&](int g) {
#ifdef CRG_ABL_DMAHOT  // timing-only ablation: every DMA piece re-reads the slice's first group / k-tile (cache-resident sources)
    const int gs = g_begin;
    const int c = gs / 3, kh = gs - 3 * c;
#else
    const int c = g / 3, kh = g - 3 * c;
#endif
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
      pp_dma16(r.base, r.bytes, r.xb + jp * 1024, vo, r.soff);
    }
  };
  auto w_piece = [&](int q, int tt) {  // piece q of this slice's weight k-tile tt into its ring slot
    if ((wave + NW * q) < WRG)
#ifdef CRG_ABL_DMAHOT
      pp_dma16(p.w, p.w_bytes, wring + (tt & 3) * WS_BYTES + (wave + NW * q) * 1024, wvo[q], (3 * g_begin) * 128);
#else
      pp_dma16(p.w, p.w_bytes, wring + (tt & 3) * WS_BYTES + (wave + NW * q) * 1024, wvo[q], (3 * g_begin + tt) * 128);
#endif
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

  // fragment byte offsets of the three tap columns (12 VGPRs, loop-invariant: with ONE fragment set there is room, and the load
  // section - which runs beside the partner's MFMA cluster and gets the leftover VALU issue slots - stays almost free of VALU work);
  // the second k-step is xoff ^ 64 (k-step = bit 2 of the 16-byte chunk index, XOR-swizzled)
  int xoff3[3][WMT];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int j = 0; j < WMT; ++j) {
      const int u = xb0[j] + kw;
      xoff3[kw][j] = (u << 7) + (((u & 7) ^ fq) << 4);
    }
  bf16x8 xf[WMT], wf[WNT];    // k-step 0
  bf16x8 xf1[WMT], wf1[WNT];  // k-step 1
  auto read_x = [&](const char* xs, int kw, int ks) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xs + (xoff3[kw][j] ^ (ks << 6)));
  };
  auto read_x1 = [&](const char* xs, int kw) {
#pragma unroll
    for (int j = 0; j < WMT; ++j) xf1[j] = *reinterpret_cast<const bf16x8*>(xs + (xoff3[kw][j] ^ 64));
  };
  const int wb0 = (wn * (16 * WNT) + frow) * 128 + ((fq ^ (frow & 7)) << 4);
  auto read_w = [&](const char* ws, int ks) {
    const char* base = ws + (wb0 ^ (ks << 6));
#pragma unroll
    for (int i = 0; i < WNT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
  };
  auto read_w1 = [&](const char* ws) {
    const char* base = ws + (wb0 ^ 64);
#pragma unroll
    for (int i = 0; i < WNT; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(base + i * 2048);
  };
  auto mma1 = [&]() {
#ifdef CRG_ABL_NOMMA
#pragma unroll
    for (int i = 0; i < WNT; ++i) asm volatile("" ::"v"(wf1[i]));
#pragma unroll
    for (int j = 0; j < WMT; ++j) asm volatile("" ::"v"(xf1[j]));
#else
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(wf1[i], xf1[j], acc[i][j]);
#endif
  };
  auto mma = [&]() {
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

  // residual / bias of this lane: requested in the load section of the LAST phase (nothing waits on the DMA ring any more)
  bf16x4 rres[PAIR ? 1 : WNT][PAIR ? 1 : WMT];
  bf16x8 r2[PAIR ? (WNT / 2 > 0 ? WNT / 2 : 1) : 1][PAIR ? WMT : 1];
  bf16x4 r1[PAIR ? WMT : 1];
  f32x4 bpre[WNT];
  const bool whole = p.splits == 1;  // this launch writes the finished output (no reduce kernel behind it)
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

  // ---- the STAGGER (MI355X_MICROARCH.md "Two waves per SIMD" item 9) -------------------------------------------------
  // One barrier per k-tile, as in the ring kernel, and the same issue / wait rule (interval t = between barriers t and t + 1:
  // W(t+3) and, in the first k-tile of a group, X(g+1) are issued; its end waits for everything issued BEFORE it), but the two
  // waves of a SIMD run half an interval apart with no barrier between the halves:
  //     waves 0-3:  barrier t | 40 MFMAs of k-tile t | issue | 18 fragment reads of k-tile t+1 | wait
  //     waves 4-7:  barrier t | 18 fragment reads of k-tile t | issue | 40 MFMAs of k-tile t   | wait
  // so one wave of every SIMD multiplies while its partner reads and issues, and the matrix pipe passes from one to the other
  // without a barrier hand-off in between (a 4-barrier ping-pong loses ~180 cycles of matrix pipe per hand-off).
  // Hazards: waves 0-3 read k-tile t+1 during interval t, so it has to be in LDS before barrier t: it was issued in interval
  // t-2 and the wait at the end of interval t-1 covers it.  W(t+3) overwrites the slot of W(t-1), last read (by waves 4-7) at
  // the start of interval t-1; X(g+1), issued in interval 3g, overwrites X(g-1), last read at the start of interval 3g-1.
  auto reads = [&](int g, int tt, auto KWc) {
    constexpr int kw = decltype(KWc)::value;
    const char* xs = xbuf + ((g - g_begin) & 1) * xbuf_bytes;
    const char* ws = wring + (tt & 3) * WS_BYTES;
    read_x(xs, kw, 0);
    read_w(ws, 0);
    read_x1(xs, kw);
    read_w1(ws);
  };
  auto issue = [&](int g, int tt, auto KWc, bool e_w3, bool e_x) {
    constexpr int kw = decltype(KWc)::value;
#ifndef CRG_ABL_NODMA
    if constexpr (kw == 0) {
      if (e_x) {
        const XGroup xg = x_group(g + 1);
#pragma unroll
        for (int i = 0; i < XI; ++i) x_piece(i, xg);
      }
    }
    if (e_w3) {
#pragma unroll
      for (int q = 0; q < WL; ++q) w_piece(q, tt + 3);
    }
#endif
  };
  auto mma2 = [&](auto KWc) {
    constexpr int kw = decltype(KWc)::value;
    if constexpr (LIN && kw != 1) {  // linear row buffer: a tap that would wrap around an image row contributes zeros
#pragma unroll
      for (int j = 0; j < WMT; ++j)
        if ((edge >> (kw == 0 ? j : 8 + j)) & 1u) { xf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; xf1[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
    }
    mma();
    mma1();
  };
  // wait at the end of interval tt: leaves in flight what the interval itself issued
  auto end_wait = [&](auto KWc, auto STEADYc) {
    constexpr int kw = decltype(KWc)::value;
    constexpr bool STEADY = decltype(STEADYc)::value;
    if constexpr (!STEADY) { wait_vmcnt<0>(); return; }
    if constexpr (kw == 0) {
      if (nW == 3) { if (nX == 5) wait_vmcnt<8>(); else wait_vmcnt<7>(); }
      else { if (nX == 5) wait_vmcnt<7>(); else wait_vmcnt<6>(); }
    } else {
      if (nW == 3) wait_vmcnt<3>(); else wait_vmcnt<2>();
    }
  };
  using std::integral_constant;
  using K0 = integral_constant<int, 0>; using K1 = integral_constant<int, 1>; using K2 = integral_constant<int, 2>;
  using T_ = integral_constant<bool, true>; using F_ = integral_constant<bool, false>;
  // one interval.  GB: waves 4-7.  LASTT: the slice's last k-tile (no wait, no next k-tile; residual / bias requested)
  auto interval = [&](int g, int tt, auto KWc, auto STEADYc, auto GBc, auto LASTc) {
    constexpr int kw = decltype(KWc)::value;
    constexpr bool STEADY = decltype(STEADYc)::value;
    constexpr bool GB = decltype(GBc)::value;
    constexpr bool LASTT = decltype(LASTc)::value;
    using KN = integral_constant<int, (kw + 1) % 3>;
    PP_BARRIER();
    PP_SB;
    if constexpr (GB) {
      reads(g, tt, KWc);
      PP_SB;
      if constexpr (STEADY) issue(g, tt, KWc, true, true);
      if constexpr (LASTT) fetch_res();
      PP_SB;
      mma2(KWc);
      PP_SB;
      if constexpr (!LASTT) end_wait(KWc, STEADYc);
    } else {
      mma2(KWc);
      PP_SB;
      if constexpr (STEADY) issue(g, tt, KWc, true, true);
      if constexpr (LASTT) fetch_res();
      if constexpr (!LASTT) {
        reads(kw == 2 ? g + 1 : g, tt + 1, KN{});
        PP_SB;
        end_wait(KWc, STEADYc);
      }
    }
  };
  auto run = [&](auto GBc) {
    constexpr bool GB = decltype(GBc)::value;
    int tt = 0;
    if constexpr (!GB) {  // waves 0-3 read their first fragments behind the prologue's wait (every later set: one interval ahead)
      PP_BARRIER();
      PP_STAMP(2);
      reads(g_begin, 0, K0{});
    }
    for (int g = g_begin; g + 1 < g_end; ++g, tt += 3) {
      interval(g, tt, K0{}, T_{}, GBc, F_{});
      interval(g, tt + 1, K1{}, T_{}, GBc, F_{});
      interval(g, tt + 2, K2{}, T_{}, GBc, F_{});
    }
    interval(g_end - 1, tt, K0{}, F_{}, GBc, F_{});
    interval(g_end - 1, tt + 1, K1{}, F_{}, GBc, F_{});
    interval(g_end - 1, tt + 2, K2{}, F_{}, GBc, T_{});
  };
  if (NT > 0) {
    // prologue: X(g0), W(0), W(1), W(2); W(0), W(1) and X(g0) have to be there (waves 0-3 read k-tile 1 during interval 0)
    PP_STAMP(1);
    {
      const XGroup xg = x_group(g_begin);
#pragma unroll
      for (int i = 0; i < XI; ++i) x_piece(i, xg);
    }
#pragma unroll
    for (int q = 0; q < WL; ++q) w_piece(q, 0);
#pragma unroll
    for (int q = 0; q < WL; ++q) w_piece(q, 1);  // NT >= 3
#pragma unroll
    for (int q = 0; q < WL; ++q) w_piece(q, 2);
    pp_wait_vm_n(nW);
    if (grp_b) { PP_BARRIER(); run(T_{}); }  // (the extra barrier pairs with the one waves 0-3 pass in front of their first reads)
    else run(F_{});
  } else {
    fetch_res();
  }
  PP_STAMP(3);
  if constexpr (PAIR) {
    // tile statistics (p.gstat_rows == TP): the waves' column sums meet in the (now idle) weight ring, 160 threads fold the four row waves
    float* const tile_lds = (p.gstat && p.gstat_rows == TP) ? reinterpret_cast<float*>(smem) : nullptr;
    gemm_epilogue_pairs<WNT, WMT>(p, acc, m0, n0, wm, wn, frow, fq, 0, r2, r1, pre_res, bpre, pre_bias, tile_lds);
    if (tile_lds) {
      __syncthreads();
      if (t < BN) {
        const int wn_ = t / (16 * WNT), col = t - wn_ * (16 * WNT);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          const float* L1 = tile_lds + (w4 * 2 + wn_) * (32 * WNT);
          s1 += L1[col];
          s2 += L1[16 * WNT + col];
        }
        const int n = n0 + t;
        if (n < p.N) {
          p.gstat[(long)tile_m * p.N + n] = s1;
          p.gstat[p.gstat_plane + (long)tile_m * p.N + n] = s2;
        }
      }
    }
  } else {
    gemm_epilogue<WNT, bf16, WMT>(p, acc, m0, n0, wm, wn, frow, fq, 0, sid, rres, pre_res, bpre, pre_bias);
  }
  PP_STAMP(4);
}

#ifdef CRG_PP_STAMPS
}  // namespace crg_mm
extern "C" int crg_debug_read_pp(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(crg_mm::crg_pp_stamps), sizeof(unsigned long long) * (size_t)n);
}
namespace crg_mm {
#endif

// Host entry (called from gemm_conv.hip's launch_kernel in place of the 2-stage 256-row kernel): bf16 in / bf16 out.
int launch_conv_pp(crg_ctx* ctx, hipStream_t st, const GemmP& p, int wnt) {
  void (*kern)(GemmP) = nullptr;
  const bool lin = p.halo_lin != 0;
#define CRG_PP_PICK(W) (p.pair ? (lin ? conv3_pp_kernel<W, true, true> : conv3_pp_kernel<W, true, false>) \
                               : (lin ? conv3_pp_kernel<W, false, true> : conv3_pp_kernel<W, false, false>))
  if (wnt == 5) kern = CRG_PP_PICK(5);
  else if (wnt == 4) kern = CRG_PP_PICK(4);
  else return crg_fail(ctx, -22, "conv pp: unsupported tile width %d", wnt);
#undef CRG_PP_PICK
  const int BN = 32 * wnt, TP = 256;
  const int XP = (p.halo_lin || p.Wo > TP) ? (TP + 2 + 7) / 8 : ((TP / p.Wo) * (p.Wo + 2) + 7) / 8;
  if (XP > 40) return crg_fail(ctx, -22, "conv pp: row buffer of %d pieces unsupported", XP);
  const size_t lds = (size_t)4 * BN * 128 + (size_t)2 * XP * 1024;
  if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), 160 * 1024, "conv pp")) return rc;
  hipLaunchKernelGGL(kern, dim3(p.tile_count * p.splits, 1, 1), dim3(512), lds, st, p);
  CRG_CHECK_LAUNCH(ctx, "conv_pp");
  return 0;
}

}  // namespace crg_mm
