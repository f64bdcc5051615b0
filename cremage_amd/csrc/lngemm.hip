// LayerNorm fused into the GEMM that consumes it, for short K (the 64x64 level of the SD1.5 UNet: K = 320, M = 32768).
//
//   Y[m][n] = epi( LN(X[m][:]) . W[n][:] + bias[n] ),   LN(x) = (x - mean) * rsqrt(var + eps) * gamma + beta
//
// replaces nn.LayerNorm + the Linear behind it in BasicTransformerBlock._forward (attention.py:908-912):
//   norm1 -> to_q | to_k | to_v (one launch over the stacked weights),  norm2 -> to_q,  norm3 -> GEGLU proj (attention.py:88-96).
// Unfused, the normalised tensor is written to HBM and read back by one to three GEMM launches whose 128 x 160 tiles have only
// FIVE k-tiles: launch, first DMA and epilogue are most of their 20-110 us.  Here a block keeps its 128 rows RESIDENT in LDS
// (128 x K bf16 = 80 KB at K = 320, staged once by LDS-DMA through a buffer descriptor, rows past M zero-filled by the range
// check), normalises them in place (8 lanes per row, two-pass mean / centred variance in fp32, one rounding to bf16 - the
// arithmetic of layernorm_kernel), and then walks ALL n-tiles of the weight: the weight k-tiles stream through a 4-slot ring
// without a break at n-tile boundaries (counted vmcnt, one barrier per k-tile), so an n-tile's epilogue (bias, GEGLU, 16-byte
// stores) runs while the next n-tile's weights are already landing.  X is read once, LN(X) never leaves the chip.
// 8 waves = 4 (m) x 2 (n), wave tile 32 x 16*WNT; LDS = 16 KB * K/64 + 4 * BN * 128 B (160 KB at K = 320, BN = 160).
#include "gemm_shared.h"
#include <stdlib.h>

namespace crg_mm {

struct LnGemmP {
  const bf16* x; long ldx;
  const float* gamma; const float* beta; float eps;
  const bf16* w; long ldw;     // packed [N][K] (GEGLU: value / gate rows interleaved in 16-row groups, crg_pack_weight)
  const float* bias;           // fp32 [N] (GEGLU: packed like the rows) or null
  const bf16* res; long ldr; unsigned res_bytes;  // optional residual [M][ldr] added after the bias (plain epilogue only)
  bf16* y; long ldy;
  int M, N, K, epi;
  unsigned x_bytes, w_bytes, y_bytes;
  // columns n >= vt_n0 (a multiple of the tile width; vt == null: none) are written TRANSPOSED: vt[(m / vt_T) * Cv + n - vt_n0][m % vt_T],
  // row length vt_ld - the V^T operand of crg_attention, so that to_q | to_k | to_v stay one launch and the flash kernel keeps
  // its conflict-free transposed-V staging (the row-major-V variant measured 10-20 % slower on the 4096-token self-attention)
  bf16* vt; int vt_n0, vt_T; long vt_ld; unsigned vt_bytes;
};

typedef __attribute__((address_space(3))) void* lptr_t;

static __device__ __forceinline__ void ln_dma16(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
#endif
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// Epilogue stores through a buffer descriptor: an out-of-range lane (row >= M, column group >= N: offset forced past the end) is
// DROPPED by the range check, so the store INSTRUCTION COUNT per wave is a compile-time constant whatever the tails are -
// which is what lets the counted vmcnt waits below leave exactly these stores in flight (vmcnt counts stores on gfx950).
static __device__ __forceinline__ void ln_store16(void* base, unsigned bytes, int voff, const bf16x8& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  u32x4 r;
  __builtin_memcpy(&r, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(r, rs, voff, 0, 0);
#endif
}
static __device__ __forceinline__ void ln_store8(void* base, unsigned bytes, int voff, const bf16x4& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  u32x2 r;
  __builtin_memcpy(&r, &v, 8);
#ifdef CRG_LN_ABL_NOSTORE  // timing ablation only: every 8-byte store goes out of range (dropped by the range check, still counted in vmcnt)
  voff = (int)0x80000000;
#endif
  __builtin_amdgcn_raw_buffer_store_b64(r, rs, voff, 0, 0);
#endif
}

static __device__ __forceinline__ bf16x8 ln_load16(const void* base, unsigned bytes, int voff) {
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
  __builtin_memcpy(&v, &r, 16);
#endif
  return v;
}
static __device__ __forceinline__ bf16x4 ln_load8(const void* base, unsigned bytes, int voff) {
  bf16x4 v = {0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
  __builtin_memcpy(&v, &r, 8);
#endif
  return v;
}

static __device__ __forceinline__ void ln_store2(void* base, unsigned bytes, int voff, bf16 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
  short r;
  __builtin_memcpy(&r, &v, 2);
  __builtin_amdgcn_raw_buffer_store_b16(r, rs, voff, 0, 0);
#endif
}

static __device__ __forceinline__ void ln_wait(int n) {  // wave-uniform s_waitcnt vmcnt(n); n outside the table waits for everything
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
    case 31: wait_vmcnt<31>(); break;
    case 32: wait_vmcnt<32>(); break;
    case 33: wait_vmcnt<33>(); break;
    case 34: wait_vmcnt<34>(); break;
    case 35: wait_vmcnt<35>(); break;
    case 36: wait_vmcnt<36>(); break;
    case 37: wait_vmcnt<37>(); break;
    case 38: wait_vmcnt<38>(); break;
    case 39: wait_vmcnt<39>(); break;
    case 40: wait_vmcnt<40>(); break;
    case 41: wait_vmcnt<41>(); break;
    case 42: wait_vmcnt<42>(); break;
    case 43: wait_vmcnt<43>(); break;
    case 44: wait_vmcnt<44>(); break;
    case 45: wait_vmcnt<45>(); break;
    case 46: wait_vmcnt<46>(); break;
    case 47: wait_vmcnt<47>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

// s_waitcnt vmcnt(nW + EXTRA) for the per-wave batch sizes that occur (WRG / 8 rounded either way): two scalar branches instead of
// the table's compare tree in front of every k-tile's barrier
template <int EXTRA>
static __device__ __forceinline__ void ln_wait_nw(int nW) {
  if (nW == 2) wait_vmcnt<2 + EXTRA>();
  else if (nW == 3) wait_vmcnt<3 + EXTRA>();
  else if (nW == 4) wait_vmcnt<4 + EXTRA>();
  else if (nW == 1) wait_vmcnt<1 + EXTRA>();
  else ln_wait(nW + EXTRA);
}

// (Round 3's ALIAS form of this kernel - the GEGLU ring laid over the resident rows for 256-column n-tiles - went when
//  lngemm_geglu_pp_kernel below, which keeps that layout, replaced it in round 4: 82 -> 77.6 us at 32768 x 2560 x 320.)
template <int WNT, int KT, bool GEGLU>
__global__ __launch_bounds__(512, 2) void lngemm_kernel(LnGemmP p) {
  constexpr bool PAIR = !GEGLU;  // plain outputs use the paired column mapping (16-byte stores); GEGLU its own [v16 | g16] packing
  constexpr int WMT = 2, NW = 8, BMR = 128, WST = 4;
  constexpr int BN = 32 * WNT;
  constexpr int WS_BYTES = BN * 128;
  constexpr int WRG = BN / 8;
  constexpr int WL = (WRG + NW - 1) / NW;
  constexpr int A_KT = BMR * 128;           // bytes of one resident k-tile [128 rows][64 k]
  constexpr int AP = KT * 16 / NW;          // A pieces per wave (KT * 16 pieces of 8 rows x 128 B)
  static_assert((KT * 16) % NW == 0, "A pieces must divide over the waves");
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ares = smem;                  // [KT][128][128 B], chunk XOR-swizzled by (row & 7)
  char* const wring = smem + KT * A_KT;     // [WST][BN x 128 B]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BMR;
  const int rsub = lane >> 3;
  const int clog = (lane & 7) ^ rsub;
  const int frow = lane & 15, fq = lane >> 4;
  const int nW = (WRG - wave + NW - 1) / NW;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int U = tiles_n * KT;               // weight k-tiles this block streams
  constexpr int NST = GEGLU ? WMT * (WNT / 2) : WMT * (WNT / 2 + (WNT & 1));  // store instructions per wave and n-tile
  constexpr int NSTT = WMT * (8 * (WNT / 2) + 4 * (WNT & 1));                // ... of an n-tile that is written transposed

  // ---- stage the 128 rows (all K) and the first three weight k-tiles ----
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int pa = wave + NW * i;           // piece: k-tile pa / 16, row group pa % 16
    const int kt = pa >> 4, rg = pa & 15;
    const int row = m0 + rg * 8 + rsub;
    const int vo = row < p.M ? (int)((long)row * p.ldx * 2) + clog * 16 : OOB;
    ln_dma16(p.x, p.x_bytes, Ares + kt * A_KT + rg * 1024, vo, kt * 128);
  }
  int wvo[WL];
#pragma unroll
  for (int q = 0; q < WL; ++q) {
    const int pos = (wave + NW * q) * 8 + rsub;
    const int nl = PAIR ? unpair_col<WNT>(pos) : pos;     // column of the tile whose weight row lives at LDS row `pos`
    wvo[q] = (wave + NW * q) < WRG ? (int)((long)nl * p.ldw * 2) + clog * 16 : OOB;
  }
  auto stage_w = [&](int u) {  // weight k-tile u = (n-tile u / KT, k-tile u % KT) into ring slot u & 3
    const int nt = u / KT, kt = u - nt * KT;
    char* ws = wring + (u & 3) * WS_BYTES;
    const int soff = (int)((long)nt * BN * p.ldw * 2) + kt * 128;  // rows past N fall beyond w_bytes: zero-filled
#pragma unroll
    for (int q = 0; q < WL; ++q)
      if ((wave + NW * q) < WRG) ln_dma16(p.w, p.w_bytes, ws + (wave + NW * q) * 1024, wvo[q], soff);
  };
  // (weight k-tiles past the last one: their rows lie beyond w_bytes, the DMA writes zeros into slots nobody reads - issuing them
  // anyway keeps every wait of the loop at its steady-state count and every fragment read unconditional)
  stage_w(0);
  stage_w(1);
  stage_w(2);

  // ---- LayerNorm of the resident rows, in place: wave w owns rows 16 w .. 16 w + 15, 8 lanes per row ----
  // (gamma == null: no LayerNorm - the kernel is then a plain row-resident GEMM for K = 320)
  if (p.gamma) {
    const int c = lane & 7, r8 = lane >> 3;
    f32x4 g0[KT], g1[KT], b0[KT], b1[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int k = kt * 64 + c * 8;
      g0[kt] = *reinterpret_cast<const f32x4*>(p.gamma + k);
      g1[kt] = *reinterpret_cast<const f32x4*>(p.gamma + k + 4);
      b0[kt] = *reinterpret_cast<const f32x4*>(p.beta + k);
      b1[kt] = *reinterpret_cast<const f32x4*>(p.beta + k + 4);
    }
    // the rows (older than the three weight batches) have landed once at most 3 nW operations of this wave are outstanding;
    // the gamma / beta loads above are younger still, the compiler waits for them itself at their first use
    ln_wait(nW * 3);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float inv_k = 1.0f / (float)(KT * 64);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = wave * 16 + it * 8 + r8;
      bf16x8 v[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) v[kt] = *reinterpret_cast<const bf16x8*>(Ares + kt * A_KT + lds_off(row, c));
      float s = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[kt][e];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * inv_k;
      float qv = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)v[kt][e] - mean;
          qv += d * d;
        }
      qv += __shfl_xor(qv, 1);
      qv += __shfl_xor(qv, 2);
      qv += __shfl_xor(qv, 4);
      const float rstd = rsqrtf(qv * inv_k + p.eps);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ga = e < 4 ? g0[kt][e] : g1[kt][e - 4], be = e < 4 ? b0[kt][e] : b1[kt][e - 4];
          o[e] = (bf16)(((float)v[kt][e] - mean) * rstd * ga + be);
        }
        *reinterpret_cast<bf16x8*>(Ares + kt * A_KT + lds_off(row, c)) = o;
      }
    }
  }
  // (the barrier of the first k-tile below orders these LDS writes before every wave's fragment reads)

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wb0 = (wn * (16 * WNT) + frow) * 128 + ((fq ^ (frow & 7)) << 4);
  int aoff[WMT];
#pragma unroll
  for (int j = 0; j < WMT; ++j) aoff[j] = lds_off(wm * 32 + j * 16 + frow, fq);  // second k-step: ^ 64

  // fragment sets: F0 = first 32-wide k-step of the CURRENT k-tile (requested during the previous k-tile, whose barrier already
  // made this k-tile's weights visible), F1 = its second k-step (requested at the top, consumed after F0's MFMAs)
  // The wave's activation fragments stay in REGISTERS for the whole kernel (KT x 2 k-steps x WMT fragments = 80 VGPRs at K = 320):
  // they are the same for every n-tile, and reading them from LDS in every k-tile was a third of the loop's ds_read_b128s.
  bf16x8 xa[KT][2][WMT];
  bf16x8 wf0[WNT], wf1[WNT];
  auto read_frags = [&](bf16x8 (&wf)[WNT], int slot, int ks) {
    const char* wbase = wring + slot * WS_BYTES + (wb0 ^ (ks << 6));
#pragma unroll
    for (int i = 0; i < WNT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wbase + i * 2048);
  };
  auto mma = [&](const bf16x8 (&xf)[WMT], const bf16x8 (&wf)[WNT]) {
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(wf[i], xf[j], acc[i][j]);
  };
  {
    // weight k-tile 0 landed (two batches may stay in flight); this barrier also publishes the normalised rows
    ln_wait(nW * 2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < WMT; ++j) xa[kt][ks][j] = *reinterpret_cast<const bf16x8*>(Ares + kt * A_KT + (aoff[j] ^ (ks << 6)));
    read_frags(wf0, 0, 0);
  }
  for (int nt = 0; nt < tiles_n; ++nt) {
    const int n0 = nt * BN;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int u = nt * KT + kt;
      // top(u): weight k-tile u + 1 has landed once only the youngest batch (u + 2) is outstanding - everything this k-tile
      // and the prefetch of the next one read is then visible behind the barrier; stores of the previous epilogue that are
      // still in the queue only make the wait err on the safe side
      // ... and the NST stores of the previous n-tile's epilogue, issued between the batches of k-tiles u_e + 3 and u_e + 4
      // (u_e = that n-tile's last k-tile), are younger than what the first two waits of an n-tile need: they stay in flight too
      if (kt < 2 && nt > 0) {
        bool transposed_prev = false;
        if constexpr (!GEGLU) {
          transposed_prev = p.vt && (nt - 1) * BN >= p.vt_n0;
          if (transposed_prev) ln_wait_nw<NSTT>(nW);
        }
        if (!transposed_prev) ln_wait_nw<NST>(nW);
      } else {
        ln_wait_nw<0>(nW);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // (ablation builds -DCRG_LN_ABL_NOMMA / -DCRG_LN_ABL_NODMA, N = 2560 GEGLU: 87 us full, 82 without the DMA, 66 without the
      // MFMAs, 53 without either: barrier + fragment reads + prologue / epilogue are the larger part.  Running the two waves of a
      // SIMD in opposite orders - w + 4 issues its first MFMA block before its fragment reads - measured 88 -> 91 us, not kept.)
      read_frags(wf1, u & 3, 1);
      __builtin_amdgcn_sched_barrier(0);
#ifndef CRG_LN_ABL_NOMMA
      mma(xa[kt][0], wf0);
#endif
      __builtin_amdgcn_sched_barrier(0);
#ifndef CRG_LN_ABL_NODMA
      stage_w(u + 3);  // slot of k-tile u - 1: every wave finished reading it before this k-tile's barrier
#endif
      // unconditional (after the last k-tile it reads a slot of zeros): behind a branch the compiler cannot count these reads and
      // makes the MFMAs below wait for lgkmcnt(0), i.e. for the prefetch as well as for their own operands
      read_frags(wf0, (u + 1) & 3, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifndef CRG_LN_ABL_NOMMA
      mma(xa[kt][1], wf1);
#else
      asm volatile("" ::"v"(wf1[0]), "v"(wf0[0]));
#endif
    }
    // ---- epilogue of n-tile nt (the ring keeps filling underneath): exactly NST store instructions per wave ----
    {
      constexpr int OOBS = (int)0x80000000;
      const int nb = n0 + wn * (16 * WNT);
      if constexpr (GEGLU) {
        // packed columns: [v 16 | g 16] groups; tiles (2u, 2u + 1) of this wave are value / gate (crg_pack_weight CRG_PACK_GEGLU)
#pragma unroll
        for (int u2 = 0; u2 < WNT / 2; ++u2) {
          const int pn = nb + u2 * 32 + fq * 4;       // packed column of the value tile
          const int jn = nb / 2 + u2 * 16 + fq * 4;   // output column
          const bool nok = pn + 20 <= p.N;            // the gate group sits 16 packed columns further
          f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = bv;
          if (p.bias) {
            const int pc = nok ? pn : 0;
            bv = *reinterpret_cast<const f32x4*>(p.bias + pc);
            bg = *reinterpret_cast<const f32x4*>(p.bias + pc + 16);
          }
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            const int m = m0 + wm * 32 + j * 16 + frow;
            const f32x4 v = acc[2 * u2][j] + bv, g = acc[2 * u2 + 1][j] + bg;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] * crg_gelu_erf_f(g[e]));
            ln_store8(p.y, p.y_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldy + jn) * 2) : OOBS, o);
          }
        }
      } else if (p.vt && n0 >= p.vt_n0) {
        // transposed n-tile (V^T): lane = one token x 8 (4) consecutive channels -> one 2-byte store per channel; the 16 lanes of
        // a quarter wave hold 16 consecutive tokens, i.e. 32 contiguous bytes of a V^T row per store instruction and quarter
        const int Cv = p.N - p.vt_n0;
#pragma unroll
        for (int j = 0; j < WMT; ++j) {
          const int m = m0 + wm * 32 + j * 16 + frow;
          const int bs = m / p.vt_T, tk = m - bs * p.vt_T;
          const long row0 = (long)bs * Cv - p.vt_n0;
#pragma unroll
          for (int u2 = 0; u2 < WNT / 2; ++u2) {
            const int n = nb + 32 * u2 + 8 * fq;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = e < 4 ? acc[2 * u2][j][e] : acc[2 * u2 + 1][j][e - 4];
              const float bsv = p.bias ? p.bias[n + e < p.N ? n + e : 0] : 0.f;
              ln_store2(p.vt, p.vt_bytes, (n + e < p.N && m < p.M) ? (int)(((row0 + n + e) * p.vt_ld + tk) * 2) : OOBS, (bf16)(v + bsv));
            }
          }
          if constexpr (WNT & 1) {
            const int n = nb + 16 * (WNT - 1) + 4 * fq;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float bsv = p.bias ? p.bias[n + e < p.N ? n + e : 0] : 0.f;
              ln_store2(p.vt, p.vt_bytes, (n + e < p.N && m < p.M) ? (int)(((row0 + n + e) * p.vt_ld + tk) * 2) : OOBS,
                        (bf16)(acc[WNT - 1][j][e] + bsv));
            }
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
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            const int m = m0 + wm * 32 + j * 16 + frow;
            f32x4 a4 = acc[2 * u2][j] + ba, b4 = acc[2 * u2 + 1][j] + bb;
            if (p.res) {  // out-of-range lanes read zeros (range check) and are dropped by the store below
              const bf16x8 r8 = ln_load16(p.res, p.res_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldr + n) * 2) : OOBS);
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
            ln_store16(p.y, p.y_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldy + n) * 2) : OOBS, o);
          }
        }
        if constexpr (WNT & 1) {
          const int n = nb + 16 * (WNT - 1) + 4 * fq;
          const bool nok = n + 4 <= p.N;
          f32x4 ba = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) ba = *reinterpret_cast<const f32x4*>(p.bias + (nok ? n : 0));
#pragma unroll
          for (int j = 0; j < WMT; ++j) {
            const int m = m0 + wm * 32 + j * 16 + frow;
            f32x4 a4 = acc[WNT - 1][j] + ba;
            if (p.res) {
              const bf16x4 r4 = ln_load8(p.res, p.res_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldr + n) * 2) : OOBS);
#pragma unroll
              for (int e = 0; e < 4; ++e) a4[e] += (float)r4[e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)a4[e];
            ln_store8(p.y, p.y_bytes, (nok && m < p.M) ? (int)(((long)m * p.ldy + n) * 2) : OOBS, o);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  wait_vmcnt<0>();  // the (zero-fill) DMA batches issued past the last k-tile must land before this block's LDS is handed on
}


// ---- GEGLU, two wave groups in opposite phases (round 4) -------------------------------------------------------------------------------
// At K = 320 the GEGLU epilogue of an n-tile costs a wave MORE vector-ALU cycles (bias, erf-GELU, product, rounding: ~110 per output,
// 32 outputs per lane) than the tile's 160 MFMAs cost matrix cycles, and in lngemm_kernel all eight waves do the one and then the other:
// the matrix pipe idles through every epilogue and the vector ALU through every k-tile (ablations on 32768 x 2560 x 320: 82 us, 72 with
// GELU = identity, 71 with the stores dropped).  Here the two waves of a SIMD are in OPPOSITE phases: waves 0-3 (one per SIMD) multiply
// the five k-tiles of an even n-tile while waves 4-7 work through the epilogue of the odd n-tile they finished before, in five pieces,
// one per step - then the roles swap.  A wave's accumulators are idle while it is in its epilogue phase, so no second set is needed;
// n-tiles are 128 packed columns wide (64 outputs) so that ONE group covers all 128 resident rows (wave tile 32 x 128, as before).
//   * one block barrier per step (= per weight k-tile of 16 KB); the weight stream is the same n-tile-major sequence as before, through
//     an 8-slot ring laid over the resident rows once their fragments are in registers (unit u -> slot (u + 5) & 7; seven units in
//     flight; every wave issues two 1 KiB pieces per step, so `s_waitcnt vmcnt(12)` at the top of step s proves that unit s has landed
//     whatever stores lie in between - a store that is older than twelve younger operations has long completed);
//   * the bias lives in LDS (10 KB at N = 2560): a global load in the loop would make the compiler drain the whole DMA queue;
//   * stores leave in every step instead of in bursts at the n-tile boundaries.
template <int KT>
__global__ __launch_bounds__(512, 2) void lngemm_geglu_pp_kernel(LnGemmP p) {
  constexpr int WNT = 8, WMT = 2, NW = 8, BMR = 128, BN = 128, NSLOT = 8;
  constexpr int WS_BYTES = BN * 128;        // 16 KB: one unit = (n-tile, k-tile) of the packed weight
  constexpr int A_KT = BMR * 128;           // 16 KB: one resident k-tile [128 rows][64 k]
  constexpr int AP = KT * 16 / NW;
  static_assert(KT <= 5 && (KT * 16) % NW == 0, "the resident rows must fit slots 0 .. 4");
  constexpr int OOB = (int)0x80000000;
  constexpr int BIAS_OFF = NSLOT * WS_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ares = smem;
  float* const sbias = reinterpret_cast<float*>(smem + BIAS_OFF);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wm = wave & 3;
  const int m0 = blockIdx.x * BMR;
  const int rsub = lane >> 3;
  const int clog = (lane & 7) ^ rsub;
  const int frow = lane & 15, fq = lane >> 4;
  const int tiles_n = p.N / BN;             // even (host)
  const int U = tiles_n * KT;

  // ---- the 128 rows (all K), the bias, the first three weight units (slots 5 .. 7 lie beyond the rows) ----
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int pa = wave + NW * i;
    const int kt = pa >> 4, rg = pa & 15;
    const int row = m0 + rg * 8 + rsub;
    const int vo = row < p.M ? (int)((long)row * p.ldx * 2) + clog * 16 : OOB;
    ln_dma16(p.x, p.x_bytes, Ares + kt * A_KT + rg * 1024, vo, kt * 128);
  }
  int wvo[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) wvo[q] = (int)((long)((wave + NW * q) * 8 + rsub) * p.ldw * 2) + clog * 16;
  int is_nt = 0, is_kt = 0;                 // the unit the next issue() requests
  auto issue = [&](int u) {                 // past the last unit: rows beyond w_bytes, zero-filled into slots nobody reads
    char* ws = smem + ((u + 5) & (NSLOT - 1)) * WS_BYTES;
    const int soff = (int)((long)is_nt * BN * p.ldw * 2) + is_kt * 128;
#pragma unroll
    for (int q = 0; q < 2; ++q) ln_dma16(p.w, p.w_bytes, ws + (wave + NW * q) * 1024, wvo[q], soff);
    if (++is_kt == KT) {
      is_kt = 0;
      ++is_nt;
    }
  };
  issue(0);
  issue(1);
  issue(2);
  f32x4 bq[4];  // the bias, on its way to LDS: all loads out before the first wait (N <= 8192, host)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = (t + 512 * r) * 4;
    bq[r] = (p.bias && i < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + i) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto put_bias = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (t + 512 * r) * 4;
      if (i < p.N) *reinterpret_cast<f32x4*>(sbias + i) = bq[r];
    }
  };

  if (p.gamma) {
    const int c = lane & 7, r8 = lane >> 3;
    f32x4 g0[KT], g1[KT], b0[KT], b1[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int k = kt * 64 + c * 8;
      g0[kt] = *reinterpret_cast<const f32x4*>(p.gamma + k);
      g1[kt] = *reinterpret_cast<const f32x4*>(p.gamma + k + 4);
      b0[kt] = *reinterpret_cast<const f32x4*>(p.beta + k);
      b1[kt] = *reinterpret_cast<const f32x4*>(p.beta + k + 4);
    }
    wait_vmcnt<0>();  // rows (and the three weight units, the bias, gamma / beta) have landed
    put_bias();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float inv_k = 1.0f / (float)(KT * 64);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = wave * 16 + it * 8 + r8;
      bf16x8 v[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) v[kt] = *reinterpret_cast<const bf16x8*>(Ares + kt * A_KT + lds_off(row, c));
      float s = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)v[kt][e];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const float mean = s * inv_k;
      float qv = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)v[kt][e] - mean;
          qv += d * d;
        }
      qv += __shfl_xor(qv, 1);
      qv += __shfl_xor(qv, 2);
      qv += __shfl_xor(qv, 4);
      const float rstd = rsqrtf(qv * inv_k + p.eps);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ga = e < 4 ? g0[kt][e] : g1[kt][e - 4], be = e < 4 ? b0[kt][e] : b1[kt][e - 4];
          o[e] = (bf16)(((float)v[kt][e] - mean) * rstd * ga + be);
        }
        *reinterpret_cast<bf16x8*>(Ares + kt * A_KT + lds_off(row, c)) = o;
      }
    }
  } else {
    wait_vmcnt<0>();
    put_bias();
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();  // normalised rows (or the raw ones) visible
  asm volatile("" ::: "memory");
  // the wave's activation fragments stay in registers for the whole kernel (KT x 2 k-steps x WMT = 80 VGPRs at K = 320)
  bf16x8 xa[KT][2][WMT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < WMT; ++j)
        xa[kt][ks][j] = *reinterpret_cast<const bf16x8*>(Ares + kt * A_KT + (lds_off(wm * 32 + j * 16 + frow, fq) ^ (ks << 6)));
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();        // every wave holds its fragments: the rows' LDS becomes ring slots 0 .. 4
  asm volatile("" ::: "memory");
  issue(3);
  issue(4);
  issue(5);
  issue(6);

  f32x4 acc[WNT][WMT];
#pragma unroll
  for (int i = 0; i < WNT; ++i)
#pragma unroll
    for (int j = 0; j < WMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wb0 = frow * 128 + ((fq ^ (frow & 7)) << 4);  // rows 16 i + frow share (row & 7): + i * 2048; second k-step: ^ 64

  int s = 0;  // step = unit multiplied in it
  auto top = [&]() {
    wait_vmcnt<12>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(s + 7);  // into the slot of unit s - 1, which every wave finished reading before this barrier
  };
  auto mma_step = [&](int kt) {
    const char* wbase = smem + ((s + 5) & (NSLOT - 1)) * WS_BYTES;
    bf16x8 wf0[WNT], wf1[WNT];
#pragma unroll
    for (int i = 0; i < WNT; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(wbase + wb0 + i * 2048);
#pragma unroll
    for (int i = 0; i < WNT; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(wbase + (wb0 ^ 64) + i * 2048);
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(wf0[i], xa[kt][0][j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
      for (int j = 0; j < WMT; ++j) acc[i][j] = CRG_MFMA_16x16x32(wf1[i], xa[kt][1][j], acc[i][j]);
  };
  // one (value, gate) tile pair x one 16-row block of n-tile nt: 4 outputs per lane, one 8-byte store; its accumulators start over
  auto epi_block = [&](int nt, int u2, int j) {
    const int pn = nt * BN + u2 * 32 + fq * 4;
    const int jn = nt * (BN / 2) + u2 * 16 + fq * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + pn), bg = *reinterpret_cast<const f32x4*>(sbias + pn + 16);
    const int m = m0 + wm * 32 + j * 16 + frow;
    const f32x4 v = acc[2 * u2][j] + bv, g = acc[2 * u2 + 1][j] + bg;
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] * crg_gelu_erf_f(g[e]));
    ln_store8(p.y, p.y_bytes, m < p.M ? (int)(((long)m * p.ldy + jn) * 2) : OOB, o);
    acc[2 * u2][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[2 * u2 + 1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // piece c = 0 .. KT - 1 of an n-tile's epilogue: its 8 blocks spread as evenly as the step count allows (KT = 5: 2 2 2 1 1)
  auto epi_piece = [&](int nt, int c) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if ((b * KT) / 8 == c) epi_block(nt, b >> 1, b & 1);
  };
  if (grp == 0) {
    for (int nt = 0; nt < tiles_n; nt += 2) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt, ++s) {
        top();
        mma_step(kt);
      }
#pragma unroll
      for (int c = 0; c < KT; ++c, ++s) {
        top();
        epi_piece(nt, c);
      }
    }
  } else {
    for (int nt = 0; nt < tiles_n; nt += 2) {
#pragma unroll
      for (int c = 0; c < KT; ++c, ++s) {
        top();
        if (nt > 0) epi_piece(nt - 1, c);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt, ++s) {
        top();
        mma_step(kt);
      }
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) epi_piece(tiles_n - 1, c);
  }
  wait_vmcnt<0>();  // the zero-fill batches issued past the last unit must land before this block's LDS is handed on
}

}  // namespace crg_mm

using namespace crg_mm;

extern "C" int crg_ln_gemm(crg_ctx* ctx, void* stream, const crg_lngemm_args* a) {
  if (!ctx || !a) return -22;
  CRG_REQUIRE(ctx, a->M > 0 && a->N > 0 && a->K > 0, "ln_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  CRG_REQUIRE(ctx, a->K == 320, "ln_gemm: the row-resident kernel is built for K = 320 (got %d): use crg_layernorm + crg_gemm", a->K);
  CRG_REQUIRE(ctx, a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->ldx >= a->K && a->ldw >= a->K, "ln_gemm: ldx=%ld / ldw=%ld must be >= K and keep rows 16-byte aligned",
              (long)a->ldx, (long)a->ldw);
  CRG_REQUIRE(ctx, (((uintptr_t)a->x | (uintptr_t)a->w | (uintptr_t)a->y | (uintptr_t)a->gamma | (uintptr_t)a->beta | (uintptr_t)a->bias) & 15) == 0,
              "ln_gemm: pointers must be 16-byte aligned");
  CRG_REQUIRE(ctx, (a->gamma == nullptr) == (a->beta == nullptr), "ln_gemm: gamma and beta come together (both NULL: no LayerNorm)");
  const bool geglu = a->epilogue == CRG_EPI_GEGLU;
  CRG_REQUIRE(ctx, a->epilogue == CRG_EPI_NONE || geglu, "ln_gemm: epilogue %d unsupported", a->epilogue);
  if (geglu) CRG_REQUIRE(ctx, a->N % 32 == 0, "ln_gemm: GEGLU needs packed N %% 32 == 0 (got %d)", a->N);
  const double xb = (double)a->M * a->ldx * 2.0, wb = (double)a->N * a->ldw * 2.0;
  CRG_REQUIRE(ctx, xb < 2147483648.0 && wb < 2147483648.0, "ln_gemm: operands must be < 2 GiB");
  LnGemmP p{};
  p.x = (const bf16*)a->x; p.ldx = a->ldx; p.gamma = a->gamma; p.beta = a->beta; p.eps = a->eps;
  p.w = (const bf16*)a->w; p.ldw = a->ldw; p.bias = a->bias; p.y = (bf16*)a->y; p.ldy = a->ldy;
  p.M = a->M; p.N = a->N; p.K = a->K; p.epi = a->epilogue;
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  if (a->residual) {
    CRG_REQUIRE(ctx, !geglu && !a->vt, "ln_gemm: the residual belongs to the plain epilogue");
    CRG_REQUIRE(ctx, a->ldr % 8 == 0 && a->ldr >= a->N && ((uintptr_t)a->residual & 15) == 0, "ln_gemm: residual rows must be 16-byte aligned and >= N wide");
    const double rb = (double)a->M * a->ldr * 2.0;
    CRG_REQUIRE(ctx, rb < 2147483648.0, "ln_gemm: residual must be < 2 GiB");
    p.res = (const bf16*)a->residual; p.ldr = a->ldr; p.res_bytes = (unsigned)rb;
  }
  if (a->vt) {
    CRG_REQUIRE(ctx, !geglu, "ln_gemm: a transposed column range and GEGLU do not combine");
    CRG_REQUIRE(ctx, a->vt_n0 > 0 && a->vt_n0 < a->N && a->vt_tokens > 0 && a->M % a->vt_tokens == 0 && a->vt_ld >= a->vt_tokens,
                "ln_gemm: transposed range n0=%d tokens=%d ld=%ld inconsistent with M=%d N=%d", a->vt_n0, a->vt_tokens, (long)a->vt_ld, a->M, a->N);
    const int bn_ = a->N % 160 == 0 ? 160 : 128;
    CRG_REQUIRE(ctx, a->vt_n0 % bn_ == 0, "ln_gemm: the transposed range must start on a %d-column tile (got %d)", bn_, a->vt_n0);
    const double vb = (double)(a->M / a->vt_tokens) * (a->N - a->vt_n0) * a->vt_ld * 2.0;
    CRG_REQUIRE(ctx, vb < 2147483648.0 && ((uintptr_t)a->vt & 1) == 0, "ln_gemm: V^T output must be < 2 GiB");
    p.vt = (bf16*)a->vt; p.vt_n0 = a->vt_n0; p.vt_T = a->vt_tokens; p.vt_ld = a->vt_ld; p.vt_bytes = (unsigned)vb;
  }
  const int n_out = geglu ? a->N / 2 : (a->vt ? a->vt_n0 : a->N);
  CRG_REQUIRE(ctx, a->ldy >= n_out, "ln_gemm: ldy=%ld < output width %d", (long)a->ldy, n_out);
  CRG_REQUIRE(ctx, geglu || (a->N % 8 == 0 && a->ldy % 8 == 0), "ln_gemm: N=%d / ldy=%ld must be multiples of 8 (16-byte output groups)", a->N, (long)a->ldy);
  CRG_REQUIRE(ctx, !geglu || a->ldy % 4 == 0, "ln_gemm: ldy=%ld must be a multiple of 4", (long)a->ldy);
  const double yb = (double)a->M * a->ldy * 2.0;
  CRG_REQUIRE(ctx, yb < 2147483648.0, "ln_gemm: output must be < 2 GiB");
  p.y_bytes = (unsigned)yb;
  const bool wide = !geglu && a->N % 160 == 0;
  hipStream_t st = (hipStream_t)stream;
  void (*kern)(LnGemmP);
  int bn;
  static const int use_pp = getenv("CRG_LN_PP") ? atoi(getenv("CRG_LN_PP")) : 1;  // dev knob: 0 = lockstep 128-column GEGLU tiles
  if (geglu && use_pp && a->N % 256 == 0 && a->N <= 8192) {  // two wave groups in opposite phases (bias in LDS: N <= 8192)
    kern = lngemm_geglu_pp_kernel<5>;
    const size_t lds_pp = (size_t)8 * 128 * 128 + (size_t)a->N * 4;
    if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), 160 * 1024, "ln_gemm")) return rc;
    const double flops = 2.0 * a->M * (double)a->N * a->K;
    const double bytes = (double)a->M * a->K * 2 + (double)a->N * a->K * 2 + (double)a->M * n_out * 2;
    crg_prof_scope ps(ctx, st, CRG_K_LNGEMM, flops, bytes);
    hipLaunchKernelGGL(kern, dim3((a->M + 127) / 128), dim3(512), lds_pp, st, p);
    CRG_CHECK_LAUNCH(ctx, "ln_gemm");
    return 0;
  }
  if (geglu) { kern = lngemm_kernel<4, 5, true>; bn = 128; }
  else if (wide) { kern = lngemm_kernel<5, 5, false>; bn = 160; }
  else { kern = lngemm_kernel<4, 5, false>; bn = 128; }
  const size_t lds = (size_t)5 * 128 * 128 + (size_t)4 * bn * 128;
  if (int rc = crg_set_dyn_lds(ctx, reinterpret_cast<const void*>(kern), 160 * 1024, "ln_gemm")) return rc;
  const double flops = 2.0 * a->M * (double)a->N * a->K;
  const double bytes = (double)a->M * a->K * 2 + (double)a->N * a->K * 2 + (double)a->M * n_out * 2;
  crg_prof_scope ps(ctx, st, CRG_K_LNGEMM, flops, bytes);
  hipLaunchKernelGGL(kern, dim3((a->M + 127) / 128), dim3(512), lds, st, p);
  CRG_CHECK_LAUNCH(ctx, "ln_gemm");
  return 0;
}
