// Flash-style attention for gfx950: O = softmax(Q K^T * scale) V, bf16 in/out, fp32 softmax and
// accumulation, never materialising the Nq x Nk score matrix (the reference's original/sliced paths
// do: attention.py:646-658, 415-424).
//
// One block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.  Per 64-key
// tile, all on v_mfma_f32_32x32x16_bf16:
//   S^T[key][query] = K_tile (A operand, from LDS) x Q^T (B operand, registers for the whole kernel)
//     -> the query sits on the LANE and the keys in the 16 accumulator registers, so the row max /
//        row sum are register reductions plus ONE cross-half shuffle;
//   O^T[d][query]  += V^T_tile (A operand, from LDS) x P^T (B operand = the S^T accumulator itself,
//        converted pairwise to bf16: guide §3 "An accumulator tile as the next MFMA's operand"; the
//        k order inside a 16-key step is row 16s + 8(j>>2) + 4h + (j&3), so V^T is read as two 8-byte
//        pieces per step at key offsets 16s+4h and 16s+8+4h).
// V arrives either TRANSPOSED from HBM ([channel][key]; emitted by a V projection GEMM with swapped operands: crg_attention)
// or ROW-MAJOR ([key][channel], the third slice of ONE fused Q|K|V projection: crg_attention_v, VRM = true).  Row-major V is
// staged as [32-channel block][64 keys][32 channels] (64-byte rows) and the V^T fragments come out of ds_read_b64_tr_b16
// (guide T10: per 16-lane group a 4-row x 16-column block, delivered column-major): the transposed projection launch
// disappears and the LDS image is conflict-free (four 64-byte rows of a block fill one 256-byte bank row).  Head dims are padded to KS*16 (QK^T) and NV*32
// (PV) with zeros: d_head 40 -> 48/64, 80 -> 80/96, 160 -> 160/160, 64 -> 64/64.
//
// LDS: K tile [64][KS*16] with rows padded to 16*(2KS+1) bytes, V^T tile [NV*32][64] with rows padded
// to 136 bytes: both fragment read patterns are bank-conflict free (guide §2 bank rules).
#include "crg_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct AttnP {
  const bf16* q; const bf16* k; const bf16* vt; bf16* o;
  long ldq, ldk, ldvt, ldo;
  int B, H, Nq, Nk, Dh;
  float scale_log2;
};

#ifdef CRG_ATTN_LAPS
// dev instrumentation (tools/attn_phase_probe.py, build with tools/build_variant.sh laps -DCRG_ATTN_LAPS): wall_clock64 laps of
// wave 0 of the first 32 query blocks of head 0
__device__ unsigned long long crg_attn_laps[32 * 8];
#define LAP(i)                                       \
  do {                                               \
    const unsigned long long now_ = wall_clock64(); \
    lap[i] += now_ - tl;                             \
    tl = now_;                                       \
  } while (0)
#else
#define LAP(i)
#endif

constexpr int VROW = 136;  // bytes per V^T LDS row (64 keys * 2 B + 8 pad)

template <int KS, int NV, bool ONES, bool VRM = false>
__global__ __launch_bounds__(256, (KS <= 4 ? 3 : (KS <= 6 ? 2 : 1))) void attn_kernel(AttnP p) {
  constexpr int KROW = KS * 32 + 16;  // bytes per K LDS row
  constexpr int KCH = 2 * KS;         // 16-byte chunks per K row
  constexpr int KLOADS = (64 * KCH + 255) / 256;
  constexpr int KBYTES = 64 * KROW;
  constexpr int VBYTES = VRM ? NV * 4096 : NV * 32 * VROW;
  // two stages of {K tile, V^T tile}: tile t+1 is written while tile t is being read, one barrier per tile
  __shared__ __attribute__((aligned(16))) char smem[2 * (KBYTES + VBYTES)];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, hh = lane >> 5;
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2 each).  In launch order the query
  // blocks of one (batch, head) land on all 8 XCDs and every L2 streams every head's K / V^T; the remap hands each XCD a
  // contiguous run of (head, query block) pairs, so a head's K / V^T is fetched by one XCD only (-1.5 % at 4096 tokens).
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int nqb = gridDim.x, total = nqb * gridDim.y;
    if ((total & 7) == 0) {
      const int id = bx + nqb * by;
      const int L = (id & 7) * (total >> 3) + (id >> 3);
      by = L / nqb;
      bx = L - by * nqb;
    }
  }
  const int b = by / p.H, h = by % p.H;
  const int q0 = bx * 128 + wave * 32;

  const bf16* Q = p.q + (long)b * p.Nq * p.ldq + (long)h * p.Dh;
  const bf16* K = p.k + (long)b * p.Nk * p.ldk + (long)h * p.Dh;
  const bf16* VT = VRM ? p.vt + (long)b * p.Nk * p.ldvt + (long)h * p.Dh : p.vt + ((long)b * p.H + h) * p.Dh * p.ldvt;
  bf16* O = p.o + (long)b * p.Nq * p.ldo + (long)h * p.Dh;

  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones8 = {one, one, one, one, one, one, one, one};
  // Row-sum trick: when the padded head dim has a spare V^T row (Dh < NV*32), row Dh of the V^T tile is
  // all ones, so O^T[Dh][q] accumulates sum_k P[q][k] on the matrix core: the softmax denominator costs
  // no VALU adds.  (Masked keys have P = 0 exactly, so the ones may cover them too.)  ONES == (Dh < NV*32), chosen by
  // the launcher so that the row-sum adds are compiled out of the VALU-bound loop.
  constexpr bool ones_row = ONES;

  // ---- Q fragments (B operand): lane (r, hh) holds Q[q0 + r][16 s + 8 hh + j] ----
  bf16x8 qf[KS];
  {
    const int query = q0 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d0 = 16 * s + 8 * hh;
      qf[s] = (query < p.Nq && d0 < p.Dh) ? *reinterpret_cast<const bf16x8*>(Q + (long)query * p.ldq + d0) : zero8;
    }
  }

  f32x16 oacc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // m_run in scaled (log2) units

  bf16x8 kreg[KLOADS], vreg[NV];
  // per-lane source pointers of the K / V^T chunks this lane stages (tile-invariant part), advanced by 64 keys per tile
  const bf16* kptr[KLOADS];
  bool kuse[KLOADS];
#pragma unroll
  for (int i = 0; i < KLOADS; ++i) {
    const int idx = t + 256 * i;
    const int row = idx / KCH, c = idx - row * KCH;
    kuse[i] = idx < 64 * KCH && c * 8 < p.Dh;
    kptr[i] = K + (long)row * p.ldk + c * 8;
  }
  const bf16* vptr[NV];
  int vmode[NV];  // 0 = zero row, 1 = data row, 2 = all-ones row
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = t + 256 * i;
    if constexpr (VRM) {  // chunk slot (key, c): 8 channels 8c .. 8c + 7 of one key; slots past Dh are constants (set once below)
      const int key = idx / (4 * NV), c = idx - key * (4 * NV);
      vmode[i] = c * 8 < p.Dh ? 1 : 0;
      vptr[i] = VT + (long)key * p.ldvt + c * 8;
    } else {
      const int row = idx >> 3, c = idx & 7;
      vmode[i] = row < p.Dh ? 1 : ((ones_row && row == p.Dh) ? 2 : 0);
      vptr[i] = VT + (long)row * p.ldvt + c * 8;
    }
  }
  if constexpr (VRM) {
    // constant part of both V stages: zeros, and 1.0 in channel Dh of every key (the softmax-denominator column)
    uint4* z = reinterpret_cast<uint4*>(smem);
    for (int i = t; i < 2 * (KBYTES + VBYTES) / 16; i += 256) z[i] = uint4{0, 0, 0, 0};
    __syncthreads();
    if (ones_row && t < 128) {
      const int st = t >> 6, key = t & 63;
      const int blk = p.Dh >> 5, col = p.Dh & 31;
      *reinterpret_cast<bf16*>(smem + st * (KBYTES + VBYTES) + KBYTES + blk * 4096 + key * 64 + col * 2) = one;
    }
    __syncthreads();
  }
  auto prefetch = [&](int tile) {
    const int kbase = tile * 64;
    if (kbase + 64 <= p.Nk) {  // interior tile (block-uniform): no per-key bounds checks
#pragma unroll
      for (int i = 0; i < KLOADS; ++i) kreg[i] = kuse[i] ? *reinterpret_cast<const bf16x8*>(kptr[i] + (long)kbase * p.ldk) : zero8;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if constexpr (VRM) vreg[i] = vmode[i] == 1 ? *reinterpret_cast<const bf16x8*>(vptr[i] + (long)kbase * p.ldvt) : zero8;
        else vreg[i] = vmode[i] == 1 ? *reinterpret_cast<const bf16x8*>(vptr[i] + kbase) : (vmode[i] == 2 ? ones8 : zero8);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < KLOADS; ++i) {
      const int row = (t + 256 * i) / KCH;
      kreg[i] = (kuse[i] && kbase + row < p.Nk) ? *reinterpret_cast<const bf16x8*>(kptr[i] + (long)kbase * p.ldk) : zero8;
    }
    if constexpr (VRM) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int key = (t + 256 * i) / (4 * NV);
        vreg[i] = (vmode[i] == 1 && kbase + key < p.Nk) ? *reinterpret_cast<const bf16x8*>(vptr[i] + (long)kbase * p.ldvt) : zero8;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int key0 = kbase + ((t + 256 * i) & 7) * 8;
      bf16x8 v = zero8;
      if (vmode[i] == 1 && key0 < p.Nk) {
        v = *reinterpret_cast<const bf16x8*>(vptr[i] + kbase);
        if (key0 + 8 > p.Nk) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (key0 + e >= p.Nk) v[e] = (bf16)0.f;
        }
      } else if (vmode[i] == 2) {
        v = ones8;
      }
      vreg[i] = v;
    }
  };
  auto commit = [&](int buf) {
    char* Ks = smem + buf * (KBYTES + VBYTES);
    char* Vs = Ks + KBYTES;
#pragma unroll
    for (int i = 0; i < KLOADS; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / KCH, c = idx - row * KCH;
      if (idx < 64 * KCH) *reinterpret_cast<bf16x8*>(Ks + row * KROW + c * 16) = kreg[i];
    }
    if constexpr (VRM) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = t + 256 * i;
        const int key = idx / (4 * NV), c = idx - key * (4 * NV);
        if (vmode[i] == 1) *reinterpret_cast<bf16x8*>(Vs + (c >> 2) * 4096 + key * 64 + (c & 3) * 16) = vreg[i];  // constant slots stay
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      const int row = idx >> 3, c = idx & 7;
      const uint2* src = reinterpret_cast<const uint2*>(&vreg[i]);
      uint2* dst = reinterpret_cast<uint2*>(Vs + row * VROW + c * 16);  // rows are only 8-byte aligned
      dst[0] = src[0];
      dst[1] = src[1];
    }
  };

  const int ntiles = (p.Nk + 63) / 64;
#ifdef CRG_ATTN_LAPS
  unsigned long long lap[5] = {0, 0, 0, 0, 0}, tl = wall_clock64();
#endif
  prefetch(0);
  commit(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool more = tile + 1 < ntiles;
    LAP(4);
    if (more) prefetch(tile + 1);
    LAP(0);
    const char* Ks = smem + (tile & 1) * (KBYTES + VBYTES);
    const char* Vs = Ks + KBYTES;

    // ---- S^T = K Q^T ----
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * KROW + (2 * s + hh) * 16);
        st[kb] = CRG_MFMA_32x32x16(kf, qf[s], st[kb]);
      }
    }
    LAP(1);
    // ---- online softmax in base 2: p = exp2(s * scale_log2 - m); the max is taken on the raw scores ----
    const bool last = (tile + 1 == ntiles) && (p.Nk & 63);
    if (last) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = tile * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= p.Nk) st[kb][e] = -INFINITY;
        }
    }
    // 32 scores -> 16 v_max3_f32: max3(a, b, c) chains, two new values per instruction
    float mx = fmaxf(fmaxf(st[0][0], st[1][0]), st[0][1]);
#pragma unroll
    for (int e = 1; e < 15; ++e) mx = fmaxf(fmaxf(mx, st[1][e]), st[0][e + 1]);
    mx = fmaxf(mx, st[1][15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2;
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new != m_run)) {  // wave-uniform: rescale only when some row's running max moved (alpha == 1 otherwise)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
      m_run = m_new;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      st[kb] = st[kb] * p.scale_log2 - m_run;  // whole-vector form: contracts to v_pk_fma_f32 (two scores per instruction)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(st[kb][e]);
        st[kb][e] = pe;
        if (!ones_row) rs += pe;
      }
    }
    if (!ones_row) l_run += rs;
    // ---- P^T fragments straight from the accumulator registers ----
    bf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kb][s2][j] = (bf16)st[kb][8 * s2 + j];
    LAP(2);
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int dv = 0; dv < NV; ++dv) {
      const char* vrow = Vs + (dv * 32 + r) * VROW;
      // VRM: transposed read.  16-lane group g = lane / 16 takes channels 16 (g & 1) .. + 15 of keys 4 (g >> 1) .. + 3 (plus the
      // k-step's base key); lane 4 q + pp of the group supplies the address of key row q, channels 4 pp .. 4 pp + 3, and
      // receives its own channel of the four keys - the k order 16 s + 8 (j >> 2) + 4 hh + (j & 3) the P fragment has.
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) s16x4* trptr_t;
      const char* vtr = Vs + dv * 4096 + (4 * (lane >> 5) + ((lane >> 2) & 3)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 vf;
          if constexpr (VRM) {
            const char* a0 = vtr + (kb * 32 + 16 * s2) * 64;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)a0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(a0 + 8 * 64));
            const short v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            __builtin_memcpy(&vf, v8, 16);
          } else {
            const int keyoff = kb * 32 + 16 * s2 + 4 * hh;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + keyoff * 2);
            const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + (keyoff + 8) * 2);
            vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
          oacc[dv] = CRG_MFMA_32x32x16(vf, pf[kb][s2], oacc[dv]);
        }
    }
    LAP(3);
    if (more) commit((tile + 1) & 1);
    __syncthreads();
  }
#ifdef CRG_ATTN_LAPS
  if (by == 0 && bx < 32 && t == 0) {
    LAP(4);
    for (int i = 0; i < 5; ++i) crg_attn_laps[bx * 8 + i] = lap[i];
  }
#endif

  // ---- normalise and store: lane = query, registers = channels (4 consecutive per group) ----
  float l_tot;
  if (ones_row) {
    // O^T row Dh sits in tile Dh/32 at register 4*((Dh%32)/8) of the hh == 0 half (Dh % 8 == 0)
    const int dvl = p.Dh >> 5, regl = ((p.Dh & 31) >> 3) * 4;
    float v = 0.f;
#pragma unroll
    for (int dv = 0; dv < NV; ++dv)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (dv == dvl && 4 * g == regl) v = oacc[dv][4 * g];
    const float other = __shfl_xor(v, 32);
    l_tot = hh == 0 ? v : other;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  const float inv = 1.0f / l_tot;
  const int query = q0 + r;
  if (query < p.Nq) {
#pragma unroll
    for (int dv = 0; dv < NV; ++dv)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dv * 32 + 8 * g + 4 * hh;
        if (d < p.Dh) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (bf16)(oacc[dv][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(O + (long)query * p.ldo + d) = o4;
        }
      }
  }
}

// ---- few keys (Nk <= 128): the cross-attentions against the 77-token prompt context, FaceID's four tokens, 8x8 self-attention ----
// Same mathematics, MFMA layout and LDS images as attn_kernel, different loop nest.  With at most two 64-key tiles the whole K / V of a
// (batch, head) fits the two LDS stages, and what attn_kernel spends its time on at these sizes is per-BLOCK cost - zeroing and
// staging the tiles, three block barriers, the launch of 2048 blocks for 32768 x 8 query rows - not arithmetic (26-29 us for a
// 4096-query cross-attention whose MFMAs take 3).  Here a block stages the K / V tiles ONCE and then walks several 128-query
// groups of its (batch, head) with no barrier at all: Q fragments from HBM, S^T = K Q^T, online softmax over the (one or two) tiles,
// O^T += V^T P^T, normalise, store.  The grid is (groups-per-block divisor, B * H), sized to at most four blocks per CU.
template <int KS, int NV, bool ONES, bool VRM>
__global__ __launch_bounds__(256, (KS <= 4 ? 3 : (KS <= 6 ? 2 : 1))) void attn_ctx_kernel(AttnP p) {
  constexpr int KROW = KS * 32 + 16;
  constexpr int KCH = 2 * KS;
  constexpr int KLOADS = (64 * KCH + 255) / 256;
  constexpr int KBYTES = 64 * KROW;
  constexpr int VBYTES = VRM ? NV * 4096 : NV * 32 * VROW;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KBYTES + VBYTES)];  // stage = key tile (at most two)

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int by = blockIdx.y;
  const int b = by / p.H, h = by % p.H;
  const bf16* Q = p.q + (long)b * p.Nq * p.ldq + (long)h * p.Dh;
  const bf16* K = p.k + (long)b * p.Nk * p.ldk + (long)h * p.Dh;
  const bf16* VT = VRM ? p.vt + (long)b * p.Nk * p.ldvt + (long)h * p.Dh : p.vt + ((long)b * p.H + h) * p.Dh * p.ldvt;
  bf16* O = p.o + (long)b * p.Nq * p.ldo + (long)h * p.Dh;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones8 = {one, one, one, one, one, one, one, one};
  constexpr bool ones_row = ONES;  // softmax denominator on the matrix core (see attn_kernel)
  const int ntiles = (p.Nk + 63) >> 6;  // 1 or 2 (launcher)

  // ---- stage every key tile once (bounds-checked form: tails are the normal case here) ----
  if constexpr (VRM) {
    uint4* z = reinterpret_cast<uint4*>(smem);
    for (int i = t; i < 2 * (KBYTES + VBYTES) / 16; i += 256) z[i] = uint4{0, 0, 0, 0};
    __syncthreads();
    if (ones_row && t < 128) {
      const int st = t >> 6, key = t & 63;
      const int blk = p.Dh >> 5, col = p.Dh & 31;
      *reinterpret_cast<bf16*>(smem + st * (KBYTES + VBYTES) + KBYTES + blk * 4096 + key * 64 + col * 2) = one;
    }
    __syncthreads();
  }
  for (int tile = 0; tile < ntiles; ++tile) {
    const int kbase = tile * 64;
    char* Ks = smem + tile * (KBYTES + VBYTES);
    char* Vs = Ks + KBYTES;
#pragma unroll
    for (int i = 0; i < KLOADS; ++i) {
      const int idx = t + 256 * i;
      const int row = idx / KCH, c = idx - row * KCH;
      if (idx < 64 * KCH) {
        const bool ok = c * 8 < p.Dh && kbase + row < p.Nk;
        *reinterpret_cast<bf16x8*>(Ks + row * KROW + c * 16) = ok ? *reinterpret_cast<const bf16x8*>(K + (long)(kbase + row) * p.ldk + c * 8) : zero8;
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      if constexpr (VRM) {  // chunk slot (key, c): channels 8c .. 8c + 7 of one key; constant slots (zeros, the ones column) stay
        const int key = idx / (4 * NV), c = idx - key * (4 * NV);
        if (c * 8 < p.Dh) {
          const bf16x8 v = kbase + key < p.Nk ? *reinterpret_cast<const bf16x8*>(VT + (long)(kbase + key) * p.ldvt + c * 8) : zero8;
          *reinterpret_cast<bf16x8*>(Vs + (c >> 2) * 4096 + key * 64 + (c & 3) * 16) = v;
        }
      } else {
        const int row = idx >> 3, c = idx & 7;
        const int key0 = kbase + c * 8;
        bf16x8 v = zero8;
        if (row < p.Dh && key0 < p.Nk) {
          v = *reinterpret_cast<const bf16x8*>(VT + (long)row * p.ldvt + key0);
          if (key0 + 8 > p.Nk) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (key0 + e >= p.Nk) v[e] = (bf16)0.f;
          }
        } else if (ones_row && row == p.Dh) {
          v = ones8;
        }
        const uint2* src = reinterpret_cast<const uint2*>(&v);
        uint2* dst = reinterpret_cast<uint2*>(Vs + row * VROW + c * 16);  // rows are only 8-byte aligned
        dst[0] = src[0];
        dst[1] = src[1];
      }
    }
  }
  __syncthreads();

  const int nqg = (p.Nq + 127) >> 7;
  // Q fragments of the NEXT group are requested before the current group is multiplied (a wave walks its groups back to back:
  // without the prefetch every group starts with an exposed HBM round trip)
  auto load_q = [&](bf16x8 (&dst)[KS], int qg) {
    const int qq = qg * 128 + wave * 32 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d0 = 16 * s + 8 * hh;
      dst[s] = (qg < nqg && qq < p.Nq && d0 < p.Dh) ? *reinterpret_cast<const bf16x8*>(Q + (long)qq * p.ldq + d0) : zero8;
    }
  };
  bf16x8 qf[KS], qn[KS];
  load_q(qn, blockIdx.x);
  for (int qg = blockIdx.x; qg < nqg; qg += gridDim.x) {
    const int query = qg * 128 + wave * 32 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qn[s];
    load_q(qn, qg + gridDim.x);
    f32x16 oacc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    for (int tile = 0; tile < ntiles; ++tile) {
      const char* Ks = smem + tile * (KBYTES + VBYTES);
      const char* Vs = Ks + KBYTES;
      f32x16 st[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + r) * KROW + (2 * s + hh) * 16);
          st[kb] = CRG_MFMA_32x32x16(kf, qf[s], st[kb]);
        }
      }
      if ((tile + 1 == ntiles) && (p.Nk & 63)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = tile * 64 + kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (key >= p.Nk) st[kb][e] = -INFINITY;
          }
      }
      float mx = fmaxf(fmaxf(st[0][0], st[1][0]), st[0][1]);
#pragma unroll
      for (int e = 1; e < 15; ++e) mx = fmaxf(fmaxf(mx, st[1][e]), st[0][e + 1]);
      mx = fmaxf(mx, st[1][15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2;
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
        m_run = m_new;
      }
      float rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        st[kb] = st[kb] * p.scale_log2 - m_run;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = __builtin_amdgcn_exp2f(st[kb][e]);
          st[kb][e] = pe;
          if (!ones_row) rs += pe;
        }
      }
      if (!ones_row) l_run += rs;
      bf16x8 pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[kb][s2][j] = (bf16)st[kb][8 * s2 + j];
#pragma unroll
      for (int dv = 0; dv < NV; ++dv) {
        const char* vrow = Vs + (dv * 32 + r) * VROW;
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* trptr_t;
        const char* vtr = Vs + dv * 4096 + (4 * (lane >> 5) + ((lane >> 2) & 3)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 vf;
            if constexpr (VRM) {
              const char* a0 = vtr + (kb * 32 + 16 * s2) * 64;
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)a0);
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(a0 + 8 * 64));
              const short v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
              __builtin_memcpy(&vf, v8, 16);
            } else {
              const int keyoff = kb * 32 + 16 * s2 + 4 * hh;
              const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + keyoff * 2);
              const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + (keyoff + 8) * 2);
              vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            oacc[dv] = CRG_MFMA_32x32x16(vf, pf[kb][s2], oacc[dv]);
          }
      }
    }
    float l_tot;
    if (ones_row) {
      const int dvl = p.Dh >> 5, regl = ((p.Dh & 31) >> 3) * 4;
      float v = 0.f;
#pragma unroll
      for (int dv = 0; dv < NV; ++dv)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (dv == dvl && 4 * g == regl) v = oacc[dv][4 * g];
      const float other = __shfl_xor(v, 32);
      l_tot = hh == 0 ? v : other;
    } else {
      l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = 1.0f / l_tot;
    if (query < p.Nq) {
#pragma unroll
      for (int dv = 0; dv < NV; ++dv)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dv * 32 + 8 * g + 4 * hh;
          if (d < p.Dh) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (bf16)(oacc[dv][4 * g + e] * inv);
            *reinterpret_cast<bf16x4*>(O + (long)query * p.ldo + d) = o4;
          }
        }
    }
  }
}

// ---- LDS-DMA form (Nk % 64 == 0, V^T from HBM) ---------------------------------------------------------------------------
// Same mathematics and MFMA layout as attn_kernel, but the K / V^T tiles go HBM -> LDS by buffer_load ... lds: no staging
// registers, no ds_write commit, no vmcnt(0) in front of a store (115 instead of 152 VGPRs at d_head 40).  The SIMD's vector issue
// port is what the loop runs out of (every v_exp_f32 holds it for 8 cycles, an MFMA for 8, an LDS-DMA piece for 60-180:
// MI355X_MICROARCH.md constants table), so the tile is shared by NW = 8 waves (256 queries per block): the DMA pieces per wave and
// tile drop from 2.75 to 1.25.
// The DMA writes a wave-instruction's 64 x 16 B linearly, so the LDS images are unpadded and the per-lane SOURCE picks the layout:
//  * K tile [64 rows][KC = Dh / 8 chunks]: LDS row rho holds key pi(rho) = rho with bits 2 and 3 swapped.  The S^T accumulator then
//    has, in a lane's registers 8 s2 .. 8 s2 + 7, the keys 16 s2 + 8 hh .. + 7 IN ORDER, so the V^T fragment of a k-step is ONE
//    16-byte chunk (ds_read_b128) instead of two 8-byte pieces.  Chunk c of row rho sits at position c ^ kswz(rho), chosen per KC
//    so that the 16 rows of every ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31} of each half) hit 16 different
//    16-byte slots (odd KC: none needed).  With Dh = 8 (mod 16) the last k-step's upper chunk does not exist: those lanes re-read
//    chunk KC - 1 (finite values against zeros in the Q fragment).
//  * V^T tile [Dh rows][8 chunks] + one all-ones row + one zero row per stage (written once; padded rows of the A operand are
//    redirected to them): chunk c of row d at position c ^ ((d >> 1) & 7), conflict-free for the same lane groups.
// Rows past Nk: the buffer descriptor's range check returns zeros.
// max(v[lane], v[lane ^ 32]) without the LDS crossbar: v_permlane32_swap exchanges the upper half of one register with the lower
// half of another in one VALU instruction.  (__shfl_xor is a ds_bpermute: its lgkmcnt(0) wait also drains the fragment reads that
// were issued ahead for the next MFMAs.)
static __device__ __forceinline__ float attn_max_halves(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
  return v;
#endif
}
typedef __attribute__((address_space(3))) void* lptr_t;
static __device__ __forceinline__ void attn_dma16(const void* base, unsigned bytes, char* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
#endif
}
static __device__ __forceinline__ void attn_wait_vm0() { __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8)); }

template <int KC>
struct AttnKSwz {  // XOR span (chunks) and the per-row swizzle of the K image
  static constexpr int SPAN = KC % 2 == 1 ? 1 : (KC % 4 == 2 ? 2 : (KC % 8 == 4 ? 4 : (KC % 16 == 8 ? 8 : 16)));
  static __device__ __host__ __forceinline__ int of(int rho) {
    if constexpr (SPAN == 1) return 0;
    else if constexpr (SPAN == 2) return (rho >> 3) & 1;
    else if constexpr (SPAN == 4) return ((rho >> 2) & 1) | (((rho >> 4) & 1) << 1);
    else if constexpr (SPAN == 8) return (rho >> 1) & 7;
    else return rho & 15;
  }
};

template <int KC, int NV, bool ONES, int OCC, int SUB, int NW>
__global__ __launch_bounds__(64 * NW, OCC) void attn_dma_kernel(AttnP p, unsigned k_bytes, unsigned v_bytes) {
  constexpr int KS = (KC + 1) / 2;     // 16-channel k-steps of Q K^T
  constexpr int SPAN = AttnKSwz<KC>::SPAN;
  constexpr int KBYTES = 64 * KC * 16;
  constexpr int VBYTES = (KC * 8 + 2) * 128;  // Dh data rows + the ones row + the zero row
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int KI = KC, VI = KC;             // wave-instructions (1 KiB pieces) per tile: K 64 x KC chunks, V^T 8 KC x 8 chunks
  constexpr int NI = KI + VI;
  constexpr int NQ = (NI + NW - 1) / NW;      // DMA slots per wave
  constexpr int TSTAGE = SUB * STAGE;         // SUB 64-key tiles per ring stage: one barrier per SUB tiles
  constexpr int Dh = KC * 8;
  __shared__ __attribute__((aligned(256))) char smem[2 * TSTAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int nqb = gridDim.x, total = nqb * gridDim.y;
    if ((total & 7) == 0) {  // XCD-aware order, as attn_kernel
      const int id = bx + nqb * by;
      const int L = (id & 7) * (total >> 3) + (id >> 3);
      by = L / nqb;
      bx = L - by * nqb;
    }
  }
  const int b = by / p.H, h = by % p.H;
  const int q0 = bx * (32 * NW) + wave * 32;
  const bf16* Q = p.q + (long)b * p.Nq * p.ldq + (long)h * Dh;
  const bf16* K = p.k + (long)b * p.Nk * p.ldk + (long)h * Dh;
  const bf16* VT = p.vt + ((long)b * p.H + h) * Dh * p.ldvt;
  bf16* O = p.o + (long)b * p.Nq * p.ldo + (long)h * Dh;

  // per-lane DMA sources of this wave's slots (tile-invariant; the tile advances through the scalar offset)
  int voff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = wave + NW * q;
    if (j < KI) {
      const int i = 64 * j + lane;
      const int rho = i / KC, cpos = i - rho * KC;
      const int c = cpos ^ AttnKSwz<KC>::of(rho);
      const int key = (rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1);
      voff[q] = (int)(key * p.ldk * 2) + c * 16;
    } else {
      const int i = 64 * (j - KI) + lane;
      const int d = i >> 3, cpos = i & 7;
      const int c = cpos ^ ((d >> 1) & 7);
      voff[q] = (int)(d * p.ldvt * 2) + c * 16;
    }
  }
  auto issue = [&](int super, int stage) {
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const int tile = super * SUB + sub;
      char* Ks = smem + stage * TSTAGE + sub * STAGE;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j = wave + NW * q;  // wave-uniform
        if (j < KI) attn_dma16(K, k_bytes, Ks + j * 1024, voff[q], tile * 64 * (int)p.ldk * 2);
        else if (j < NI) attn_dma16(VT, v_bytes, Ks + KBYTES + (j - KI) * 1024, voff[q], tile * 128);
      }
    }
  };

  // constant rows of every stage: row Dh = ones (softmax denominator on the matrix core), row Dh + 1 = zeros
  if (t < 32 * SUB) {
    const int st = t >> 4, row = (t >> 3) & 1, ch = t & 7;
    const bf16 one = (bf16)1.0f;
    const bf16x8 ones8 = {one, one, one, one, one, one, one, one};
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<bf16x8*>(smem + st * STAGE + KBYTES + (Dh + row) * 128 + ch * 16) = (ONES && row == 0) ? ones8 : zero8;
  }

  bf16x8 qf[KS];
  {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int query = q0 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d0 = 16 * s + 8 * hh;
      qf[s] = (query < p.Nq && d0 < Dh) ? *reinterpret_cast<const bf16x8*>(Q + (long)query * p.ldq + d0) : zero8;
    }
  }
  f32x16 oacc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // fragment addresses.  K: row r (+ 32 kb), chunk (2 s + hh) ^ kswz(r) - the XOR touches the low log2(SPAN) chunk bits only, the
  // rest of 2 s is an immediate offset.  V^T: row min(32 dv + r, Dh + 1), chunk (4 kb + 2 s2 + hh) ^ ((row >> 1) & 7).
  const int kaddr = r * (KC * 16) + ((hh ^ AttnKSwz<KC>::of(r)) << 4);
  const int kaddr_last = (KC & 1) ? r * (KC * 16) : kaddr;  // odd KC: k-step KS - 1 has chunk KC - 1 only (both halves read it)
  int vaddr[NV];
#pragma unroll
  for (int dv = 0; dv < NV; ++dv) {
    const int R = dv * 32 + r;
    const int lrow = R < Dh + 1 ? R : Dh + 1;
    vaddr[dv] = KBYTES + lrow * 128 + ((((lrow >> 1) & 7) ^ hh) << 4);
  }

  const int nsuper = (p.Nk + 64 * SUB - 1) / (64 * SUB);  // a key tail (Nk % 64, SUB == 1 only) is masked in the last tile below
#ifdef CRG_ATTN_LAPS
  unsigned long long lap[5] = {0, 0, 0, 0, 0}, tl = wall_clock64();
  const unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = tl;
#endif
  issue(0, 0);
  attn_wait_vm0();
  __syncthreads();
  for (int super = 0; super < nsuper; ++super) {
    LAP(4);
    if (super + 1 < nsuper) issue(super + 1, (super + 1) & 1);
    if (SUB == 1 && super + 1 == nsuper && (p.Nk & 63)) {
      // key tail: the V^T chunks were copied raw, so the columns Nk .. of the last tile hold the producer's row padding or the next
      // channel's keys; P is 0 there, but 0 x NaN is not - zero them in LDS (block-uniform branch, last tile only)
      char* Vs = smem + (super & 1) * TSTAGE + KBYTES;
      for (int i = t; i < Dh * 8; i += 64 * NW) {
        const int d = i >> 3, cpos = i & 7;
        const int key0 = super * 64 + (cpos ^ ((d >> 1) & 7)) * 8;
        if (key0 + 8 > p.Nk) {
          bf16x8* cp = reinterpret_cast<bf16x8*>(Vs + d * 128 + cpos * 16);
          bf16x8 v = *cp;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (key0 + e >= p.Nk) v[e] = (bf16)0.f;
          *cp = v;
        }
      }
      __syncthreads();
    }
    LAP(0);
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const char* Ks = smem + (super & 1) * TSTAGE + sub * STAGE;

      f32x16 st[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int base = ((KC & 1) && s == KS - 1) ? kaddr_last : kaddr;
          const int lo = ((2 * s) & (SPAN - 1)) << 4, hi = ((2 * s) & ~(SPAN - 1)) << 4;
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + ((base ^ lo) + hi + kb * 32 * KC * 16));
          st[kb] = CRG_MFMA_32x32x16(kf, qf[s], st[kb]);
        }
      }
      LAP(1);
      if (SUB == 1 && super + 1 == nsuper && (p.Nk & 63)) {  // block-uniform: keys past Nk (their K rows were zero-filled) leave the softmax
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            // S^T row rho = 32 kb + (e & 3) + 8 (e >> 2) + 4 hh holds key pi(rho): bits 2 and 3 swapped
            const int key = super * 64 + kb * 32 + (e & 3) + 4 * ((e >> 2) & 1) + 8 * hh + 16 * (e >> 3);
            if (key >= p.Nk) st[kb][e] = -INFINITY;
          }
      }
      float mx = fmaxf(fmaxf(st[0][0], st[1][0]), st[0][1]);
#pragma unroll
      for (int e = 1; e < 15; ++e) mx = fmaxf(fmaxf(mx, st[1][e]), st[0][e + 1]);
      mx = fmaxf(mx, st[1][15]);
      mx = attn_max_halves(mx) * p.scale_log2;
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new != m_run)) {  // wave-uniform: rescale only when some row's running max moved
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
        m_run = m_new;
      }
      float rs = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#ifdef CRG_ATTN_PKFMA
        st[kb] = st[kb] * p.scale_log2 - m_run;
#endif
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#ifdef CRG_ATTN_PKFMA
          const float pe = __builtin_amdgcn_exp2f(st[kb][e]);
#else
          // scalar v_fma_f32 on purpose: beside MFMAs a v_pk_fma_f32 costs far more issue time than the two v_fma_f32 it replaces
          // (MI355X_MICROARCH.md constants table); the opaque asm keeps the SLP vectoriser from re-pairing them
          float x = __builtin_fmaf(st[kb][e], p.scale_log2, -m_run);
#if defined(__HIP_DEVICE_COMPILE__)
          asm volatile("" : "+v"(x));
#endif
          const float pe = __builtin_amdgcn_exp2f(x);
#endif
          st[kb][e] = pe;
          if (!ONES) rs += pe;
        }
      }
      if (!ONES) l_run += rs;
      bf16x8 pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[kb][s2][j] = (bf16)st[kb][8 * s2 + j];
      LAP(2);
#if defined(CRG_ATTN_DUMMY_VALU) && defined(__HIP_DEVICE_COMPILE__)
      float dmy[4] = {0.f, 0.f, 0.f, 0.f};
      // sensitivity probe: extra independent VALU instructions per tile (does the vector issue port bound the loop?)
#pragma unroll
      for (int i = 0; i < CRG_ATTN_DUMMY_VALU; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dmy[i & 3]));
#endif
#if defined(CRG_ATTN_DUMMY_MFMA) && defined(__HIP_DEVICE_COMPILE__)
      // sensitivity probe: extra MFMAs per tile on a scratch accumulator (does the matrix pipe bound the loop?)
      {
        f32x16 dz;
#pragma unroll
        for (int e = 0; e < 16; ++e) dz[e] = 0.f;
#pragma unroll
        for (int i = 0; i < CRG_ATTN_DUMMY_MFMA; ++i) dz = CRG_MFMA_32x32x16(qf[0], pf[0][0], dz);
        asm volatile("" ::"v"(dz));
      }
#endif
#pragma unroll
      for (int dv = 0; dv < NV; ++dv) {
        const int va = vaddr[dv];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Ks + (va ^ ((kb * 4 + 2 * s2) << 4)));
            oacc[dv] = CRG_MFMA_32x32x16(vf, pf[kb][s2], oacc[dv]);
          }
      }
      LAP(3);
    }
    attn_wait_vm0();  // this wave's pieces of the next stage have landed (issued a whole stage ago)
    __syncthreads();  // ... everyone's have, and everyone is done reading this stage
  }
#ifdef CRG_ATTN_LAPS
  if (by == 0 && bx < 32 && t == 0) {
    LAP(4);
    for (int i = 0; i < 5; ++i) crg_attn_laps[bx * 8 + i] = lap[i];
    crg_attn_laps[bx * 8 + 5] = __builtin_amdgcn_s_memtime() - mt0;  // shader cycles
    crg_attn_laps[bx * 8 + 6] = wall_clock64() - rt0;               // 100 MHz ticks
  }
#endif

  float l_tot;
  if (ONES) {
    constexpr int dvl = Dh >> 5, regl = ((Dh & 31) >> 3) * 4;  // O^T row Dh: tile Dh / 32, register 4 ((Dh % 32) / 8) of the hh == 0 half
    const float v = oacc[dvl < NV ? dvl : 0][regl];
    const float other = __shfl_xor(v, 32);
    l_tot = hh == 0 ? v : other;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  const float inv = 1.0f / l_tot;
  const int query = q0 + r;
  if (query < p.Nq) {
#pragma unroll
    for (int dv = 0; dv < NV; ++dv)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dv * 32 + 8 * g + 4 * hh;
        if (d < Dh) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (bf16)(oacc[dv][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(O + (long)query * p.ldo + d) = o4;
        }
      }
  }
}

template <int KC, int NV, int OCC, int SUB, int NW>
int launch_attn_dma(crg_ctx* ctx, hipStream_t st, const AttnP& p) {
  dim3 grid((p.Nq + 32 * NW - 1) / (32 * NW), p.B * p.H);
  const unsigned k_bytes = (unsigned)((((long)p.Nk - 1) * p.ldk + p.Dh) * 2);
  const unsigned v_bytes = (unsigned)((((long)p.Dh - 1) * p.ldvt + (p.Nk + 7) / 8 * 8) * 2);  // whole 16-byte chunks (ldvt >= roundup(Nk, 8))
  hipLaunchKernelGGL((attn_dma_kernel<KC, NV, (KC * 8 < NV * 32), OCC, SUB, NW>), grid, dim3(64 * NW), 0, st, p, k_bytes, v_bytes);
  CRG_CHECK_LAUNCH(ctx, "attention (LDS-DMA)");
  return 0;
}

// ---- software-pipelined form -------------------------------------------------------------------------------------------------
// What the sensitivity builds showed (tools/attn_occ_probe.py; attn_dma_kernel and a two-group ping-pong form of it alike): per tile and wave the
// loop costs MFMA time PLUS VALU time.  Waves sharing a SIMD do not hide each other's VALU under MFMAs (moving work between them is
// zero-sum, guide "Two waves per SIMD" item 3); what does overlap is VALU issued by the SAME wave in the shadow of its own MFMA
// (24 of the 32 cycles of a v_mfma_f32_32x32x16_bf16 are free issue slots).  So one wave works on three tiles at once:
//   iteration t:  matrix pipe: O^T += V^T(t-1) P^T(t-1)  and  S^T(t+1) = K(t+1) Q^T     (14 MFMAs)
//                 vector ALU : softmax of S^T(t) -> P(t)                                 (~100 VALU, placed between those MFMAs)
// The state is parity-indexed (S^T and P of even / odd tiles) and the loop is unrolled by two, so nothing is copied.  K is staged
// two tiles ahead, V^T one: iteration t reads K(t+1) and V^T(t-1) while K(t+2) and V^T(t) land in the other stages.
template <int KC, int NV, bool ONES, int NW, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void attn_sp_kernel(AttnP p, unsigned k_bytes, unsigned v_bytes) {
  constexpr int KS = (KC + 1) / 2;
  constexpr int SPAN = AttnKSwz<KC>::SPAN;
  constexpr int KBYTES = 64 * KC * 16;
  constexpr int VBYTES = (KC * 8 + 2) * 128;
  constexpr int NI = 2 * KC;  // DMA pieces per iteration: K(t+2) and V^T(t)
  constexpr int NQ = (NI + NW - 1) / NW;
  constexpr int Dh = KC * 8;
  // Dh = 8 (mod 16): the last k-step of Q K^T has eight padded channels.  They carry the softmax's shift: Q is pre-multiplied by
  // scale * log2(e) when its fragments are loaded and its channels Dh, Dh + 1 hold -m_ref (as hi + lo), K's channels Dh.. read a chunk of ones, so the MFMA
  // delivers s * scale * log2(e) - m_ref and P = exp2 of it directly - no v_fma per score.  m_ref (per query, exactly hi + lo) follows the
  // row maximum lazily: it moves only when a tile's maximum exceeds it by more than QPAD_T (P then stays below 2^QPAD_T), which
  // costs that tile one subtraction per score and the accumulators one rescale; always on the first tile.
#ifdef CRG_ATTN_NO_QPAD
  constexpr bool QPAD = false;
#else
  constexpr bool QPAD = (KC & 1) && ONES;
#endif
  constexpr float QPAD_T = 10.0f;
  __shared__ __attribute__((aligned(256))) char smem[2 * KBYTES + 2 * VBYTES];  // [K0][K1][V0][V1]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31, hh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int nqb = gridDim.x, total = nqb * gridDim.y;
    if ((total & 7) == 0) {
      const int id = bx + nqb * by;
      const int L = (id & 7) * (total >> 3) + (id >> 3);
      by = L / nqb;
      bx = L - by * nqb;
    }
  }
  const int b = by / p.H, h = by % p.H;
  const int q0 = bx * (32 * NW) + wave * 32;
  const bf16* Q = p.q + (long)b * p.Nq * p.ldq + (long)h * Dh;
  const bf16* K = p.k + (long)b * p.Nk * p.ldk + (long)h * Dh;
  const bf16* VT = p.vt + ((long)b * p.H + h) * Dh * p.ldvt;
  bf16* O = p.o + (long)b * p.Nq * p.ldo + (long)h * Dh;

  int voff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = wave + NW * q;
    if (j < KC) {
      const int i = 64 * j + lane;
      const int rho = i / KC, cpos = i - rho * KC;
      const int c = cpos ^ AttnKSwz<KC>::of(rho);
      const int key = (rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1);
      voff[q] = (int)(key * p.ldk * 2) + c * 16;
    } else {
      const int i = 64 * (j - KC) + lane;
      const int d = i >> 3, cpos = i & 7;
      const int c = cpos ^ ((d >> 1) & 7);
      voff[q] = (int)(d * p.ldvt * 2) + c * 16;
    }
  }
  const int T = p.Nk >> 6;  // even (launcher)
  // K tile kt -> K stage kt & 1, V^T tile vt -> V stage vt & 1 (tiles past the end: the range check writes zeros, nobody reads them)
  auto issue = [&](int kt, int vt) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int j = wave + NW * q;  // wave-uniform
      if (j < KC) {
        if (kt >= 0) attn_dma16(K, k_bytes, smem + (kt & 1) * KBYTES + j * 1024, voff[q], kt * 64 * (int)p.ldk * 2);
      } else if (j < NI) {
        if (vt >= 0) attn_dma16(VT, v_bytes, smem + 2 * KBYTES + (vt & 1) * VBYTES + (j - KC) * 1024, voff[q], vt * 128);
      }
    }
  };

  if (t < 32) {
    const int st = t >> 4, row = (t >> 3) & 1, ch = t & 7;
    const bf16 one = (bf16)1.0f;
    const bf16x8 ones8 = {one, one, one, one, one, one, one, one};
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<bf16x8*>(smem + 2 * KBYTES + st * VBYTES + (Dh + row) * 128 + ch * 16) = (ONES && row == 0) ? ones8 : zero8;
  }

  bf16x8 qf[KS];
  {
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int query = q0 + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d0 = 16 * s + 8 * hh;
      qf[s] = (query < p.Nq && d0 < Dh) ? *reinterpret_cast<const bf16x8*>(Q + (long)query * p.ldq + d0) : zero8;
      if (QPAD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = (bf16)((float)qf[s][j] * p.scale_log2);
      }
    }
  }
  float m_ref = 0.f;  // QPAD: the shift currently stored (negated) in channel Dh of the Q fragments
  f32x16 oacc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 st[2][2];    // [tile parity][key half]: scores of tile t (parity t & 1)
  bf16x8 pf[2][2][2];  // [tile parity][key half][k-step]: P^T fragments of tile t

  const int kaddr = r * (KC * 16) + ((hh ^ AttnKSwz<KC>::of(r)) << 4);
  const int kaddr_last = (KC & 1) ? r * (KC * 16) : kaddr;
  const int ones_addr = 2 * KBYTES + Dh * 128;  // V^T stage 0, row Dh: 64 x 1.0 (written once, never overwritten by the DMA)
  int vaddr[NV];
#pragma unroll
  for (int dv = 0; dv < NV; ++dv) {
    const int R = dv * 32 + r;
    const int lrow = R < Dh + 1 ? R : Dh + 1;
    vaddr[dv] = 2 * KBYTES + lrow * 128 + ((((lrow >> 1) & 7) ^ hh) << 4);
  }

  auto qk = [&](int kt, f32x16 (&dst)[2]) {  // S^T(kt) = K(kt) Q^T (prologue only)
    const char* Ks = smem + (kt & 1) * KBYTES;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) dst[kb][e] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int base = ((KC & 1) && s == KS - 1) ? kaddr_last : kaddr;
        const int lo = ((2 * s) & (SPAN - 1)) << 4, hi = ((2 * s) & ~(SPAN - 1)) << 4;
        const char* ka = Ks + ((base ^ lo) + hi + kb * 32 * KC * 16);
        if (QPAD && s == KS - 1) ka = hh ? smem + ones_addr : ka;
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ka);
        dst[kb] = CRG_MFMA_32x32x16(kf, qf[s], dst[kb]);
      }
    }
  };
  auto pv = [&](int vt, const bf16x8 (&pp)[2][2]) {  // O^T += V^T(vt) P^T(vt) (epilogue only)
    const char* Vs = smem + (vt & 1) * VBYTES;
#pragma unroll
    for (int dv = 0; dv < NV; ++dv) {
      const int va = vaddr[dv];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (va ^ ((kb * 4 + 2 * s2) << 4)));
          oacc[dv] = CRG_MFMA_32x32x16(vf, pp[kb][s2], oacc[dv]);
        }
    }
  };
  auto tile_end = [&]() {
    attn_wait_vm0();
    __syncthreads();
  };
  // Iteration tt (parity C = tt & 1), hand-placed: every fragment of the iteration is read first (the stages became valid at the
  // barrier; the reads' latency passes under the row-max chain), then each MFMA is followed by its share of the tile's softmax -
  // G gaps for 21 work items (4 partial row maxima, the cross-half reduction + new running max, 16 x {1 pk_fma, 2 exp, 1 cvt_pk}) -
  // and a sched_barrier pins every gap (the compiler otherwise regroups MFMAs and VALU into two blocks).
  auto step = [&](int tt, auto c_c, auto pv_c, auto qk_c) {
    constexpr int C = decltype(c_c)::value;
    constexpr bool PV = decltype(pv_c)::value, QK = decltype(qk_c)::value;
    constexpr int NPV = PV ? 4 * NV : 0, NQK = QK ? 2 * KS : 0, G = NPV + NQK;
    constexpr int ITEMS = 21;
    __builtin_amdgcn_sched_barrier(0);
    issue(tt + 2, tt);
    // fragment of gap g: V^T for the PV MFMAs (g < NPV), K for the QK^T MFMAs; read D gaps ahead of its MFMA (the stages became
    // valid at the barrier, so the first D reads lead the loop and their latency passes under the row-max chain)
#ifndef CRG_ATTN_SP_DEPTH
#define CRG_ATTN_SP_DEPTH 2
#endif
    constexpr int D = CRG_ATTN_SP_DEPTH;
    bf16x8 fr[G];
    const char* Vs = smem + ((tt - 1) & 1) * VBYTES;
    const char* Ks = smem + ((tt + 1) & 1) * KBYTES;
    auto fetch = [&](int g) {
      if (PV && g < NPV) {
        fr[g] = *reinterpret_cast<const bf16x8*>(Vs + (vaddr[g >> 2] ^ ((((g >> 1) & 1) * 4 + 2 * (g & 1)) << 4)));
      } else {
        const int i = g - NPV, kb = i / KS, s = i - kb * KS;
        const int base = ((KC & 1) && s == KS - 1) ? kaddr_last : kaddr;
        const int lo = ((2 * s) & (SPAN - 1)) << 4, hi = ((2 * s) & ~(SPAN - 1)) << 4;
        const char* ka = Ks + ((base ^ lo) + hi + kb * 32 * KC * 16);
        if (QPAD && s == KS - 1) ka = hh ? smem + ones_addr : ka;
        fr[g] = *reinterpret_cast<const bf16x8*>(ka);
      }
    };
#pragma unroll
    for (int g = 0; g < (D < G ? D : G); ++g) fetch(g);
    __builtin_amdgcn_sched_barrier(0);
    float mx = -INFINITY, m_new = 0.f, alpha = 1.f, rs = 0.f;
    f32x16(&sc)[2] = st[C];
    auto item = [&](int it) {
      if (it < 4) {  // partial row maximum over 8 scores
        const int kb = it >> 1, e0 = 8 * (it & 1);
        mx = fmaxf(fmaxf(mx, sc[kb][e0]), sc[kb][e0 + 1]);
        mx = fmaxf(fmaxf(mx, sc[kb][e0 + 2]), sc[kb][e0 + 3]);
        mx = fmaxf(fmaxf(mx, sc[kb][e0 + 4]), sc[kb][e0 + 5]);
        mx = fmaxf(fmaxf(mx, sc[kb][e0 + 6]), sc[kb][e0 + 7]);
      } else if (it == 4) {
        if constexpr (QPAD) {
          mx = attn_max_halves(mx);  // already in exp2 units relative to m_ref
          const bool move = !PV || mx > QPAD_T;  // !PV: the first tile sets the reference
          if (__any(move)) {                    // wave-uniform, rare after the first tiles
            // the new reference as TWO bf16 values (channels Dh and Dh + 1: hi + lo, 16 significant bits): with one, its rounding
            // error (2^-9 of a large logit) could put the tile's maximum far above or below zero
            const float want = m_ref + mx;
            const bf16 r_hi = (bf16)want;
            const bf16 r_lo = (bf16)(want - (float)r_hi);
            const float ref_new = move ? (float)r_hi + (float)r_lo : m_ref;  // exact in fp32
            const float delta = ref_new - m_ref;
            m_ref = ref_new;
            alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
              for (int e = 0; e < 16; ++e) sc[kb][e] -= delta;
            // the shift of every LATER tile (its last k-step has not been issued yet in this iteration): Q channels Dh, Dh + 1 of the
            // upper half-wave, against the ones chunk of K
            if (move && hh) {
              qf[KS - 1][0] = (bf16)(-(float)r_hi);
              qf[KS - 1][1] = (bf16)(-(float)r_lo);
            }
          }
        } else {
          mx = attn_max_halves(mx) * p.scale_log2;
          m_new = fmaxf(m_run, mx);
          alpha = __builtin_amdgcn_exp2f(m_run - m_new);
          m_run = m_new;
        }
      } else {
        const int j = it - 5, kb = j >> 3, e = 2 * (j & 7);
        float x0, x1;
        if constexpr (QPAD) {
          x0 = sc[kb][e];
          x1 = sc[kb][e + 1];
        } else {
          x0 = __builtin_fmaf(sc[kb][e], p.scale_log2, -m_new);
          x1 = __builtin_fmaf(sc[kb][e + 1], p.scale_log2, -m_new);
        }
        const float p0 = __builtin_amdgcn_exp2f(x0), p1 = __builtin_amdgcn_exp2f(x1);
        if (!ONES) rs += p0 + p1;
        pf[C][kb][e >> 3][e & 7] = (bf16)p0;
        pf[C][kb][e >> 3][(e & 7) + 1] = (bf16)p1;
      }
    };
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g + D < G) fetch(g + D);
      if (PV && g < NPV) {
        oacc[g >> 2] = CRG_MFMA_32x32x16(fr[g], pf[1 - C][(g >> 1) & 1][g & 1], oacc[g >> 2]);
      } else {
        const int i = g - NPV, kb = i / KS, s = i - kb * KS;
        if (s == 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) st[1 - C][kb][e] = 0.f;
        }
        st[1 - C][kb] = CRG_MFMA_32x32x16(fr[g], qf[s], st[1 - C][kb]);
      }
#pragma unroll
      for (int it = g * ITEMS / G; it < (g + 1) * ITEMS / G; ++it) item(it);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!ONES) l_run = l_run * alpha + rs;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(pf[C][0][0]), "+v"(pf[C][0][1]), "+v"(pf[C][1][0]), "+v"(pf[C][1][1]));
#endif
    if (__any(alpha != 1.0f)) {  // wave-uniform, rare once the running max has settled; after the PV MFMAs by data dependence
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
    }
    tile_end();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using TT = std::true_type;
  using FF = std::false_type;

  issue(0, -1);
  issue(1, -1);
  tile_end();
  qk(0, st[0]);
  __syncthreads();  // K stage 0 is about to be overwritten by K(2)
  step(0, I0{}, FF{}, TT{});
  for (int tt = 1; tt + 1 < T; tt += 2) {
    step(tt, I1{}, TT{}, TT{});
    step(tt + 1, I0{}, TT{}, TT{});
  }
  step(T - 1, I1{}, TT{}, FF{});
  pv(T - 1, pf[1]);

  float l_tot;
  if (ONES) {
    constexpr int dvl = Dh >> 5, regl = ((Dh & 31) >> 3) * 4;
    const float v = oacc[dvl < NV ? dvl : 0][regl];
    const float other = __shfl_xor(v, 32);
    l_tot = hh == 0 ? v : other;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  const float inv = 1.0f / l_tot;
  const int query = q0 + r;
  if (query < p.Nq) {
#pragma unroll
    for (int dv = 0; dv < NV; ++dv)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dv * 32 + 8 * g + 4 * hh;
        if (d < Dh) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (bf16)(oacc[dv][4 * g + e] * inv);
          *reinterpret_cast<bf16x4*>(O + (long)query * p.ldo + d) = o4;
        }
      }
  }
}

template <int KC, int NV, int NW, int OCC>
int launch_attn_sp(crg_ctx* ctx, hipStream_t st, const AttnP& p) {
  dim3 grid((p.Nq + 32 * NW - 1) / (32 * NW), p.B * p.H);
  const unsigned k_bytes = (unsigned)((((long)p.Nk - 1) * p.ldk + p.Dh) * 2);
  const unsigned v_bytes = (unsigned)((((long)p.Dh - 1) * p.ldvt + (p.Nk + 7) / 8 * 8) * 2);  // whole 16-byte chunks (ldvt >= roundup(Nk, 8))
  hipLaunchKernelGGL((attn_sp_kernel<KC, NV, (KC * 8 < NV * 32), NW, OCC>), grid, dim3(64 * NW), 0, st, p, k_bytes, v_bytes);
  CRG_CHECK_LAUNCH(ctx, "attention (software-pipelined)");
  return 0;
}

template <int KS, int NV>
int launch_attn_ctx(crg_ctx* ctx, hipStream_t st, const AttnP& p, bool vrm) {
  // blocks per (batch, head): the largest divisor of the 128-query group count that keeps the grid within four blocks per CU,
  // so that every block walks the same number of groups behind ONE staging of the key tiles
  const int nqg = (p.Nq + 127) / 128, bh = p.B * p.H;
  int gx = 1;
  for (int d = 1; d <= nqg; ++d)
    if (nqg % d == 0 && (long)d * bh <= 1024) gx = d;
  dim3 grid(gx, bh);
  if (vrm) {
    if (p.Dh < NV * 32) hipLaunchKernelGGL((attn_ctx_kernel<KS, NV, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_ctx_kernel<KS, NV, false, true>), grid, dim3(256), 0, st, p);
  } else {
    if (p.Dh < NV * 32) hipLaunchKernelGGL((attn_ctx_kernel<KS, NV, true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_ctx_kernel<KS, NV, false, false>), grid, dim3(256), 0, st, p);
  }
  CRG_CHECK_LAUNCH(ctx, "attention (few keys)");
  return 0;
}

template <int KS, int NV>
int launch_attn(crg_ctx* ctx, hipStream_t st, const AttnP& p, bool vrm) {
  dim3 grid((p.Nq + 127) / 128, p.B * p.H);
  if (vrm) {
    if (p.Dh < NV * 32) hipLaunchKernelGGL((attn_kernel<KS, NV, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_kernel<KS, NV, false, true>), grid, dim3(256), 0, st, p);
  } else {
    if (p.Dh < NV * 32) hipLaunchKernelGGL((attn_kernel<KS, NV, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_kernel<KS, NV, false>), grid, dim3(256), 0, st, p);
  }
  CRG_CHECK_LAUNCH(ctx, "attention");
  return 0;
}

}  // namespace

static int attention_entry(crg_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           void* o, int64_t ldo, int B, int H, int Nq, int Nk, int Dh, float scale, int dtype, bool vrm) {
  if (!ctx) return -22;
  CRG_REQUIRE(ctx, dtype == CRG_BF16, "attention: only bf16 is implemented (dtype %d)", dtype);
  CRG_REQUIRE(ctx, B > 0 && H > 0 && Nq > 0 && Nk > 0, "attention: empty problem B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
  CRG_REQUIRE(ctx, Dh % 8 == 0 && Dh >= 8 && Dh <= 160, "attention: head dim %d unsupported (multiple of 8, <= 160)", Dh);
  CRG_REQUIRE(ctx, ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attention: leading dimensions must keep 16-byte alignment");
  if (vrm) CRG_REQUIRE(ctx, ldv >= (int64_t)H * Dh, "attention: ldv=%ld must cover H*Dh=%d", (long)ldv, H * Dh);
  else CRG_REQUIRE(ctx, ldv >= ((Nk + 7) / 8) * 8, "attention: ldvt=%ld must cover Nk=%d rounded up to 8", (long)ldv, Nk);
  CRG_REQUIRE(ctx, (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)o & 7) == 0, "attention: pointers must be 16-byte aligned");
  AttnP p{(const bf16*)q, (const bf16*)k, (const bf16*)v, (bf16*)o, (long)ldq, (long)ldk, (long)ldv, (long)ldo, B, H, Nq, Nk, Dh,
          scale * 1.4426950408889634f};
  hipStream_t st = (hipStream_t)stream;
  const double flops = 4.0 * B * H * (double)Nq * Nk * Dh;
  const double bytes = 2.0 * B * H * Dh * (2.0 * Nq + 2.0 * Nk);
  crg_prof_scope ps(ctx, st, CRG_K_ATTN, flops, bytes);
  const int ks = (Dh + 15) / 16;
  // few keys (cross-attention against the prompt context, FaceID tokens, 8x8 self-attention): K / V staged once per block, no barrier
  // in the loop.  CRG_ATTN_CTX (developer knob): 0 = the general kernels for these shapes too
  static const int use_ctx = getenv("CRG_ATTN_CTX") ? atoi(getenv("CRG_ATTN_CTX")) : 1;
  // ... used where a block then walks at least two query groups behind one staging of the keys (measured, kernel trace: 8 x 4096
  // queries x 77 keys, d 40: 22.3 -> 19.6 us; with ONE group per block - 1024 queries, d 80 - the general kernel's interior-tile
  // staging is the faster one, 12.6 vs 13.5 us).  These launches are bound by their 80-byte-per-head Q / O row slices (32 cache
  // lines per wave-instruction), not by block count: see DESIGN 4.
  const int nqg_ = (Nq + 127) / 128;
  if (use_ctx && Nk <= 128 && (long)nqg_ * B * H >= 2048) {
    switch (ks) {
      case 1: return launch_attn_ctx<1, 1>(ctx, st, p, vrm);
      case 2: return launch_attn_ctx<2, 1>(ctx, st, p, vrm);
      case 3: return launch_attn_ctx<3, 2>(ctx, st, p, vrm);
      case 4: return launch_attn_ctx<4, 2>(ctx, st, p, vrm);
      case 5: return launch_attn_ctx<5, 3>(ctx, st, p, vrm);
      case 6: return launch_attn_ctx<6, 3>(ctx, st, p, vrm);
      case 7: return launch_attn_ctx<7, 4>(ctx, st, p, vrm);
      case 8: return launch_attn_ctx<8, 4>(ctx, st, p, vrm);
      case 9: return launch_attn_ctx<9, 5>(ctx, st, p, vrm);
      case 10: return launch_attn_ctx<10, 5>(ctx, st, p, vrm);
    }
  }
  // LDS-DMA forms: transposed V, whole 64-key tiles, extents addressable through a 32-bit buffer descriptor.
  // CRG_ATTN_DMA (developer knob): 0 = register-staged kernel only, 1 (default) = software-pipelined where instantiated, else the
  // plain LDS-DMA kernel, 8 / 4 = plain LDS-DMA kernel on 8 / 4 waves
  static const int use_dma = getenv("CRG_ATTN_DMA") ? atoi(getenv("CRG_ATTN_DMA")) : 1;
  static const int dma_tail = getenv("CRG_ATTN_TAIL") ? atoi(getenv("CRG_ATTN_TAIL")) : 1;  // key tails on the LDS-DMA kernel (0: register-staged)
  if (use_dma && !vrm && (Nk % 64 == 0 || (dma_tail && Nk > 8)) && ((long)Nk * ldk * 2 < (1l << 31)) && ((long)Dh * ldv * 2 < (1l << 31))) {
    const bool sp = use_dma == 1 && Nk % 128 == 0;
    static const int use_occ4 = getenv("CRG_ATTN_OCC4") ? atoi(getenv("CRG_ATTN_OCC4")) : 1;
    if (Dh == 40) {
      if (sp) return use_occ4 ? launch_attn_sp<5, 2, 8, 4>(ctx, st, p) : launch_attn_sp<5, 2, 8, 2>(ctx, st, p);
      // few key tiles (cross-attention: 77 keys): 128-query blocks, i.e. twice the blocks and four-wave barriers (19.2 vs 19.9 us)
      return (use_dma == 4 || (use_dma == 1 && Nk <= 256)) ? launch_attn_dma<5, 2, 4, 1, 4>(ctx, st, p) : launch_attn_dma<5, 2, 4, 1, 8>(ctx, st, p);
    }
    if (Dh == 64) {  // measured (B4 N4096 h10: register-staged 256 us, pipelined 253, LDS-DMA on 8 waves 233, on 4 waves 226): plain form, 4 waves
      if (use_dma == 5 && Nk % 128 == 0) return launch_attn_sp<8, 2, 8, 2>(ctx, st, p);
      return use_dma == 8 ? launch_attn_dma<8, 2, 4, 1, 8>(ctx, st, p) : launch_attn_dma<8, 2, 4, 1, 4>(ctx, st, p);
    }
    if (Dh == 80) {
      if (sp) return launch_attn_sp<10, 3, 8, 2>(ctx, st, p);
      return use_dma == 4 ? launch_attn_dma<10, 3, 3, 1, 4>(ctx, st, p) : launch_attn_dma<10, 3, 2, 1, 8>(ctx, st, p);
    }
  }
  switch (ks) {
    case 1: return launch_attn<1, 1>(ctx, st, p, vrm);
    case 2: return launch_attn<2, 1>(ctx, st, p, vrm);
    case 3: return launch_attn<3, 2>(ctx, st, p, vrm);
    case 4: return launch_attn<4, 2>(ctx, st, p, vrm);
    case 5: return launch_attn<5, 3>(ctx, st, p, vrm);
    case 6: return launch_attn<6, 3>(ctx, st, p, vrm);
    case 7: return launch_attn<7, 4>(ctx, st, p, vrm);
    case 8: return launch_attn<8, 4>(ctx, st, p, vrm);
    case 9: return launch_attn<9, 5>(ctx, st, p, vrm);
    case 10: return launch_attn<10, 5>(ctx, st, p, vrm);
  }
  return crg_fail(ctx, -22, "attention: head dim %d unsupported", Dh);
}

extern "C" int crg_attention(crg_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k, int64_t ldk,
                             const void* vt, int64_t ldvt, void* o, int64_t ldo, int B, int H, int Nq, int Nk, int Dh,
                             float scale, int dtype) {
  return attention_entry(ctx, stream, q, ldq, k, ldk, vt, ldvt, o, ldo, B, H, Nq, Nk, Dh, scale, dtype, false);
}

extern "C" int crg_attention_v(crg_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k, int64_t ldk,
                               const void* v, int64_t ldv, void* o, int64_t ldo, int B, int H, int Nq, int Nk, int Dh,
                               float scale, int dtype) {
  return attention_entry(ctx, stream, q, ldq, k, ldk, v, ldv, o, ldo, B, H, Nq, Nk, Dh, scale, dtype, true);
}

#ifdef CRG_ATTN_LAPS
extern "C" int crg_debug_read_attn(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(crg_attn_laps), sizeof(unsigned long long) * (size_t)n);
}
#endif
