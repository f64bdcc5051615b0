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
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kb], 0, 0, 0);
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
          oacc[dv] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s2], oacc[dv], 0, 0, 0);
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
