// Dev probe (standalone, round 4): operand layout and issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 - the block-scaled MX
// matrix instruction (MI355X_MICROARCH.md "Matrix cores": e4m3 operands 2x, e2m3 / e2m1 operands 4x the bf16 rate per clock).
//   1. layout: D[16][16] = sum_k A[i][k] B[k][j] * 2^(sa - 127) * 2^(sb - 127) with random e4m3 bytes, under the hypothesis
//      "lane l supplies row / column l & 15 and the 32 consecutive k of block l >> 4, byte b of its 8 registers = k 32 (l >> 4) + b;
//       its scale byte (op_sel 0 = byte 0 of the scale register) applies to that block"; C/D: col = l & 15, row = 4 (l >> 4) + r.
//   2. rate: cycles per instruction in a bare dependent-free loop, one wave per SIMD, for fp8 / fp6 / fp4 and for v_mfma_f32_16x16x32_f16.
// Build: hipcc --offload-arch=gfx950 -O3 mx_layout.hip -o mx_layout
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

static float e4m3_to_float(unsigned char v) {  // OCP e4m3fn: bias 7, no inf, 0x7f / 0xff = NaN
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

__global__ void mx_once(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* D) {
  const int l = threadIdx.x, r = l & 15, blk = l >> 4;
  v8i a, b;
  for (int q = 0; q < 8; ++q) {
    a[q] = *reinterpret_cast<const int*>(A + r * 128 + blk * 32 + q * 4);   // A[row r][k]
    b[q] = *reinterpret_cast<const int*>(B + r * 128 + blk * 32 + q * 4);   // B^T[col r][k]
  }
  const int sa = SA[r * 4 + blk], sb = SB[r * 4 + blk];
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  for (int q = 0; q < 4; ++q) D[(blk * 4 + q) * 16 + r] = acc[q];
}

template <int FMT>
__global__ __launch_bounds__(256) void mx_rate(int iters, float* out, unsigned long long* cyc) {
  v8i a, b;
  for (int q = 0; q < 8; ++q) { a[q] = 0x38383838 + threadIdx.x * 0x01000100 * q; b[q] = 0x30303030 + threadIdx.x * 0x00010001 * (q + 1); }
  v4f c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
  h8 ha, hb;
  for (int q = 0; q < 8; ++q) { ha[q] = (_Float16)(0.01f * (threadIdx.x + q)); hb[q] = (_Float16)(0.02f * q - 0.05f); }
  const int sc = 127;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define ONE(C)                                                                                                                         \
  if (FMT == 9) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C) : "v"(ha), "v"(hb));                                     \
  else if (FMT == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(C) : "v"(a), "v"(b), "v"(sc)); \
  else if (FMT == 2) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(C) : "v"(*reinterpret_cast<__attribute__((ext_vector_type(6))) int*>(&a)), "v"(*reinterpret_cast<__attribute__((ext_vector_type(6))) int*>(&b)), "v"(sc)); \
  else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(C) : "v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) int*>(&a)), "v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) int*>(&b)), "v"(sc));
    ONE(c0) ONE(c1) ONE(c2) ONE(c3) ONE(c4) ONE(c5) ONE(c6) ONE(c7)
#undef ONE
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FMT>
void rate(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  hipLaunchKernelGGL((mx_rate<FMT>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
  hipLaunchKernelGGL((mx_rate<FMT>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
  printf("%-28s %6.1f cycles per instruction (one wave per SIMD, every CU busy)\n", name, s / 256 / (iters * 8.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  unsigned char hA[16 * 128], hB[16 * 128], hSA[64], hSB[64];
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) {
    do { hA[i] = rand() & 255; } while ((hA[i] & 0x7f) == 0x7f || ((hA[i] >> 3) & 15) > 9);   // no NaN, moderate magnitudes
    do { hB[i] = rand() & 255; } while ((hB[i] & 0x7f) == 0x7f || ((hB[i] >> 3) & 15) > 9);
  }
  unsigned char *dA, *dB, *dSA, *dSB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int variant = 0; variant < 4; ++variant) {
    // 0: every scale 1.0; 1: scales differ per row only; 2: per k-block only; 3: per (row, block)
    for (int i = 0; i < 64; ++i) {
      const int row = i / 4, blk = i % 4;
      hSA[i] = variant == 0 ? 127 : variant == 1 ? 120 + row % 9 : variant == 2 ? 121 + 2 * blk : 120 + (row * 5 + blk * 3) % 10;
      hSB[i] = variant == 0 ? 127 : variant == 1 ? 122 + row % 7 : variant == 2 ? 124 + blk : 122 + (row * 3 + blk * 5) % 8;
    }
    hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mx_once, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD);
    float hD[256]; hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0, worst_t = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k)
          ref += (double)e4m3_to_float(hA[i * 128 + k]) * e4m3_to_float(hB[j * 128 + k]) * ldexp(1.0, hSA[i * 4 + k / 32] - 127) * ldexp(1.0, hSB[j * 4 + k / 32] - 127);
        worst = fmax(worst, fabs(ref - hD[i * 16 + j]));
        worst_t = fmax(worst_t, fabs(ref - hD[j * 16 + i]));
        ref_max = fmax(ref_max, fabs(ref));
      }
    printf("scales variant %d: max |D - ref| = %.3e (transposed D: %.3e), max |ref| = %.3e -> %s\n", variant, worst, worst_t, ref_max,
           worst <= 1e-5 * ref_max + 1e-12 ? "CONFIRMED" : (worst_t <= 1e-5 * ref_max + 1e-12 ? "CONFIRMED with D transposed" : "WRONG"));
  }
  rate<9>("v_mfma_f32_16x16x32_f16");
  rate<0>("mx 16x16x128 fp8 (e4m3)");
  rate<2>("mx 16x16x128 fp6 (e2m3)");
  rate<4>("mx 16x16x128 fp4 (e2m1)");
  return 0;
}
