// Dev probe (standalone): issue cost of the attention loop's instruction classes on gfx950 (shader cycles from clock64) and
// whether VALU work of one wave overlaps MFMA work of another wave on the same SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mfma_rate.hip -o valu_mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct St {
  float x[32];
  f32x16 acc0, acc1;
  bf16x8 a, b;
};
template <int OP>
__device__ __forceinline__ void body(St& s) {
  if constexpr (OP == 0) {  // 32 independent v_exp_f32
#pragma unroll
    for (int i = 0; i < 32; ++i) s.x[i] = __builtin_amdgcn_exp2f(s.x[i]);
  } else if constexpr (OP == 1) {  // 16 independent v_pk_fma_f32
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      f32x2 v = {s.x[i], s.x[i + 1]};
      v = v * f32x2{1.0001f, 0.9999f} + f32x2{0.5f, 0.25f};
      s.x[i] = v[0];
      s.x[i + 1] = v[1];
    }
  } else if constexpr (OP == 2) {  // 8 MFMAs, two accumulator chains
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s.acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a, s.b, s.acc0, 0, 0, 0);
      s.acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s.a, s.b, s.acc1, 0, 0, 0);
    }
  } else if constexpr (OP == 4) {  // 8 MFMAs + 32 exp2, independent, same wave
    body<2>(s);
    body<0>(s);
  } else if constexpr (OP == 5) {  // dependent v_max3 chain (16)
    float m = s.x[0];
#pragma unroll
    for (int i = 1; i < 31; i += 2) m = fmaxf(fmaxf(m, s.x[i]), s.x[i + 1]);
    s.x[0] = m * 0.999f;
  } else if constexpr (OP == 6) {  // 16 v_cvt_pk_bf16_f32
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
      bf16x2 h = {(__bf16)s.x[i], (__bf16)s.x[i + 1]};
      unsigned u;
      __builtin_memcpy(&u, &h, 4);
      s.x[i] = __uint_as_float(u | 0x3f000000u);
    }
  } else if constexpr (OP == 7) {  // 64 dependent v_add_f32
    float v = s.x[0];
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(s.x[1]));
    s.x[0] = v;
  } else if constexpr (OP == 8) {  // 64 independent v_add_f32 (2 per value)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.x[(i + 1) & 31]));
  }
}

// MODE 3: waves 0-3 run MFMAs, waves 4-7 run exp2 (wave w and w + 4 share a SIMD); otherwise every wave runs body<MODE>
template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  St s;
#pragma unroll
  for (int i = 0; i < 32; ++i) s.x[i] = -0.001f * (threadIdx.x + i);
  s.acc0 = f32x16{0};
  s.acc1 = f32x16{0};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s.a[i] = (__bf16)(0.01f * i);
    s.b[i] = (__bf16)(0.02f * i);
  }
  __syncthreads();
  const long long c0 = clock64();
  if constexpr (MODE == 3) {
    if (wave < 4) {
      for (int it = 0; it < iters; ++it) body<2>(s);
    } else {
      for (int it = 0; it < iters; ++it) body<0>(s);
    }
  } else {
    for (int it = 0; it < iters; ++it) body<MODE>(s);
  }
  const long long c1 = clock64();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) r += s.x[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) r += s.acc0[i] + s.acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int MODE>
static void run(const char* name, int threads, int iters, float* out, long long* cyc) {
  static long long h[256 * 8];
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, iters, out, cyc);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double w0 = 0, w4 = 0;
  for (int i = 0; i < 256; ++i) {
    w0 += h[i * 8];
    w4 += h[i * 8 + 4];
  }
  printf("%d waves/SIMD  %-44s: wave 0 %8.1f cycles per iteration", threads / 256, name, w0 / 256 / iters);
  if (threads == 512) printf(", wave 4 %8.1f", w4 / 256 / iters);
  printf("\n");
}

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * sizeof(float));
  (void)hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, 400000, out, cyc);  // ~100 ms warm-up: clocks up
  (void)hipDeviceSynchronize();
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads += 256) {
    run<0>("32 x v_exp_f32 (independent)", threads, iters, out, cyc);
    run<1>("16 x v_pk_fma_f32 (independent)", threads, iters, out, cyc);
    run<8>("64 x v_add_f32 (independent)", threads, iters, out, cyc);
    run<7>("64 x v_add_f32 (dependent chain)", threads, iters, out, cyc);
    run<5>("16 x v_max3_f32 (dependent chain)", threads, iters, out, cyc);
    run<6>("16 x v_cvt_pk_bf16_f32 + or", threads, iters, out, cyc);
    run<2>("8 x mfma 32x32x16 bf16 (2 chains)", threads, iters, out, cyc);
    run<4>("8 mfma + 32 exp2 in one wave", threads, iters, out, cyc);
    if (threads == 512) run<3>("waves 0-3: 8 mfma | waves 4-7: 32 exp2", threads, iters, out, cyc);
  }
  return 0;
}
