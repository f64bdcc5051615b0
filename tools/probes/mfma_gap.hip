// Dev probe (standalone): cycles per MFMA gap when K independent VALU instructions of one kind follow every v_mfma_f32_32x32x16_bf16
// of ONE wave (dependent accumulator chain), with 1, 2 or 4 waves per SIMD running the same stream.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_gap.hip -o mfma_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int K>
__global__ __launch_bounds__(1024) void probe(int iters, float* out, long long* cyc) {
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * i); b[i] = (__bf16)(0.02f * (threadIdx.x & 7)); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i);
  __syncthreads();
  const long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
        else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[k & 7]));
        else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[k & 7]));
        else asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&x[2 * (k & 3)])));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&cyc[0], (unsigned long long)(c1 - c0));  // the last finisher
}

template <int KIND, int K>
void run(int threads, const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 4 * 4); hipMalloc(&cyc, 8 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<KIND, K>), dim3(1), dim3(threads), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((probe<KIND, K>), dim3(1), dim3(threads), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-10s K=%d waves/SIMD=%d: %6.1f cycles per MFMA gap of the slowest wave, %6.1f per SIMD-MFMA\n", name, K, threads / 256,
         (double)c / (iters * 8.0), (double)c / (iters * 8.0) / (threads / 256));
  hipFree(out); hipFree(cyc);
}

template <int KIND>
void sweep(const char* name) {
  for (int th : {256, 512, 1024}) {
    run<KIND, 0>(th, name); run<KIND, 2>(th, name); run<KIND, 4>(th, name); run<KIND, 6>(th, name); run<KIND, 8>(th, name);
  }
}
int main() {
  sweep<0>("v_exp");
  sweep<1>("v_fma");
  sweep<2>("v_cvt_pk");
  sweep<3>("v_pk_fma");
  return 0;
}
