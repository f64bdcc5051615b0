// Dev probe (standalone, VERDICT r3 item 7 / MI355X_MICROARCH.md 'DVFS give-back' item 7): does the attention loop's matrix work run faster
// as v_mfma_f32_16x16x32_bf16 than as v_mfma_f32_32x32x16_bf16 once the chip's clock under load is part of the answer?
// Every CU runs the same stream on RANDOM operands (registers, no memory traffic in the loop): per "64-key tile" either 14 MFMAs 32x32x16
// or 28 MFMAs 16x16x32 (the same FLOPs) interleaved with the flash-attention softmax mix of cremage_amd/csrc/attention.hip per tile and
// wave (32 v_exp_f32, 16 v_cvt_pk_bf16_f32, 16 v_max3_f32, 16 v_add_f32, 15 v_mul_f32 - the d_head 40 kernel after the shift moved into
// the MFMA), or with no vector work at all (the bare loops of the guide).  Reported: wall time per tile (hipEvents over a >= 20 ms
// launch), wave-cycles per tile (s_memtime) and the clock the chip held (cycles / s_memrealtime at 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape_mix.hip -o mfma_shape_mix ; run: ./mfma_shape_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SHAPE, bool MIX>
__global__ __launch_bounds__(256, 2) void probe(int tiles, float* out, unsigned long long* stamps) {
  bf16x8 a[4], b[4];
  for (int r = 0; r < 4; ++r)
    for (int i = 0; i < 8; ++i) {
      const unsigned h = hash32((blockIdx.x * 256 + threadIdx.x) * 64 + r * 8 + i);
      a[r][i] = (__bf16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
      b[r][i] = (__bf16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
    }
  f32x16 acc32[2] = {};
  f32x4 acc16[8] = {};
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = -0.01f * ((threadIdx.x + 3 * i) & 31) - 0.5f;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int g = 0; g < 14; ++g) {
      if (SHAPE == 32) {
        acc32[g & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g & 3], b[(g + 1) & 3], acc32[g & 1], 0, 0, 0);
      } else {
        acc16[(2 * g) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g & 3], b[(g + 1) & 3], acc16[(2 * g) & 7], 0, 0, 0);
        acc16[(2 * g + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(g + 2) & 3], b[(g + 3) & 3], acc16[(2 * g + 1) & 7], 0, 0, 0);
      }
      if (MIX) {  // 95 vector instructions per tile over the 14 gaps: 7 per gap (the last gap 4)
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[g & 15]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[(g + 5) & 15]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[(g + 1) & 15]) : "v"(x[(g + 2) & 15]), "v"(x[(g + 3) & 15]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[(g + 4) & 15]) : "v"(x[(g + 6) & 15]), "v"(x[(g + 7) & 15]));
        if (g < 13) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(g + 8) & 15]) : "v"(x[(g + 9) & 15]));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[(g + 10) & 15]) : "v"(x[(g + 11) & 15]));
          if (g & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(g + 12) & 15])); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[(g + 13) & 15]) : "v"(x[g & 15]), "v"(x[(g + 14) & 15]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc32[0][i] + acc32[1][i] + x[i];
  for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, bool MIX>
void run(const char* name, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, tiles = 200000;
  float* out; unsigned long long* st;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&st, (size_t)blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<SHAPE, MIX>), dim3(blocks), dim3(256), 0, 0, tiles, out, st);  // ~1 s of warm load
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<SHAPE, MIX>), dim3(blocks), dim3(256), 0, 0, tiles, out, st);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc((size_t)blocks * 16);
  hipMemcpy(h, st, (size_t)blocks * 16, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  cyc /= blocks; rt /= blocks;
  const double flop = 14.0 * 2 * 32 * 32 * 16 * (double)tiles * blocks * 4;  // per wave and tile 14 x 32x32x16 (or 28 x 16x16x32)
  printf("%-26s %d waves/SIMD: %7.1f ns wall per tile, %7.1f wave-cycles per tile, clock %.2f GHz, %7.1f TFLOP/s\n", name, blocks_per_cu, ms * 1e6 / tiles,
         cyc / tiles, cyc / rt * 0.1, flop / (ms * 1e-3) / 1e12);
  free(h); hipFree(out); hipFree(st);
}

int main() {
  for (int bpc : {1, 2}) {
    run<32, false>("32x32x16 bare", bpc);
    run<16, false>("16x16x32 bare", bpc);
    run<32, true>("32x32x16 + softmax mix", bpc);
    run<16, true>("16x16x32 + softmax mix", bpc);
  }
  return 0;
}
