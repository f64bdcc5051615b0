// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (raw buffer, stride 0) write ZEROS to its LDS slot,
// or leave the slot untouched?  (conv_ring.hip wants to get its zero padding from the range check.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned* src, int nbytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* s32 = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < 256; i += 64) s32[i] = 0xDEADBEEFu;   // poison
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x80000000;          // odd lanes: far out of range
  if (threadIdx.x == 62) voff = nbytes - 8;        // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)smem, 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = s32[i];
}
int main() {
  std::vector<unsigned> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, 1024, o);
  std::vector<unsigned> r(256);
  hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) if (l < 6 || l >= 60) printf("lane %2d: %08x %08x %08x %08x\n", l, r[4*l], r[4*l+1], r[4*l+2], r[4*l+3]);
  int zeros = 0, poison = 0;
  for (int l = 1; l < 64; l += 2) { if (r[4*l] == 0) ++zeros; if (r[4*l] == 0xDEADBEEFu) ++poison; }
  printf("odd (out-of-range) lanes: %d zero-filled, %d left poisoned\n", zeros, poison);
  return 0;
}
