#include <hip/hip_runtime.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
// probe: LDS filled with element index (as fp16 integers 0..2047); lane l supplies byte address addr[l]; dumps the 4 halfs each lane gets
extern "C" __global__ void probe(const int* addr, float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (_Float16)(float)i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  unsigned a = base + (unsigned)addr[threadIdx.x];
  h4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}
