// tools only: hand-written device copies -- this box's HBM ceiling for a 1-read + 1-write stream, and how the access shape
// (float4s in flight per thread, grid size, non-temporal hints) moves it.  python tools/hbm_bench.py prints the table.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// U float4 per thread and block-iteration, all loads issued before the first store; G = 1: grid-stride over a fixed grid
template <int U, int NT>
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, long n4) {
  const long step = (long)gridDim.x * 256 * U;
  for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += step) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 256;
      if (i < n4) v[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 256;
      if (i < n4) {
        if (NT) __builtin_nontemporal_store(v[u], b + i); else b[i] = v[u];
      }
    }
  }
}
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f4* __restrict__ a, float* __restrict__ out, long n4) {
  const long step = (long)gridDim.x * 256 * U;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += step) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 256;
      if (i < n4) s += a[i];
    }
  }
  if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;
}
template <int U>
__global__ __launch_bounds__(256) void write_kernel(f4* __restrict__ b, long n4) {
  const long step = (long)gridDim.x * 256 * U;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += step) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * 256;
      if (i < n4) b[i] = v;
    }
  }
}

extern "C" int hbm_copy_launch(const void* a, void* b, long n4, int variant, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)blocks), t(256);
  switch (variant) {
    case 0: hipLaunchKernelGGL((copy_kernel<1, 0>), g, t, 0, st, (const f4*)a, (f4*)b, n4); break;
    case 1: hipLaunchKernelGGL((copy_kernel<2, 0>), g, t, 0, st, (const f4*)a, (f4*)b, n4); break;
    case 2: hipLaunchKernelGGL((copy_kernel<4, 0>), g, t, 0, st, (const f4*)a, (f4*)b, n4); break;
    case 3: hipLaunchKernelGGL((copy_kernel<8, 0>), g, t, 0, st, (const f4*)a, (f4*)b, n4); break;
    case 4: hipLaunchKernelGGL((copy_kernel<4, 1>), g, t, 0, st, (const f4*)a, (f4*)b, n4); break;
    case 5: hipLaunchKernelGGL((read_kernel<4>), g, t, 0, st, (const f4*)a, (float*)b, n4); break;
    case 6: hipLaunchKernelGGL((write_kernel<4>), g, t, 0, st, (f4*)b, n4); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
