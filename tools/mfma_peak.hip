// micro-benchmark (tools only): sustained v_mfma_f32_32x32x2_f32 rate on this box (sets the practical fp32 MFMA roof)
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters, float a0) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = 1.0f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int mfma_peak_launch(float* out, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.5f);
  return (int)hipGetLastError();
}
