// Multi-tensor Adam and EMA: one launch walks up to 48 parameter tensors (descriptors travel as kernel
// arguments, so nothing is staged through device memory and the call is re-entrant).
// Replaces torch.optim.Adam.step (BigGAN_PyTorch/trainer.py:158-171; betas=(0,0.999), eps=1e-6 in every
// shipped config) and utils.ema.update (BigGAN_PyTorch/utils.py:1055-1067: per state_dict key incl. buffers).
// HBM-bound: Adam streams 4 reads + 3 writes of 4 B per element, EMA 2 reads + 1 write.
#include "icg_common.h"

#define ICG_MT_MAX 48
#define ICG_MT_CHUNK 4096  // elements per block-iteration

struct AdamPack {
  icg_adam_tensor t[ICG_MT_MAX];
  int blk_start[ICG_MT_MAX + 1];
  int n;
};
struct EmaPack {
  icg_ema_tensor t[ICG_MT_MAX];
  int blk_start[ICG_MT_MAX + 1];
  int n;
};

template <typename Pack>
__device__ __forceinline__ int find_tensor(const Pack& p, int blk) {
  int lo = 0, hi = p.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.blk_start[mid] <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ void adam_one(float g, float& m, float& v, float& pw, float w, float beta2, float one_minus_beta2, float eps,
                                         float step_size, float bc2_sqrt) {
  // exp_avg.lerp_(grad, 1-beta1)  (ATen lerp: two-sided formula)
  m = (w < 0.5f) ? m + w * (g - m) : g - (g - m) * (1.f - w);
  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
  v = v * beta2 + one_minus_beta2 * (g * g);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  pw = pw - step_size * (m / denom);
}

// A block owns ICG_MT_CHUNK elements of one tensor.  Tensors whose four buffers are 16-byte aligned (every PyTorch allocation;
// a chunk starts at a multiple of 4096 elements) go through float4 accesses with all 16 loads of a thread in flight before the
// first store -- 4-byte accesses left this at 0.63 of 8 TB/s (profiles/r03_hbm_kernels_microbench.txt); same arithmetic per element.
__global__ __launch_bounds__(256) void adam_kernel(AdamPack p, float beta1, float beta2, float one_minus_beta2,
                                                   float eps, float step_size, float bc2_sqrt) {
  const int ti = find_tensor(p, blockIdx.x);
  const icg_adam_tensor t = p.t[ti];
  const long base = (long)(blockIdx.x - p.blk_start[ti]) * ICG_MT_CHUNK;
  const long end = min((long)t.numel, base + ICG_MT_CHUNK);
  const float w = 1.f - beta1;
  const bool vec = (((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0 &&
                   end - base == ICG_MT_CHUNK;
  if (vec) {
    constexpr int U = ICG_MT_CHUNK / 4 / 256;       // float4 per thread
    float4 g[U], m[U], v[U], pw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = (base >> 2) + threadIdx.x + u * 256;
      g[u] = reinterpret_cast<const float4*>(t.grad)[i];
      m[u] = reinterpret_cast<const float4*>(t.exp_avg)[i];
      v[u] = reinterpret_cast<const float4*>(t.exp_avg_sq)[i];
      pw[u] = reinterpret_cast<const float4*>(t.param)[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = (base >> 2) + threadIdx.x + u * 256;
      adam_one(g[u].x, m[u].x, v[u].x, pw[u].x, w, beta2, one_minus_beta2, eps, step_size, bc2_sqrt);
      adam_one(g[u].y, m[u].y, v[u].y, pw[u].y, w, beta2, one_minus_beta2, eps, step_size, bc2_sqrt);
      adam_one(g[u].z, m[u].z, v[u].z, pw[u].z, w, beta2, one_minus_beta2, eps, step_size, bc2_sqrt);
      adam_one(g[u].w, m[u].w, v[u].w, pw[u].w, w, beta2, one_minus_beta2, eps, step_size, bc2_sqrt);
      reinterpret_cast<float4*>(t.param)[i] = pw[u];
      reinterpret_cast<float4*>(t.exp_avg)[i] = m[u];
      reinterpret_cast<float4*>(t.exp_avg_sq)[i] = v[u];
    }
    return;
  }
  for (long i = base + threadIdx.x; i < end; i += 256) {
    const float g = t.grad[i];
    float m = t.exp_avg[i], v = t.exp_avg_sq[i], pw = t.param[i];
    adam_one(g, m, v, pw, w, beta2, one_minus_beta2, eps, step_size, bc2_sqrt);
    t.param[i] = pw;
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(EmaPack p, float decay, float one_minus_decay) {
  const int ti = find_tensor(p, blockIdx.x);
  const icg_ema_tensor t = p.t[ti];
  const long base = (long)(blockIdx.x - p.blk_start[ti]) * ICG_MT_CHUNK;
  const long end = min((long)t.numel, base + ICG_MT_CHUNK);
  if (((((uintptr_t)t.target | (uintptr_t)t.source) & 15) == 0) && end - base == ICG_MT_CHUNK) {
    constexpr int U = ICG_MT_CHUNK / 4 / 256;
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = (base >> 2) + threadIdx.x + u * 256;
      a[u] = reinterpret_cast<const float4*>(t.target)[i];
      b[u] = reinterpret_cast<const float4*>(t.source)[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = (base >> 2) + threadIdx.x + u * 256;
      float4 o;
      o.x = a[u].x * decay + b[u].x * one_minus_decay; o.y = a[u].y * decay + b[u].y * one_minus_decay;
      o.z = a[u].z * decay + b[u].z * one_minus_decay; o.w = a[u].w * decay + b[u].w * one_minus_decay;
      reinterpret_cast<float4*>(t.target)[i] = o;
    }
    return;
  }
  for (long i = base + threadIdx.x; i < end; i += 256)
    t.target[i] = t.target[i] * decay + t.source[i] * one_minus_decay;
}

extern "C" int icg_adam_multi(const icg_adam_tensor* tensors, int n, float lr, float beta1, float beta2, float eps,
                              int step, void* stream) {
  ICG_REQUIRE(tensors && n >= 0 && step >= 1);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float omb2 = (float)(1.0 - (double)beta2);
  int done = 0;
  while (done < n) {
    AdamPack p;
    p.n = 0;
    int blocks = 0;
    while (done < n && p.n < ICG_MT_MAX) {
      const icg_adam_tensor& t = tensors[done++];
      if (t.numel <= 0) continue;
      ICG_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq);
      p.t[p.n] = t;
      p.blk_start[p.n] = blocks;
      blocks += (int)icg_cdiv(t.numel, ICG_MT_CHUNK);
      p.n++;
    }
    if (p.n == 0) break;
    p.blk_start[p.n] = blocks;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, beta1, beta2, omb2, eps,
                       step_size, bc2_sqrt);
    int rc = icg_check_launch();
    if (rc != ICG_OK) return rc;
  }
  return ICG_OK;
}

extern "C" int icg_ema_multi(const icg_ema_tensor* tensors, int n, float decay, void* stream) {
  ICG_REQUIRE(tensors && n >= 0);
  const float omd = (float)(1.0 - (double)decay);
  int done = 0;
  while (done < n) {
    EmaPack p;
    p.n = 0;
    int blocks = 0;
    while (done < n && p.n < ICG_MT_MAX) {
      const icg_ema_tensor& t = tensors[done++];
      if (t.numel <= 0) continue;
      ICG_REQUIRE(t.target && t.source);
      p.t[p.n] = t;
      p.blk_start[p.n] = blocks;
      blocks += (int)icg_cdiv(t.numel, ICG_MT_CHUNK);
      p.n++;
    }
    if (p.n == 0) break;
    p.blk_start[p.n] = blocks;
    hipLaunchKernelGGL(ema_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, decay, omd);
    int rc = icg_check_launch();
    if (rc != ICG_OK) return rc;
  }
  return ICG_OK;
}
