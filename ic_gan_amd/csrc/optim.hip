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

__global__ __launch_bounds__(256) void adam_kernel(AdamPack p, float beta1, float beta2, float one_minus_beta2,
                                                   float eps, float step_size, float bc2_sqrt) {
  const int ti = find_tensor(p, blockIdx.x);
  const icg_adam_tensor t = p.t[ti];
  const long base = (long)(blockIdx.x - p.blk_start[ti]) * ICG_MT_CHUNK;
  const long end = min((long)t.numel, base + ICG_MT_CHUNK);
  const float w = 1.f - beta1;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    const float g = t.grad[i];
    float m = t.exp_avg[i], v = t.exp_avg_sq[i];
    // exp_avg.lerp_(grad, 1-beta1)  (ATen lerp: two-sided formula)
    m = (w < 0.5f) ? m + w * (g - m) : g - (g - m) * (1.f - w);
    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    v = v * beta2 + one_minus_beta2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    t.param[i] = t.param[i] - step_size * (m / denom);
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(EmaPack p, float decay, float one_minus_decay) {
  const int ti = find_tensor(p, blockIdx.x);
  const icg_ema_tensor t = p.t[ti];
  const long base = (long)(blockIdx.x - p.blk_start[ti]) * ICG_MT_CHUNK;
  const long end = min((long)t.numel, base + ICG_MT_CHUNK);
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
