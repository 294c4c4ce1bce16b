// StyleGAN2 custom ops for gfx950: bias_act and upfirdn2d (fp32).
// Same contracts as the reference's CUDA plugins
//   stylegan2_ada_pytorch/torch_utils/ops/bias_act.{cpp,cu}  (bias_act.cpp:35-100, bias_act.cu:26-150)
//   stylegan2_ada_pytorch/torch_utils/ops/upfirdn2d.{cpp,cu} (upfirdn2d.cpp:19-104, upfirdn2d.cu:32-203)
// re-derived from their PyTorch reference implementations (_bias_act_ref bias_act.py:177-207,
// _upfirdn2d_ref upfirdn2d.py:199-246).  Both are HBM-bound: one read + one write per element.
#include "icg_common.h"

// activation ids follow the reference table (bias_act.py:25-106, `cuda_idx`)
enum { ACT_LINEAR = 1, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID, ACT_ELU, ACT_SELU, ACT_SOFTPLUS, ACT_SWISH };

#define SELU_S 1.0507009873554804934193349852946f
#define SELU_A 1.6732632423543772848170429916717f

__device__ __forceinline__ float act_value(int act, float x, float alpha) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_LRELU: return x > 0.f ? x : x * alpha;
    case ACT_TANH: {
      if (x < -80.f) return -1.f;
      if (x > 80.f) return 1.f;
      const float c = expf(x), d = 1.f / c;
      return (c - d) / (c + d);
    }
    case ACT_SIGMOID: return x < -80.f ? 0.f : 1.f / (expf(-x) + 1.f);
    case ACT_ELU: return x >= 0.f ? x : expf(x) - 1.f;
    case ACT_SELU: return x >= 0.f ? SELU_S * x : (SELU_S * SELU_A) * (expf(x) - 1.f);
    case ACT_SOFTPLUS: return x > 80.f ? x : logf(expf(x) + 1.f);
    case ACT_SWISH: return x < -80.f ? 0.f : x / (expf(-x) + 1.f);
    default: return x;
  }
}

// first derivative expressed through the forward output yy (= yref/gain) or the forward input xr (swish)
__device__ __forceinline__ float act_d1(int act, float yy, float xr, float alpha) {
  switch (act) {
    case ACT_RELU: return yy > 0.f ? 1.f : 0.f;
    case ACT_LRELU: return yy > 0.f ? 1.f : alpha;
    case ACT_TANH: return 1.f - yy * yy;
    case ACT_SIGMOID: return yy * (1.f - yy);
    case ACT_ELU: return yy >= 0.f ? 1.f : yy + 1.f;
    case ACT_SELU: return yy >= 0.f ? SELU_S : yy + SELU_S * SELU_A;
    case ACT_SOFTPLUS: return 1.f - expf(-yy);
    case ACT_SWISH: {
      if (xr > 40.f) return 1.f;
      const float c = expf(xr), d = c + 1.f;
      return c * (xr + d) / (d * d);
    }
    default: return 1.f;
  }
}

__device__ __forceinline__ float act_d2(int act, float yy, float xr) {
  switch (act) {
    case ACT_TANH: return (1.f - yy * yy) * (-2.f * yy);
    case ACT_SIGMOID: return yy * (1.f - yy) * (1.f - 2.f * yy);
    case ACT_ELU: return yy >= 0.f ? 0.f : yy + 1.f;
    case ACT_SELU: return yy >= 0.f ? 0.f : yy + SELU_S * SELU_A;
    case ACT_SOFTPLUS: {
      const float c = expf(-yy);
      return c * (1.f - c);
    }
    case ACT_SWISH: {
      if (xr > 40.f) return 0.f;
      const float c = expf(xr), d = c + 1.f;
      return c * (xr * (2.f - d) + 2.f * d) / (d * d * d);
    }
    default: return 0.f;   // linear / relu / lrelu have no second-order term
  }
}

__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                       const float* __restrict__ xref, const float* __restrict__ yref,
                                                       const float* __restrict__ dy, float* __restrict__ y, long n,
                                                       long step_b, int size_b, int grad, int act, float alpha,
                                                       float gain, float clamp) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float bias = b ? b[(i / step_b) % size_b] : 0.f;
    const float up = dy ? dy[i] : 1.f;
    float out;
    if (grad == 0) {
      out = act_value(act, x[i] + bias, alpha) * (gain * up);
      if (clamp >= 0.f) out = (out > -clamp && out < clamp) ? out : (out >= 0.f ? clamp : -clamp);
    } else {
      const float xr = (xref ? xref[i] : 0.f) + bias;
      float yr = yref ? yref[i] : 0.f;
      const float yy = gain != 0.f ? yr / gain : 0.f;
      const float d = (grad == 1) ? act_d1(act, yy, xr, alpha) : act_d2(act, yy, xr);
      out = x[i] * d * (gain * up);
      if (act == ACT_SWISH) yr = act_value(ACT_SWISH, xr, alpha) * gain;   // swish saves x, not y
      if (clamp >= 0.f) out = (yr > -clamp && yr < clamp) ? out : 0.f;
    }
    y[i] = out;
  }
}

extern "C" int icg_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy,
                            float* y, int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain,
                            float clamp, void* stream) {
  ICG_REQUIRE(x && y && n > 0 && grad >= 0 && grad <= 2 && act >= ACT_LINEAR && act <= ACT_SWISH);
  if (b) ICG_REQUIRE(step_b > 0 && size_b > 0);
  long blocks = icg_cdiv(n, 1024);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, b, xref, yref, dy,
                     y, (long)n, (long)step_b, size_b, grad, act, alpha, gain, clamp);
  return icg_check_launch();
}

// ---------------------------------------------------------------- upfirdn2d
// out[n,c,oy,ox] = gain * sum_{ty,tx} Z[oy*downy + ty - pady0][ox*downx + tx - padx0] * g[ty][tx]
//   Z = x with (up-1) zeros inserted after every sample,  g = f flipped (true convolution) unless `flip`.
__device__ __forceinline__ int ceil_div_i(int a, int b) { return (a >= 0) ? (a + b - 1) / b : -((-a) / b); }

__global__ __launch_bounds__(256) void upfirdn2d_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                        float* __restrict__ y, int NC, int H, int W, int fh, int fw,
                                                        int upx, int upy, int downx, int downy, int padx0, int pady0,
                                                        int flip, float gain, int outH, int outW) {
  const long total = (long)NC * outH * outW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % outW);
    long t = i / outW;
    const int oy = (int)(t % outH);
    const long nc = t / outH;
    const int by = oy * downy - pady0, bx = ox * downx - padx0;   // Z coordinate of tap (0,0)
    // input rows iy with  0 <= iy*upy - by < fh
    int iy0 = ceil_div_i(by, upy), iy1 = ceil_div_i(by + fh, upy);
    int ix0 = ceil_div_i(bx, upx), ix1 = ceil_div_i(bx + fw, upx);
    iy0 = max(iy0, 0); iy1 = min(iy1, H);
    ix0 = max(ix0, 0); ix1 = min(ix1, W);
    const float* xp = x + nc * (long)H * W;
    float acc = 0.f;
    for (int iy = iy0; iy < iy1; ++iy) {
      const int ty = iy * upy - by;
      const int fy = flip ? ty : fh - 1 - ty;
      for (int ix = ix0; ix < ix1; ++ix) {
        const int tx = ix * upx - bx;
        const int fx = flip ? tx : fw - 1 - tx;
        acc = fmaf(xp[(long)iy * W + ix], f[fy * fw + fx], acc);
      }
    }
    y[i] = acc * gain;
  }
}

extern "C" int icg_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw,
                             int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                             int flip, float gain, int outH, int outW, void* stream) {
  ICG_REQUIRE(x && f && y && N > 0 && C > 0 && H > 0 && W > 0 && fh >= 1 && fw >= 1);
  ICG_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1);
  ICG_REQUIRE(outW == (W * upx + padx0 + padx1 - fw + downx) / downx);
  ICG_REQUIRE(outH == (H * upy + pady0 + pady1 - fh + downy) / downy);
  ICG_REQUIRE(outW >= 1 && outH >= 1);
  const long total = (long)N * C * outH * outW;
  long blocks = icg_cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, f, y, N * C, H, W,
                     fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH, outW);
  return icg_check_launch();
}
