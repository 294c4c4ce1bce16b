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

// 16-byte version: one float4 per thread and iteration.  BMODE 0: no bias; 1: plane-major (NCHW, step_b % 4 == 0: the four
// elements share one bias); 2: channel-minor (NHWC, step_b == 1, size_b % 4 == 0: four consecutive biases).  ACT is a
// compile-time constant for the activations the networks use (linear / relu / lrelu), 0 = decided at run time.
template <int ACT, int GRAD, int BMODE>
__global__ __launch_bounds__(256) void bias_act_vec_kernel(const float4* __restrict__ x, const float* __restrict__ b,
                                                           const float4* __restrict__ xref,
                                                           const float4* __restrict__ yref, const float4* __restrict__ dy,
                                                           float4* __restrict__ y, unsigned n4, unsigned step_b4,
                                                           unsigned size_b, int act_rt, float alpha, float gain,
                                                           float clamp) {
  const int act = ACT ? ACT : act_rt;
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (BMODE == 1) {
      const float v = b[(i / step_b4) % size_b];
      bias[0] = bias[1] = bias[2] = bias[3] = v;
    } else if (BMODE == 2) {
      const float4 v = *reinterpret_cast<const float4*>(b + (i * 4u) % size_b);
      bias[0] = v.x; bias[1] = v.y; bias[2] = v.z; bias[3] = v.w;
    }
    const float4 xv4 = x[i];
    const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
    float up[4] = {1.f, 1.f, 1.f, 1.f};
    if (dy) { const float4 t = dy[i]; up[0] = t.x; up[1] = t.y; up[2] = t.z; up[3] = t.w; }
    float xr[4] = {0.f, 0.f, 0.f, 0.f}, yr[4] = {0.f, 0.f, 0.f, 0.f};
    if (GRAD != 0) {
      if (xref) { const float4 t = xref[i]; xr[0] = t.x; xr[1] = t.y; xr[2] = t.z; xr[3] = t.w; }
      if (yref) { const float4 t = yref[i]; yr[0] = t.x; yr[1] = t.y; yr[2] = t.z; yr[3] = t.w; }
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (GRAD == 0) {
        float out = act_value(act, xv[j] + bias[j], alpha) * (gain * up[j]);
        if (clamp >= 0.f) out = (out > -clamp && out < clamp) ? out : (out >= 0.f ? clamp : -clamp);
        o[j] = out;
      } else {
        const float xrj = xr[j] + bias[j];
        float yrj = yr[j];
        const float yy = gain != 0.f ? yrj / gain : 0.f;
        const float d = (GRAD == 1) ? act_d1(act, yy, xrj, alpha) : act_d2(act, yy, xrj);
        float out = xv[j] * d * (gain * up[j]);
        if (act == ACT_SWISH) yrj = act_value(ACT_SWISH, xrj, alpha) * gain;
        if (clamp >= 0.f) out = (yrj > -clamp && yrj < clamp) ? out : 0.f;
        o[j] = out;
      }
    }
    y[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

template <int ACT, int GRAD>
static void launch_bias_act_vec(int bmode, dim3 grid, hipStream_t st, const float* x, const float* b, const float* xref,
                                const float* yref, const float* dy, float* y, unsigned n4, unsigned step_b4,
                                unsigned size_b, int act, float alpha, float gain, float clamp) {
#define ICG_BA(BM)                                                                                                  \
  hipLaunchKernelGGL((bias_act_vec_kernel<ACT, GRAD, BM>), grid, dim3(256), 0, st, (const float4*)x, b,               \
                     (const float4*)xref, (const float4*)yref, (const float4*)dy, (float4*)y, n4, step_b4, size_b, act, \
                     alpha, gain, clamp)
  if (bmode == 0) ICG_BA(0); else if (bmode == 1) ICG_BA(1); else ICG_BA(2);
#undef ICG_BA
}

static bool al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int icg_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy,
                            float* y, int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain,
                            float clamp, void* stream) {
  ICG_REQUIRE(x && y && n > 0 && grad >= 0 && grad <= 2 && act >= ACT_LINEAR && act <= ACT_SWISH);
  if (b) ICG_REQUIRE(step_b > 0 && size_b > 0);
  // 16-byte path: whole tensor in float4s, a float4 never straddles a bias boundary (or lies across 4 channels, NHWC)
  int bmode = -1;
  if (!b) bmode = 0;
  else if (step_b % 4 == 0) bmode = 1;
  else if (step_b == 1 && size_b % 4 == 0 && al16(b)) bmode = 2;
  if (bmode >= 0 && n % 4 == 0 && n < 0xffffffffL && al16(x) && al16(y) && al16(xref) && al16(yref) && al16(dy)) {
    const unsigned n4 = (unsigned)(n / 4);
    long vb = icg_cdiv((long)n4, 256);
    if (vb > 256 * 16) vb = 256 * 16;
    const dim3 grid((unsigned)vb);
    hipStream_t st = (hipStream_t)stream;
    const unsigned sb4 = (unsigned)(bmode == 1 ? step_b / 4 : 1), sz = (unsigned)size_b;
#define ICG_BA_ACT(A)                                                                                               \
  do {                                                                                                              \
    if (grad == 0) launch_bias_act_vec<A, 0>(bmode, grid, st, x, b, xref, yref, dy, y, n4, sb4, sz, act, alpha, gain, clamp); \
    else if (grad == 1) launch_bias_act_vec<A, 1>(bmode, grid, st, x, b, xref, yref, dy, y, n4, sb4, sz, act, alpha, gain, clamp); \
    else launch_bias_act_vec<A, 2>(bmode, grid, st, x, b, xref, yref, dy, y, n4, sb4, sz, act, alpha, gain, clamp);   \
  } while (0)
    if (act == ACT_LINEAR) ICG_BA_ACT(ACT_LINEAR);
    else if (act == ACT_LRELU) ICG_BA_ACT(ACT_LRELU);
    else if (act == ACT_RELU) ICG_BA_ACT(ACT_RELU);
    else ICG_BA_ACT(0);
#undef ICG_BA_ACT
    return icg_check_launch();
  }
  long blocks = icg_cdiv(n, 1024);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
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

// Channels-last version: x is [N][H][W][C], y is [N][outH][outW][C], C % 4 == 0.  One thread produces a vertical strip of
// TY output pixels for 4 consecutive channels: lanes run over channels first (16-byte coalesced loads/stores), every
// loaded input row is reused by all outputs of the strip that it contributes to, the filter sits in LDS.
template <int TY>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                             float* __restrict__ y, int N, int H, int W, int C4, int fh,
                                                             int fw, int upx, int upy, int downx, int downy, int padx0,
                                                             int pady0, int flip, float gain, int outH, int outW) {
  __shared__ float fs[256];
  for (int i = threadIdx.x; i < fh * fw; i += blockDim.x) {
    const int ty = i / fw, tx = i - ty * fw;
    fs[i] = f[(flip ? ty : fh - 1 - ty) * fw + (flip ? tx : fw - 1 - tx)] * gain;   // fs[ty][tx]: plain correlation taps
  }
  __syncthreads();
  const int strips = (outH + TY - 1) / TY;
  const long total = (long)N * strips * outW * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % outW);
    t /= outW;
    const int ys = (int)(t % strips);
    const int n = (int)(t / strips);
    const int oy0 = ys * TY;
    const int bx = ox * downx - padx0;
    int ix0 = ceil_div_i(bx, upx), ix1 = ceil_div_i(bx + fw, upx);
    ix0 = max(ix0, 0); ix1 = min(ix1, W);
    const int by0 = oy0 * downy - pady0;                         // Z row of tap 0 of the strip's first output
    const int byl = (min(oy0 + TY, outH) - 1) * downy - pady0;   // ... of its last output
    int iy0 = ceil_div_i(by0, upy), iy1 = ceil_div_i(byl + fh, upy);
    iy0 = max(iy0, 0); iy1 = min(iy1, H);
    float4 acc[TY];
#pragma unroll
    for (int j = 0; j < TY; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xp = reinterpret_cast<const float4*>(x) + (long)n * H * W * C4 + c4;
    for (int iy = iy0; iy < iy1; ++iy) {
      const int zy = iy * upy - by0;                             // tap row seen by output j: zy - j*downy
      for (int ix = ix0; ix < ix1; ++ix) {
        const int tx = ix * upx - bx;
        const float4 v = xp[((long)iy * W + ix) * C4];
#pragma unroll
        for (int j = 0; j < TY; ++j) {
          const int ty = zy - j * downy;
          if (ty >= 0 && ty < fh) {
            const float w = fs[ty * fw + tx];
            acc[j].x = fmaf(v.x, w, acc[j].x); acc[j].y = fmaf(v.y, w, acc[j].y);
            acc[j].z = fmaf(v.z, w, acc[j].z); acc[j].w = fmaf(v.w, w, acc[j].w);
          }
        }
      }
    }
    float4* yp = reinterpret_cast<float4*>(y) + (((long)n * outH + oy0) * outW + ox) * C4 + c4;
#pragma unroll
    for (int j = 0; j < TY; ++j)
      if (oy0 + j < outH) yp[(long)j * outW * C4] = acc[j];
  }
}

// The case the networks run all the time — a 4x4 filter ([1,3,3,1] (x) [1,3,3,1]), no zero insertion, decimation 1 or 2 —
// fully unrolled: taps in registers, no predicates in the accumulation, (TY-1)*DOWN + 4 input rows x 4 columns per strip.
template <int DOWN, int TY>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_f4_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                                float* __restrict__ y, int N, int H, int W, int C4,
                                                                int padx0, int pady0, int flip, float gain, int outH,
                                                                int outW) {
  __shared__ float fs[16];
  if (threadIdx.x < 16) {
    const int ty = threadIdx.x >> 2, tx = threadIdx.x & 3;
    fs[threadIdx.x] = f[(flip ? ty : 3 - ty) * 4 + (flip ? tx : 3 - tx)] * gain;
  }
  __syncthreads();
  float w[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[a][c] = fs[a * 4 + c];
  constexpr int NR = (TY - 1) * DOWN + 4;
  const int strips = (outH + TY - 1) / TY;
  const long total = (long)N * strips * outW * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % outW);
    t /= outW;
    const int ys = (int)(t % strips);
    const int n = (int)(t / strips);
    const int oy0 = ys * TY;
    const int bx = ox * DOWN - padx0, by = oy0 * DOWN - pady0;
    float4 acc[TY];
#pragma unroll
    for (int j = 0; j < TY; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xp = reinterpret_cast<const float4*>(x) + (long)n * H * W * C4 + c4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ix = bx + c;
      const bool cok = (unsigned)ix < (unsigned)W;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int iy = by + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cok && (unsigned)iy < (unsigned)H) v = xp[((long)iy * W + ix) * C4];
#pragma unroll
        for (int j = 0; j < TY; ++j) {
          const int ty = r - j * DOWN;          // compile-time after unrolling
          if (ty >= 0 && ty < 4) {
            const float ww = w[ty][c];
            acc[j].x = fmaf(v.x, ww, acc[j].x); acc[j].y = fmaf(v.y, ww, acc[j].y);
            acc[j].z = fmaf(v.z, ww, acc[j].z); acc[j].w = fmaf(v.w, ww, acc[j].w);
          }
        }
      }
    }
    float4* yp = reinterpret_cast<float4*>(y) + (((long)n * outH + oy0) * outW + ox) * C4 + c4;
#pragma unroll
    for (int j = 0; j < TY; ++j)
      if (oy0 + j < outH) yp[(long)j * outW * C4] = acc[j];
  }
}

extern "C" int icg_upfirdn2d_nhwc(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw,
                                  int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                                  int flip, float gain, int outH, int outW, void* stream) {
  ICG_REQUIRE(x && f && y && N > 0 && C > 0 && (C % 4 == 0) && H > 0 && W > 0 && fh >= 1 && fw >= 1 && fh * fw <= 256);
  ICG_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1);
  ICG_REQUIRE(outW == (W * upx + padx0 + padx1 - fw + downx) / downx);
  ICG_REQUIRE(outH == (H * upy + pady0 + pady1 - fh + downy) / downy);
  ICG_REQUIRE(outW >= 1 && outH >= 1);
  ICG_REQUIRE(al16(x) && al16(y));
  constexpr int TY = 4;
  const long total = (long)N * ((outH + TY - 1) / TY) * outW * (C / 4);
  long blocks = icg_cdiv(total, 256);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
  if (fh == 4 && fw == 4 && upx == 1 && upy == 1 && downx == downy && (downx == 1 || downx == 2)) {
    if (downx == 1)
      hipLaunchKernelGGL((upfirdn2d_nhwc_f4_kernel<1, TY>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, f,
                         y, N, H, W, C / 4, padx0, pady0, flip, gain, outH, outW);
    else
      hipLaunchKernelGGL((upfirdn2d_nhwc_f4_kernel<2, TY>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, f,
                         y, N, H, W, C / 4, padx0, pady0, flip, gain, outH, outW);
    return icg_check_launch();
  }
  hipLaunchKernelGGL(upfirdn2d_nhwc_kernel<TY>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, f, y, N, H,
                     W, C / 4, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH, outW);
  return icg_check_launch();
}

// NCHW, 4x4 filter, no zero insertion, decimation 1 or 2: lanes run along x (coalesced 4-byte accesses), a thread produces
// a vertical strip of TY outputs from (TY-1)*DOWN + 4 input rows x 4 columns, taps in registers, fully unrolled.
template <int DOWN, int TY>
__global__ __launch_bounds__(256) void upfirdn2d_nchw_f4_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                                float* __restrict__ y, int NC, int H, int W, int padx0,
                                                                int pady0, int flip, float gain, int outH, int outW) {
  __shared__ float fs[16];
  if (threadIdx.x < 16) {
    const int ty = threadIdx.x >> 2, tx = threadIdx.x & 3;
    fs[threadIdx.x] = f[(flip ? ty : 3 - ty) * 4 + (flip ? tx : 3 - tx)] * gain;
  }
  __syncthreads();
  float w[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[a][c] = fs[a * 4 + c];
  constexpr int NR = (TY - 1) * DOWN + 4;
  const int strips = (outH + TY - 1) / TY;
  const long total = (long)NC * strips * outW;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int ox = (int)(i % outW);
    long t = i / outW;
    const int ys = (int)(t % strips);
    const long nc = t / strips;
    const int oy0 = ys * TY;
    const int bx = ox * DOWN - padx0, by = oy0 * DOWN - pady0;
    const float* xp = x + nc * (long)H * W;
    float acc[TY];
#pragma unroll
    for (int j = 0; j < TY; ++j) acc[j] = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int iy = by + r;
      const bool rok = (unsigned)iy < (unsigned)H;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ix = bx + c;
        float v = 0.f;
        if (rok && (unsigned)ix < (unsigned)W) v = xp[(long)iy * W + ix];
#pragma unroll
        for (int j = 0; j < TY; ++j) {
          const int ty = r - j * DOWN;
          if (ty >= 0 && ty < 4) acc[j] = fmaf(v, w[ty][c], acc[j]);
        }
      }
    }
    float* yp = y + (nc * outH + oy0) * (long)outW + ox;
#pragma unroll
    for (int j = 0; j < TY; ++j)
      if (oy0 + j < outH) yp[(long)j * outW] = acc[j];
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
  if (fh == 4 && fw == 4 && upx == 1 && upy == 1 && downx == downy && (downx == 1 || downx == 2)) {
    constexpr int TY = 4;
    const long tot = (long)N * C * ((outH + TY - 1) / TY) * outW;
    long nb = icg_cdiv(tot, 256);
    if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
    if (downx == 1)
      hipLaunchKernelGGL((upfirdn2d_nchw_f4_kernel<1, TY>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, f, y,
                         N * C, H, W, padx0, pady0, flip, gain, outH, outW);
    else
      hipLaunchKernelGGL((upfirdn2d_nchw_f4_kernel<2, TY>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, f, y,
                         N * C, H, W, padx0, pady0, flip, gain, outH, outW);
    return icg_check_launch();
  }
  const long total = (long)N * C * outH * outW;
  long blocks = icg_cdiv(total, 256);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, f, y, N * C, H, W,
                     fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH, outW);
  return icg_check_launch();
}

// ---------------------------------------------------------------- gradient sanitising of the training loop, all parameters at once
// training_loop.py:511-515 runs torch.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad) per parameter: 144
// launches per iteration at cfg4 for 60 MB of gradients, in a loop that is bound by the host.  Same packing as icg_adam_multi
// (optim.hip): up to NTN_MAX tensors per launch, descriptors in the kernel arguments, a block owns NTN_CHUNK elements of one tensor.
#define NTN_MAX 64
#define NTN_CHUNK 4096
struct NtnPack {
  icg_f32_buffer t[NTN_MAX];
  int blk_start[NTN_MAX + 1];
  int n;
};
__device__ __forceinline__ float ntn_one(float v, float nan_v, float posinf, float neginf) {
  // isnan / isinf spelled on the bits: immune to any finite-math assumption of the build
  const unsigned u = __float_as_uint(v), mag = u & 0x7fffffffu;
  if (mag > 0x7f800000u) return nan_v;
  if (mag == 0x7f800000u) return (u >> 31) ? neginf : posinf;
  return v;
}
__global__ __launch_bounds__(256) void nan_to_num_multi_kernel(NtnPack p, float nan_v, float posinf, float neginf) {
  int lo = 0, hi = p.n - 1;                       // the tensor whose block range holds blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= p.blk_start[mid]) lo = mid; else hi = mid - 1;
  }
  const icg_f32_buffer t = p.t[lo];
  const long base = (long)((int)blockIdx.x - p.blk_start[lo]) * NTN_CHUNK;
  const long end = min((long)t.numel, base + NTN_CHUNK);
  if ((((uintptr_t)t.data & 15) == 0) && end - base == NTN_CHUNK) {
#pragma unroll
    for (int u = 0; u < NTN_CHUNK / 4 / 256; ++u) {
      float4* q = reinterpret_cast<float4*>(t.data) + (base >> 2) + threadIdx.x + u * 256;
      float4 v = *q;
      v.x = ntn_one(v.x, nan_v, posinf, neginf); v.y = ntn_one(v.y, nan_v, posinf, neginf);
      v.z = ntn_one(v.z, nan_v, posinf, neginf); v.w = ntn_one(v.w, nan_v, posinf, neginf);
      *q = v;
    }
    return;
  }
  for (long i = base + threadIdx.x; i < end; i += 256) t.data[i] = ntn_one(t.data[i], nan_v, posinf, neginf);
}

extern "C" int icg_nan_to_num_multi(const icg_f32_buffer* tensors, int n, float nan_v, float posinf, float neginf, void* stream) {
  ICG_REQUIRE(tensors && n >= 0);
  int done = 0;
  while (done < n) {
    NtnPack p;
    p.n = 0;
    int blocks = 0;
    while (done < n && p.n < NTN_MAX) {
      const icg_f32_buffer& t = tensors[done++];
      if (t.numel <= 0) continue;
      ICG_REQUIRE(t.data);
      p.t[p.n] = t;
      p.blk_start[p.n] = blocks;
      blocks += (int)icg_cdiv(t.numel, NTN_CHUNK);
      p.n++;
    }
    if (p.n == 0) break;
    p.blk_start[p.n] = blocks;
    hipLaunchKernelGGL(nan_to_num_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, nan_v, posinf, neginf);
    const int rc = icg_check_launch();
    if (rc != ICG_OK) return rc;
  }
  return ICG_OK;
}
