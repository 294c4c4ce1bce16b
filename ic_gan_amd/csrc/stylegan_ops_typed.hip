// StyleGAN2 custom ops, fp16 / fp64 STORAGE variants of bias_act and upfirdn2d for gfx950.
//
// The reference plugins are templated on the tensor dtype (bias_act.cu:155-167 `choose_bias_act_kernel`: half, float, double;
// upfirdn2d.cu:208-344 likewise) and compute in `InternalType<T>::scalar_t` -- fp32 for half and float, fp64 for double
// (bias_act.cu:18-21, upfirdn2d.cu:17-20): one rounding to the storage type at the end.  Same here.  The fp32 entry points
// (stylegan_ops.hip) stay as they are; `dtype` = 0 forwards to them, 1 = fp16, 2 = fp64.  The filter of upfirdn2d is always
// fp32 (upfirdn2d.cpp:27).  fp16 is the storage type of the reference's `num_fp16_res` blocks (training/networks.py:505-515);
// both ops are HBM-bound, so halving the bytes is the whole point: 16-byte accesses = 8 halves per lane.
#include "icg_common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

extern "C" int icg_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* y,
                            int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain, float clamp,
                            void* stream);
extern "C" int icg_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw, int upx,
                             int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                             int outH, int outW, void* stream);
extern "C" int icg_upfirdn2d_nhwc(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw,
                                  int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip,
                                  float gain, int outH, int outW, void* stream);

enum { T_ACT_LINEAR = 1, T_ACT_RELU, T_ACT_LRELU, T_ACT_TANH, T_ACT_SIGMOID, T_ACT_ELU, T_ACT_SELU, T_ACT_SOFTPLUS, T_ACT_SWISH };
enum { DT_F32 = 0, DT_F16 = 1, DT_F64 = 2 };

__device__ __forceinline__ float xexp(float v) { return expf(v); }
__device__ __forceinline__ double xexp(double v) { return exp(v); }
__device__ __forceinline__ float xlog(float v) { return logf(v); }
__device__ __forceinline__ double xlog(double v) { return log(v); }

template <typename S> struct SeluC {
  static constexpr S s = (S)1.0507009873554804934193349852946;
  static constexpr S a = (S)1.6732632423543772848170429916717;
};

// the activation table of bias_act.py:25-106 in the compute type S (float for fp16 storage, double for fp64)
template <typename S> __device__ __forceinline__ S t_act_value(int act, S x, S alpha) {
  switch (act) {
    case T_ACT_RELU: return x > (S)0 ? x : (S)0;
    case T_ACT_LRELU: return x > (S)0 ? x : x * alpha;
    case T_ACT_TANH: {
      if (x < (S)-80) return (S)-1;
      if (x > (S)80) return (S)1;
      const S c = xexp(x), d = (S)1 / c;
      return (c - d) / (c + d);
    }
    case T_ACT_SIGMOID: return x < (S)-80 ? (S)0 : (S)1 / (xexp(-x) + (S)1);
    case T_ACT_ELU: return x >= (S)0 ? x : xexp(x) - (S)1;
    case T_ACT_SELU: return x >= (S)0 ? SeluC<S>::s * x : (SeluC<S>::s * SeluC<S>::a) * (xexp(x) - (S)1);
    case T_ACT_SOFTPLUS: return x > (S)80 ? x : xlog(xexp(x) + (S)1);
    case T_ACT_SWISH: return x < (S)-80 ? (S)0 : x / (xexp(-x) + (S)1);
    default: return x;
  }
}
template <typename S> __device__ __forceinline__ S t_act_d1(int act, S yy, S xr, S alpha) {
  switch (act) {
    case T_ACT_RELU: return yy > (S)0 ? (S)1 : (S)0;
    case T_ACT_LRELU: return yy > (S)0 ? (S)1 : alpha;
    case T_ACT_TANH: return (S)1 - yy * yy;
    case T_ACT_SIGMOID: return yy * ((S)1 - yy);
    case T_ACT_ELU: return yy >= (S)0 ? (S)1 : yy + (S)1;
    case T_ACT_SELU: return yy >= (S)0 ? SeluC<S>::s : yy + SeluC<S>::s * SeluC<S>::a;
    case T_ACT_SOFTPLUS: return (S)1 - xexp(-yy);
    case T_ACT_SWISH: {
      if (xr > (S)40) return (S)1;
      const S c = xexp(xr), d = c + (S)1;
      return c * (xr + d) / (d * d);
    }
    default: return (S)1;
  }
}
template <typename S> __device__ __forceinline__ S t_act_d2(int act, S yy, S xr) {
  switch (act) {
    case T_ACT_TANH: return ((S)1 - yy * yy) * ((S)-2 * yy);
    case T_ACT_SIGMOID: return yy * ((S)1 - yy) * ((S)1 - (S)2 * yy);
    case T_ACT_ELU: return yy >= (S)0 ? (S)0 : yy + (S)1;
    case T_ACT_SELU: return yy >= (S)0 ? (S)0 : yy + SeluC<S>::s * SeluC<S>::a;
    case T_ACT_SOFTPLUS: {
      const S c = xexp(-yy);
      return c * ((S)1 - c);
    }
    case T_ACT_SWISH: {
      if (xr > (S)40) return (S)0;
      const S c = xexp(xr), d = c + (S)1;
      return c * (xr * ((S)2 - d) + (S)2 * d) / (d * d * d);
    }
    default: return (S)0;
  }
}

template <typename T> struct Cvt;
template <> struct Cvt<__half> {
  typedef float S;
  static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
  static __device__ __forceinline__ void st(__half* p, float v) { *p = __float2half(v); }   // round to nearest even
};
template <> struct Cvt<double> {
  typedef double S;
  static __device__ __forceinline__ double ld(const double* p) { return *p; }
  static __device__ __forceinline__ void st(double* p, double v) { *p = v; }
};
template <> struct Cvt<float> {
  typedef float S;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};

// VEC consecutive elements of 16 bytes (8 halves / 2 doubles) <-> the compute type
template <typename T, int VEC> struct Pack;
template <> struct Pack<__half, 8> {
  static __device__ __forceinline__ void ld(const __half* p, float v[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const __half2* h = reinterpret_cast<const __half2*>(&r);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float2 f = __half22float2(h[j]); v[2 * j] = f.x; v[2 * j + 1] = f.y; }
  }
  static __device__ __forceinline__ void st(__half* p, const float v[8]) {
    uint4 r;
    __half2* h = reinterpret_cast<__half2*>(&r);
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = __floats2half2_rn(v[2 * j], v[2 * j + 1]);
    *reinterpret_cast<uint4*>(p) = r;
  }
};
template <> struct Pack<double, 2> {
  static __device__ __forceinline__ void ld(const double* p, double v[2]) {
    const double2 r = *reinterpret_cast<const double2*>(p);
    v[0] = r.x; v[1] = r.y;
  }
  static __device__ __forceinline__ void st(double* p, const double v[2]) { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); }
};
template <typename T> struct Pack<T, 1> {
  typedef typename Cvt<T>::S S;
  static __device__ __forceinline__ void ld(const T* p, S v[1]) { v[0] = Cvt<T>::ld(p); }
  static __device__ __forceinline__ void st(T* p, const S v[1]) { Cvt<T>::st(p, v[0]); }
};

template <typename S>
__device__ __forceinline__ S bias_act_one(int grad, int act, S xv, S bias, S xr0, S yr, S up, S alpha, S gain, S clamp) {
  S out;
  if (grad == 0) {
    out = t_act_value<S>(act, xv + bias, alpha) * (gain * up);
    if (clamp >= (S)0) out = (out > -clamp && out < clamp) ? out : (out >= (S)0 ? clamp : -clamp);
  } else {
    const S xr = xr0 + bias;
    const S yy = gain != (S)0 ? yr / gain : (S)0;
    const S d = (grad == 1) ? t_act_d1<S>(act, yy, xr, alpha) : t_act_d2<S>(act, yy, xr);
    out = xv * d * (gain * up);
    if (act == T_ACT_SWISH) yr = t_act_value<S>(T_ACT_SWISH, xr, alpha) * gain;   // swish saves x, not y
    if (clamp >= (S)0) out = (yr > -clamp && yr < clamp) ? out : (S)0;
  }
  return out;
}

// BMODE 0: no bias; 1: the VEC elements of a pack share one bias (step_b % VEC == 0); 2: VEC consecutive biases (channel-minor,
// step_b == 1, size_b % VEC == 0); 3: general index per element (VEC = 1 only).
template <typename T, int VEC, int BMODE>
__global__ __launch_bounds__(256) void bias_act_typed_kernel(const T* __restrict__ x, const T* __restrict__ b,
                                                             const T* __restrict__ xref, const T* __restrict__ yref,
                                                             const T* __restrict__ dy, T* __restrict__ y, long npack,
                                                             long step_bp, int size_b, int grad, int act, float alpha_f,
                                                             float gain_f, float clamp_f) {
  typedef typename Cvt<T>::S S;
  const S alpha = (S)alpha_f, gain = (S)gain_f, clamp = (S)clamp_f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npack; i += stride) {
    S bias[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bias[j] = (S)0;
    if (BMODE == 1 || BMODE == 3) {
      const S v = Cvt<T>::ld(b + (i / step_bp) % size_b);
#pragma unroll
      for (int j = 0; j < VEC; ++j) bias[j] = v;
    } else if (BMODE == 2) {
      Pack<T, VEC>::ld(b + (i * VEC) % size_b, bias);
    }
    S xv[VEC], up[VEC], xr[VEC], yr[VEC], o[VEC];
    Pack<T, VEC>::ld(x + i * VEC, xv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { up[j] = (S)1; xr[j] = (S)0; yr[j] = (S)0; }
    if (dy) Pack<T, VEC>::ld(dy + i * VEC, up);
    if (grad != 0) {
      if (xref) Pack<T, VEC>::ld(xref + i * VEC, xr);
      if (yref) Pack<T, VEC>::ld(yref + i * VEC, yr);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = bias_act_one<S>(grad, act, xv[j], bias[j], xr[j], yr[j], up[j], alpha, gain, clamp);
    Pack<T, VEC>::st(y + i * VEC, o);
  }
}

static bool t_al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int VEC>
static int launch_bias_act_typed(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                                 int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain, float clamp,
                                 hipStream_t st) {
  int bmode = -1;
  bool vec = (n % VEC == 0) && t_al16(x) && t_al16(y) && t_al16(xref) && t_al16(yref) && t_al16(dy);
  if (!b) bmode = 0;
  else if (step_b % VEC == 0) bmode = 1;
  else if (step_b == 1 && size_b % VEC == 0 && t_al16(b)) bmode = 2;
  if (!vec || bmode < 0) {       // general form, one element per thread
    long blocks = icg_cdiv((long)n, 256);
    if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
    if (b)
      hipLaunchKernelGGL((bias_act_typed_kernel<T, 1, 3>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (const T*)b,
                         (const T*)xref, (const T*)yref, (const T*)dy, (T*)y, (long)n, (long)step_b, size_b, grad, act, alpha,
                         gain, clamp);
    else
      hipLaunchKernelGGL((bias_act_typed_kernel<T, 1, 0>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (const T*)b,
                         (const T*)xref, (const T*)yref, (const T*)dy, (T*)y, (long)n, 1L, 1, grad, act, alpha, gain, clamp);
    return icg_check_launch();
  }
  const long npack = n / VEC;
  long blocks = icg_cdiv(npack, 256);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
#define ICG_TBA(BM)                                                                                                       \
  hipLaunchKernelGGL((bias_act_typed_kernel<T, VEC, BM>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (const T*)b, \
                     (const T*)xref, (const T*)yref, (const T*)dy, (T*)y, npack, (long)(BM == 1 ? step_b / VEC : 1), size_b,  \
                     grad, act, alpha, gain, clamp)
  if (bmode == 0) ICG_TBA(0); else if (bmode == 1) ICG_TBA(1); else ICG_TBA(2);
#undef ICG_TBA
  return icg_check_launch();
}

extern "C" int icg_bias_act_typed(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                                  int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain,
                                  float clamp, int dtype, void* stream) {
  ICG_REQUIRE(x && y && n > 0 && grad >= 0 && grad <= 2 && act >= T_ACT_LINEAR && act <= T_ACT_SWISH);
  ICG_REQUIRE(dtype == DT_F32 || dtype == DT_F16 || dtype == DT_F64);
  if (b) ICG_REQUIRE(step_b > 0 && size_b > 0);
  if (dtype == DT_F32)
    return icg_bias_act((const float*)x, (const float*)b, (const float*)xref, (const float*)yref, (const float*)dy, (float*)y,
                        n, step_b, size_b, grad, act, alpha, gain, clamp, stream);
  if (dtype == DT_F16)
    return launch_bias_act_typed<__half, 8>(x, b, xref, yref, dy, y, n, step_b, size_b, grad, act, alpha, gain, clamp,
                                            (hipStream_t)stream);
  return launch_bias_act_typed<double, 2>(x, b, xref, yref, dy, y, n, step_b, size_b, grad, act, alpha, gain, clamp,
                                          (hipStream_t)stream);
}

// ---------------------------------------------------------------- upfirdn2d
__device__ __forceinline__ int t_ceil_div(int a, int b) { return (a >= 0) ? (a + b - 1) / b : -((-a) / b); }

// NCHW, any filter / up / down: one output element per thread, accumulation in S (upfirdn2d.cu:32-95 `upfirdn2d_kernel_large`)
template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_typed_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                              T* __restrict__ y, int NC, int H, int W, int fh, int fw, int upx,
                                                              int upy, int downx, int downy, int padx0, int pady0, int flip,
                                                              float gain, int outH, int outW) {
  typedef typename Cvt<T>::S S;
  const long total = (long)NC * outH * outW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % outW);
    long t = i / outW;
    const int oy = (int)(t % outH);
    const long nc = t / outH;
    const int by = oy * downy - pady0, bx = ox * downx - padx0;
    int iy0 = t_ceil_div(by, upy), iy1 = t_ceil_div(by + fh, upy);
    int ix0 = t_ceil_div(bx, upx), ix1 = t_ceil_div(bx + fw, upx);
    iy0 = max(iy0, 0); iy1 = min(iy1, H);
    ix0 = max(ix0, 0); ix1 = min(ix1, W);
    const T* xp = x + nc * (long)H * W;
    S acc = (S)0;
    for (int iy = iy0; iy < iy1; ++iy) {
      const int ty = iy * upy - by;
      const int fy = flip ? ty : fh - 1 - ty;
      for (int ix = ix0; ix < ix1; ++ix) {
        const int tx = ix * upx - bx;
        const int fx = flip ? tx : fw - 1 - tx;
        acc += Cvt<T>::ld(xp + (long)iy * W + ix) * (S)f[fy * fw + fx];
      }
    }
    Cvt<T>::st(y + i, acc * (S)gain);
  }
}

// channels-last: x [N][H][W][C], y [N][outH][outW][C]; a thread owns VEC consecutive channels (16 bytes) of a vertical strip
// of TY outputs; filter taps (pre-flipped, gain folded in) in LDS
template <typename T, int VEC, int TY>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_typed_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                                   T* __restrict__ y, int N, int H, int W, int CV, int fh,
                                                                   int fw, int upx, int upy, int downx, int downy, int padx0,
                                                                   int pady0, int flip, float gain, int outH, int outW) {
  typedef typename Cvt<T>::S S;
  __shared__ float fs[256];
  for (int i = threadIdx.x; i < fh * fw; i += blockDim.x) {
    const int ty = i / fw, tx = i - ty * fw;
    fs[i] = f[(flip ? ty : fh - 1 - ty) * fw + (flip ? tx : fw - 1 - tx)] * gain;
  }
  __syncthreads();
  const int strips = (outH + TY - 1) / TY;
  const long total = (long)N * strips * outW * CV;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int cv = (int)(i % CV);
    long t = i / CV;
    const int ox = (int)(t % outW);
    t /= outW;
    const int ys = (int)(t % strips);
    const int n = (int)(t / strips);
    const int oy0 = ys * TY;
    const int bx = ox * downx - padx0;
    int ix0 = t_ceil_div(bx, upx), ix1 = t_ceil_div(bx + fw, upx);
    ix0 = max(ix0, 0); ix1 = min(ix1, W);
    const int by0 = oy0 * downy - pady0;
    const int byl = (min(oy0 + TY, outH) - 1) * downy - pady0;
    int iy0 = t_ceil_div(by0, upy), iy1 = t_ceil_div(byl + fh, upy);
    iy0 = max(iy0, 0); iy1 = min(iy1, H);
    S acc[TY][VEC];
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[j][e] = (S)0;
    const T* xp = x + ((long)n * H * W * CV + cv) * VEC;
    for (int iy = iy0; iy < iy1; ++iy) {
      const int zy = iy * upy - by0;
      for (int ix = ix0; ix < ix1; ++ix) {
        const int tx = ix * upx - bx;
        S v[VEC];
        Pack<T, VEC>::ld(xp + ((long)iy * W + ix) * CV * VEC, v);
#pragma unroll
        for (int j = 0; j < TY; ++j) {
          const int ty = zy - j * downy;
          if (ty >= 0 && ty < fh) {
            const S w = (S)fs[ty * fw + tx];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[j][e] += v[e] * w;
          }
        }
      }
    }
    T* yp = y + ((((long)n * outH + oy0) * outW + ox) * CV + cv) * VEC;
#pragma unroll
    for (int j = 0; j < TY; ++j)
      if (oy0 + j < outH) Pack<T, VEC>::st(yp + (long)j * outW * CV * VEC, acc[j]);
  }
}

// Channels-last FIR without resampling (up = down = 1, filter <= 4 x 4: the 4 x 4 blur after an up-sampling convolution, before a
// down-sampling one, and their adjoints -- 43 launches per cfg4 iteration) through an LDS tile, as upfirdn2d.cu:100-203 does for the
// NCHW layout.  The kernel above has every thread fetch its (TY + 3) x 4 input vectors itself: 7 16-byte L1 requests per 16 output
// bytes (0.32 - 0.46 of the HBM peak, profiles/r05_hbm_kernels_microbench.txt).  Measured gain of the tile: 8 % on the fp16 256 x 256
// blur (0.104 -> 0.096 ms) -- the op is co-bound by the VALU: 128 FMAs + 28 fp16 -> fp32 conversions per 16 output bytes.  A workgroup owns a
// 16 x 8 pixel tile of 8 channel vectors (8 x 16 bytes = 128 contiguous bytes per pixel): the (16 + 3) x (8 + 3) input pixels are
// loaded once (1.6 loads per output instead of 7), zero outside the image, and every thread reads its 7 x 4 window from LDS
// (consecutive lanes = consecutive 16-byte vectors: conflict-free) for a vertical strip of 4 outputs.  Taps are accumulated in the
// order of the kernel above (rows, then columns), so finite inputs give the same bits.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upfirdn2d_nhwc_tile_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                                  T* __restrict__ y, int N, int H, int W, int CV, int fh, int fw,
                                                                  int padx0, int pady0, int flip, float gain, int outH, int outW,
                                                                  int tiles_x, int tiles_y) {
  typedef typename Cvt<T>::S S;
  constexpr int TW = 16, TH = 8, IW = TW + 3, IH = TH + 3, CG = 8;
  __shared__ float fs[16];
  __shared__ __attribute__((aligned(16))) char tile[IH * IW * CG * 16];
  if (threadIdx.x < 16) {
    const int ty = threadIdx.x >> 2, tx = threadIdx.x & 3;
    fs[threadIdx.x] = (ty < fh && tx < fw) ? f[(flip ? ty : fh - 1 - ty) * fw + (flip ? tx : fw - 1 - tx)] * gain : 0.f;
  }
  const int cgroups = CV / CG;
  unsigned t = blockIdx.x;
  const int cg = (int)(t % (unsigned)cgroups);
  t /= (unsigned)cgroups;
  const int tx_ = (int)(t % (unsigned)tiles_x);
  t /= (unsigned)tiles_x;
  const int ty_ = (int)(t % (unsigned)tiles_y);
  const int n = (int)(t / (unsigned)tiles_y);
  const int ox0 = tx_ * TW, oy0 = ty_ * TH;
  const int ix0 = ox0 - padx0, iy0 = oy0 - pady0;
  const T* xn = x + ((long)n * H * W * CV + (long)cg * CG) * VEC;
  for (int i = threadIdx.x; i < IH * IW * CG; i += 256) {
    const int cv = i & (CG - 1), pc = i >> 3;
    const int c = pc % IW, r = pc / IW;
    const int iy = iy0 + r, ix = ix0 + c;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(xn + (((long)iy * W + ix) * CV + cv) * VEC);
    reinterpret_cast<uint4*>(tile)[i] = v;
  }
  __syncthreads();
  const int cv = threadIdx.x & 7, px = (threadIdx.x >> 3) & 15, ys = threadIdx.x >> 7;
  S acc[4][VEC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[j][e] = (S)0;
  const T* tp = reinterpret_cast<const T*>(tile) + (((4 * ys) * IW + px) * CG + cv) * VEC;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      S v[VEC];
      Pack<T, VEC>::ld(tp + ((r * IW + c) * CG) * VEC, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ty = r - j;
        if (ty >= 0 && ty < 4) {
          const S w = (S)fs[ty * 4 + c];
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[j][e] += v[e] * w;
        }
      }
    }
  }
  const int ox = ox0 + px;
  if (ox >= outW) return;
  T* yp = y + ((((long)n * outH + oy0 + 4 * ys) * outW + ox) * CV + (long)cg * CG + cv) * VEC;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (oy0 + 4 * ys + j < outH) Pack<T, VEC>::st(yp + (long)j * outW * CV * VEC, acc[j]);
}

template <typename T, int VEC>
static int launch_upfirdn2d_typed(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx,
                                  int upy, int downx, int downy, int padx0, int pady0, int flip, float gain, int outH, int outW,
                                  int channels_last, hipStream_t st) {
  if (channels_last) {
    ICG_REQUIRE(C % VEC == 0 && t_al16(x) && t_al16(y) && fh * fw <= 256);
    static const bool tiled = [] { const char* e = getenv("ICG_FIR_TILE"); return !(e && e[0] == '0'); }();      // measurement switch
    if (tiled && upx == 1 && upy == 1 && downx == 1 && downy == 1 && fh <= 4 && fw <= 4 && C % (8 * VEC) == 0) {
      const int tiles_x = (outW + 15) / 16, tiles_y = (outH + 7) / 8;
      const long blocks = (long)N * tiles_y * tiles_x * (C / (8 * VEC));
      ICG_REQUIRE(blocks < 0x7fffffffL);
      hipLaunchKernelGGL((upfirdn2d_nhwc_tile_kernel<T, VEC>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, f, (T*)y, N, H, W,
                         C / VEC, fh, fw, padx0, pady0, flip, gain, outH, outW, tiles_x, tiles_y);
      return icg_check_launch();
    }
    constexpr int TY = 4;
    const long total = (long)N * ((outH + TY - 1) / TY) * outW * (C / VEC);
    long blocks = icg_cdiv(total, 256);
    if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
    hipLaunchKernelGGL((upfirdn2d_nhwc_typed_kernel<T, VEC, TY>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, f, (T*)y,
                       N, H, W, C / VEC, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH, outW);
    return icg_check_launch();
  }
  const long total = (long)N * C * outH * outW;
  long blocks = icg_cdiv(total, 256);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
  hipLaunchKernelGGL((upfirdn2d_typed_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, f, (T*)y, N * C, H, W, fh,
                     fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH, outW);
  return icg_check_launch();
}

extern "C" int icg_upfirdn2d_typed(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx,
                                   int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip,
                                   float gain, int outH, int outW, int dtype, int channels_last, void* stream) {
  ICG_REQUIRE(x && f && y && N > 0 && C > 0 && H > 0 && W > 0 && fh >= 1 && fw >= 1);
  ICG_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1);
  ICG_REQUIRE(dtype == DT_F32 || dtype == DT_F16 || dtype == DT_F64);
  ICG_REQUIRE(outW == (W * upx + padx0 + padx1 - fw + downx) / downx);
  ICG_REQUIRE(outH == (H * upy + pady0 + pady1 - fh + downy) / downy);
  ICG_REQUIRE(outW >= 1 && outH >= 1);
  if (dtype == DT_F32) {
    if (channels_last)
      return icg_upfirdn2d_nhwc((const float*)x, f, (float*)y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0,
                                pady1, flip, gain, outH, outW, stream);
    return icg_upfirdn2d((const float*)x, f, (float*)y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
                         flip, gain, outH, outW, stream);
  }
  if (dtype == DT_F16)
    return launch_upfirdn2d_typed<__half, 8>(x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH,
                                             outW, channels_last, (hipStream_t)stream);
  return launch_upfirdn2d_typed<double, 2>(x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, outH,
                                           outW, channels_last, (hipStream_t)stream);
}


// ---- column sums of an fp16 channels-last tensor: out[c] = sum_rows x[row][c] in fp32 (the bias gradient of bias_act in the fp16
// blocks, bias_act.py:127 `db = dx.sum(...)`: 80 torch reductions of 30 - 80 us per cfg4 iteration).  C = 8 V, V a power of two <= 256:
// thread t owns the 16-byte vector t % V of the rows t / V + k (256 / V), 8 fp32 sums each; fixed-order combine over the block's row
// lanes in LDS, per-block partials in the caller's workspace, fixed-order sum over the blocks in the second kernel.
__global__ __launch_bounds__(256) void colsum_f16_partial_kernel(const __half* __restrict__ x, long rows, int V, int rows_per_block,
                                                                 float* __restrict__ part) {
  __shared__ float red[256][8];
  const int v = threadIdx.x % V, rl = threadIdx.x / V, nrl = 256 / V;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (long r = r0 + rl; r < r1; r += nrl) {
    float f[8];
    Pack<__half, 8>::ld(x + (r * V + v) * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < V * 8; c += 256) {  // channel c = (vector v2, component j): sum over the row lanes in a fixed order
    const int v2 = c >> 3, j = c & 7;
    float s = 0.f;
    for (int k = 0; k < nrl; ++k) s += red[k * V + v2][j];
    part[(long)blockIdx.x * (V * 8) + c] = s;
  }
}
__global__ __launch_bounds__(256) void colsum_f16_final_kernel(const float* __restrict__ part, int nblocks, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 1 < nblocks; b += 2) { s0 += part[(long)b * C + c]; s1 += part[(long)(b + 1) * C + c]; }
  if (b < nblocks) s0 += part[(long)b * C + c];
  out[c] = s0 + s1;
}

static int colsum_f16_blocks(int64_t rows) {
  long nb = icg_cdiv(rows, 256);                   // >= 256 rows per block ...
  if (nb > 2048) nb = 2048;                        // ... and at most 2048 blocks (8 per CU)
  return (int)(nb < 1 ? 1 : nb);
}
extern "C" int icg_colsum_f16_applies(int C) {
  if (C < 8 || C % 8 != 0) return 0;
  const int V = C / 8;
  return (V <= 256 && (V & (V - 1)) == 0) ? 1 : 0;
}
extern "C" size_t icg_colsum_f16_workspace_bytes(int64_t rows, int C) {
  return (rows > 0 && C > 0) ? (size_t)colsum_f16_blocks(rows) * C * sizeof(float) : 0;
}
extern "C" int icg_colsum_f16(const void* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && out && workspace && rows > 0 && icg_colsum_f16_applies(C));
  ICG_REQUIRE(((uintptr_t)x % 16) == 0 && workspace_bytes >= icg_colsum_f16_workspace_bytes(rows, C));
  const int nb = colsum_f16_blocks(rows), V = C / 8;
  const int rpb = (int)icg_cdiv(rows, nb);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_f16_partial_kernel, dim3(nb), dim3(256), 0, st, (const __half*)x, (long)rows, V, rpb, (float*)workspace);
  hipLaunchKernelGGL(colsum_f16_final_kernel, dim3((unsigned)icg_cdiv(C, 256)), dim3(256), 0, st, (const float*)workspace, nb, C, out);
  return icg_check_launch();
}
