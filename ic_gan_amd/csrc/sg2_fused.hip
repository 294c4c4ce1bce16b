// Fused StyleGAN2 layer kernels: everything a synthesis layer / toRGB layer / discriminator convolution does AROUND
// its contraction -- weight preparation, style modulation, demodulation, noise, bias, activation, clamp and all their first-order
// gradients -- as a handful of launches per layer and pass, where the reference (and rounds 2 - 4 of this repo) issue ~35 elementwise
// framework operations (stylegan2_ada_pytorch/training/networks.py:37-117 modulated_conv2d, 361-444 SynthesisLayer, 450-486
// ToRGBLayer, 171-242 Conv2dLayer, 121-165 FullyConnectedLayer).
//
// Arithmetic contract: fp16 storage keeps the reference's rounding points -- every tensor the reference materialises in fp16 (x * s,
// the convolution output, x * d + noise, the activation, and their gradients) is rounded to fp16 at the same place here, inside
// registers; only the reductions (bias / style / demodulation / noise-strength gradients), which the reference sums in fp16 storage,
// are kept in fp32.  fp32 storage has no intermediate roundings at all.
//
// Layout: activations [N][HW][C] (NHWC memory), per-sample vectors (styles, demodulation coefficients) fp32 [N][C].
// All HBM-bound: 16-byte accesses, one pass over each activation tensor per kernel, fixed-order reductions (per-block partial sums
// in the caller's workspace + one final kernel; no atomics).
#include "icg_common.h"
#include <hip/hip_fp16.h>
#include <math.h>

namespace {

template <typename T> struct Sg;
template <> struct Sg<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void ld(const float* p, float v[4]) {
    const float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
  static __device__ __forceinline__ float rnd(float v) { return v; }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Sg<__half> {
  static constexpr int VEC = 8;
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
  static __device__ __forceinline__ float rnd(float v) { return __half2float(__float2half_rn(v)); }
  static __device__ __forceinline__ float ld1(const __half* p) { return __half2float(*p); }
  static __device__ __forceinline__ void st1(__half* p, float v) { *p = __float2half_rn(v); }
};

__device__ __forceinline__ float block_sum_256(float v, float* sh /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------------------------------------------
// weight preparation (one launch pair for ALL layers of a network)
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int SGW_MAX = 40;
struct SgwPack {
  icg_sg2_weight t[SGW_MAX];
  int blk0[SGW_MAX + 1];
  int n;
};

// block per (layer, output channel): scale[o] (pre-normalisation or gain), argmax, wsq[o][i]
__global__ __launch_bounds__(256) void sg2_wprep_rows_kernel(SgwPack p) {
  __shared__ float sh_m[4];
  __shared__ int sh_i[4];
  int li = 0;
  while (li + 1 < p.n && (int)blockIdx.x >= p.blk0[li + 1]) ++li;
  const icg_sg2_weight L = p.t[li];
  const int o = blockIdx.x - p.blk0[li];
  const int RR = L.R * L.R, len = L.I * RR;
  const float* w = L.w + (size_t)o * len;
  float scale = L.gain;
  if (L.prenorm) {
    float m = -1.f;
    int mi = 0;
    for (int e = threadIdx.x; e < len; e += 256) {
      const float a = fabsf(w[e]);
      if (a > m) { m = a; mi = e; }
    }
    // first index of the maximum (torch's norm(inf) gradient goes to the maxima; ties have measure zero)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float m2 = __shfl_xor(m, off, 64);
      const int i2 = __shfl_xor(mi, off, 64);
      if (m2 > m || (m2 == m && i2 < mi)) { m = m2; mi = i2; }
    }
    if ((threadIdx.x & 63) == 0) { sh_m[threadIdx.x >> 6] = m; sh_i[threadIdx.x >> 6] = mi; }
    __syncthreads();
    m = sh_m[0]; mi = sh_i[0];
    for (int k = 1; k < 4; ++k)
      if (sh_m[k] > m || (sh_m[k] == m && sh_i[k] < mi)) { m = sh_m[k]; mi = sh_i[k]; }
    scale = (1.f / m) * L.gain;                           // gain = fp32(1 / sqrt(I R R)) here (networks.py:57-63: c0 / norm = reciprocal * c0)
    if (threadIdx.x == 0 && L.warg) L.warg[o] = mi;
  }
  if (threadIdx.x == 0) L.wscale[o] = scale;
  if (L.wsq) {
    for (int i = threadIdx.x; i < L.I; i += 256) {
      float s = 0.f;
      for (int k = 0; k < RR; ++k) { const float v = w[i * RR + k] * scale; s += v * v; }
      L.wsq[(size_t)o * L.I + i] = s;
    }
  }
}

// tile (32 o x 32 i) per block: w [O][I][R][R] * scale[o] -> w_fwd [O][R][R][I] and w_adj [I][R][R][O] (taps reversed), storage T
template <typename T>
__device__ __forceinline__ void sg2_wprep_tile(const icg_sg2_weight& L, int to, int ti, float* tile /*[32][32*RR + 1]*/) {
  const int RR = L.R * L.R, rowlen = 32 * RR, ld = rowlen + 1;
  const int o0 = to * 32, i0 = ti * 32;
  for (int e = threadIdx.x; e < 32 * rowlen; e += 256) {
    const int ol = e / rowlen, f = e - ol * rowlen;            // f = il * RR + k
    const int o = o0 + ol, i = i0 + f / RR;
    float v = 0.f;
    if (o < L.O && i < L.I) v = L.w[((size_t)o * L.I + i0) * RR + f] * L.wscale[o];
    tile[ol * ld + f] = v;
  }
  __syncthreads();
  T* wf = (T*)L.w_fwd;
  T* wa = (T*)L.w_adj;
  // w_fwd[o][tap'][i]: i fastest
  for (int e = threadIdx.x; e < 32 * RR * 32; e += 256) {
    const int il = e & 31, rest = e >> 5, k = rest % RR, ol = rest / RR;
    const int o = o0 + ol, i = i0 + il;
    if (o < L.O && i < L.I) {
      const int kf = L.flip ? RR - 1 - k : k;
      Sg<T>::st1(wf + ((size_t)o * RR + kf) * L.I + i, tile[ol * ld + il * RR + k]);
    }
  }
  if (wa) {
    // w_adj[i][RR - 1 - tap'][o]: o fastest
    for (int e = threadIdx.x; e < 32 * RR * 32; e += 256) {
      const int ol = e & 31, rest = e >> 5, k = rest % RR, il = rest / RR;
      const int o = o0 + ol, i = i0 + il;
      if (o < L.O && i < L.I) {
        const int kf = L.flip ? RR - 1 - k : k;
        Sg<T>::st1(wa + ((size_t)i * RR + (RR - 1 - kf)) * L.O + o, tile[ol * ld + il * RR + k]);
      }
    }
  }
}
__global__ __launch_bounds__(256) void sg2_wprep_layout_kernel(SgwPack p) {
  extern __shared__ float sg2_tile[];
  int li = 0;
  while (li + 1 < p.n && (int)blockIdx.x >= p.blk0[li + 1]) ++li;
  const icg_sg2_weight L = p.t[li];
  const int b = blockIdx.x - p.blk0[li];
  const int tiles_i = (L.I + 31) / 32;
  if (L.dtype == 1) sg2_wprep_tile<__half>(L, b / tiles_i, b % tiles_i, sg2_tile);
  else sg2_wprep_tile<float>(L, b / tiles_i, b % tiles_i, sg2_tile);
}

// ------------------------------------------------------------------------------------------------------------------------------
// styles: s0 = (lin + bias * bias_gain) * post_gain;  s = s0 / max|s0| (fp16 pre-normalisation);  d[n][o] = rsqrt(sum_i s^2 wsq + 1e-8)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg2_style_prep_kernel(const float* __restrict__ lin, const float* __restrict__ bias, float bias_gain,
                                                             float post_gain, const float* __restrict__ wsq, int I, int O, int prenorm,
                                                             float* __restrict__ s_out, float* __restrict__ smax, int* __restrict__ sarg,
                                                             float* __restrict__ d_out) {
  extern __shared__ float sh_s[];             // [I]
  __shared__ float sh_m[4];
  __shared__ int sh_i[4];
  const int n = blockIdx.x;
  float m = -1.f;
  int mi = 0;
  for (int i = threadIdx.x; i < I; i += 256) {
    const float v = (lin[(size_t)n * I + i] + (bias ? bias[i] * bias_gain : 0.f)) * post_gain;
    sh_s[i] = v;
    const float a = fabsf(v);
    if (a > m) { m = a; mi = i; }
  }
  float sm = 1.f;
  if (prenorm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float m2 = __shfl_xor(m, off, 64);
      const int i2 = __shfl_xor(mi, off, 64);
      if (m2 > m || (m2 == m && i2 < mi)) { m = m2; mi = i2; }
    }
    if ((threadIdx.x & 63) == 0) { sh_m[threadIdx.x >> 6] = m; sh_i[threadIdx.x >> 6] = mi; }
    __syncthreads();
    m = sh_m[0]; mi = sh_i[0];
    for (int k = 1; k < 4; ++k)
      if (sh_m[k] > m || (sh_m[k] == m && sh_i[k] < mi)) { m = sh_m[k]; mi = sh_i[k]; }
    sm = m;
  }
  __syncthreads();
  const float s0_at_max = prenorm ? sh_s[mi] : 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < I; i += 256) {
    const float v = prenorm ? sh_s[i] / sm : sh_s[i];
    sh_s[i] = v;
    if (blockIdx.y == 0) s_out[(size_t)n * I + i] = v;
  }
  if (blockIdx.y == 0 && threadIdx.x == 0 && prenorm) { smax[n] = s0_at_max; sarg[n] = mi; }
  __syncthreads();
  if (!wsq) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < 4; ++k) {
    const int o = blockIdx.y * 16 + wv * 4 + k;
    if (o >= O) break;
    const float* row = wsq + (size_t)o * I;
    float q0 = 0.f, q1 = 0.f;
    int i = lane;
    for (; i + 64 < I; i += 128) {
      const float s0 = sh_s[i], s1 = sh_s[i + 64];
      q0 += s0 * s0 * row[i]; q1 += s1 * s1 * row[i + 64];
    }
    if (i < I) { const float s0 = sh_s[i]; q0 += s0 * s0 * row[i]; }
    const float q = wave_sum(q0 + q1);
    if (lane == 0) d_out[(size_t)n * O + o] = rsqrtf(q + 1e-8f);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// elementwise: modulate, demodulate + noise + bias + activation + clamp
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sg2_modulate_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ xs, unsigned nrows,
                                                           unsigned HW, int V) {
  constexpr int VEC = Sg<T>::VEC;
  const unsigned t0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned v = t0 % (unsigned)V, rstep = gridDim.x * 256u / (unsigned)V;
  for (unsigned row = t0 / (unsigned)V; row < nrows; row += rstep) {
    const unsigned n = row / HW;
    const size_t g = (size_t)row * V + v;
    float xv[VEC];
    Sg<T>::ld(x + g * VEC, xv);
    const float* sp = s + ((size_t)n * V + v) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) xv[j] = xv[j] * Sg<T>::rnd(sp[j]);
    Sg<T>::st(xs + g * VEC, xv);
  }
}

__device__ __forceinline__ float sg2_act(int act, float v, float alpha) { return (act == 3 && v < 0.f) ? v * alpha : v; }

// y = clamp(gain * act(c * d[n][o] + noise[n][p] * strength + bias[o]))   (act 1 = linear, 3 = lrelu)
// The grid stride is a multiple of V (a power of two <= 256), so a thread keeps its channel vector: bias in registers, 32-bit index math
template <typename T>
__global__ __launch_bounds__(256) void sg2_act_fwd_kernel(const T* __restrict__ c, const float* __restrict__ d, const float* __restrict__ noise,
                                                          long noise_bstride, const float* __restrict__ strength, const float* __restrict__ bias,
                                                          T* __restrict__ y, unsigned nrows, unsigned HW, int V, int act, float alpha, float gain,
                                                          float clamp) {
  constexpr int VEC = Sg<T>::VEC;
  const float st = (noise && strength) ? *strength : 1.f;
  const unsigned t0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned v = t0 % (unsigned)V, rstep = gridDim.x * 256u / (unsigned)V;
  float bv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) bv[j] = bias ? Sg<T>::rnd(bias[v * VEC + j]) : 0.f;
  for (unsigned row = t0 / (unsigned)V; row < nrows; row += rstep) {
    const unsigned n = row / HW, p = row - n * HW;
    const size_t g = (size_t)row * V + v;
    float cv[VEC];
    Sg<T>::ld(c + g * VEC, cv);
    const float nz = noise ? Sg<T>::rnd(noise[(size_t)n * noise_bstride + p] * st) : 0.f;
    if (d) {
      float dv[VEC];
      Sg<float>::ld(d + ((size_t)n * V + v) * VEC, dv);
      if (VEC == 8) Sg<float>::ld(d + ((size_t)n * V + v) * VEC + 4, dv + 4);
#pragma unroll
      for (int j = 0; j < VEC; ++j) cv[j] = Sg<T>::rnd(__fmaf_rn(cv[j], Sg<T>::rnd(dv[j]), nz));      // fma.fma = addcmul: one rounding
    } else if (noise) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) cv[j] = Sg<T>::rnd(cv[j] + nz);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float o = sg2_act(act, cv[j] + bv[j], alpha) * gain;
      if (clamp >= 0.f) o = fminf(fmaxf(o, -clamp), clamp);
      cv[j] = o;
    }
    Sg<T>::st(y + g * VEC, cv);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// row kernels with per-sample channel sums.  A block owns rows [r0, r1) of ONE sample; thread (v, rl) walks rows rl, rl + nrl, ...
// of channel vector v and keeps NACC fp32 sums per channel; the block's sums go to part[(n * chunks + chunk)][ctot] in a fixed order.
// MODE 0 (activation backward):  dz = dy * gain * act'(y) [|y| < clamp];  dc = dz * d[n][o];  sums: dz | dz * c | (column 2 O) dz * noise
// MODE 1 (modulation backward):  dx = dxs * s[n][c];                       sums: dxs * x
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void sg2_rows_kernel(const T* __restrict__ a /*dy | dxs*/, const T* __restrict__ b /*y | x*/,
                                                       const T* __restrict__ c /*conv output | -*/, const float* __restrict__ sv /*d | s*/,
                                                       const float* __restrict__ noise, long noise_bstride, T* __restrict__ out /*dc | dx*/,
                                                       float* __restrict__ part, int ctot, long HW, int V, int rpb, int chunks, int act,
                                                       float alpha, float gain, float clamp) {
  constexpr int VEC = Sg<T>::VEC;
  constexpr int NACC = MODE == 0 ? 2 : 1;
  __shared__ float red[256 * VEC];
  __shared__ float shn[4];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int v = threadIdx.x % V, rl = threadIdx.x / V, nrl = 256 / V;
  const long r0 = (long)chunk * rpb, r1 = min(r0 + rpb, HW);
  const int C = V * VEC;
  float acc[NACC][VEC], accn = 0.f, svv[VEC];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) svv[j] = sv ? Sg<T>::rnd(sv[((long)n * V + v) * VEC + j]) : 1.f;
  for (long r = r0 + rl; r < r1; r += nrl) {
    const long g = ((long)n * HW + r) * V + v;
    float av[VEC], bv[VEC], o[VEC];
    Sg<T>::ld(a + g * VEC, av);
    Sg<T>::ld(b + g * VEC, bv);
    if (MODE == 0) {
      float cv[VEC];
      if (c) Sg<T>::ld(c + g * VEC, cv);
      float rowsum = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float dz = av[j] * (gain * ((act == 3 && !(bv[j] > 0.f)) ? alpha : 1.f));
        // bias_act.cu: the slope is read off y / gain (sign of y); the clamp mask off y
        if (clamp >= 0.f && !(bv[j] > -clamp && bv[j] < clamp)) dz = 0.f;
        dz = Sg<T>::rnd(dz);
        acc[0][j] += dz;
        if (c) acc[1][j] += dz * cv[j];
        rowsum += dz;
        o[j] = sv ? Sg<T>::rnd(dz * svv[j]) : dz;
      }
      if (noise) accn += rowsum * noise[n * noise_bstride + r];
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc[0][j] += av[j] * bv[j];
        o[j] = Sg<T>::rnd(av[j] * svv[j]);
      }
    }
    if (out) Sg<T>::st(out + g * VEC, o);
  }
  float* prow = part + (size_t)blockIdx.x * ctot;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[k][j];
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {
      const int v2 = ch / VEC, j = ch % VEC;
      float s = 0.f;
      for (int q = 0; q < nrl; ++q) s += red[(q * V + v2) * VEC + j];
      prow[k * C + ch] = s;
    }
  }
  if (MODE == 0 && ctot > 2 * C) {
    const float t = block_sum_256(accn, shn);
    if (threadIdx.x == 0) prow[2 * C] = t;
  }
}

// u = x * a[n][c] + g * b[n][c]   (second-order passes: the cotangent of a modulated tensor; g / b may be null)
template <typename T>
__global__ __launch_bounds__(256) void sg2_mod2_kernel(const T* __restrict__ x, const float* __restrict__ a, const T* __restrict__ g,
                                                       const float* __restrict__ b, T* __restrict__ u, unsigned nrows, unsigned HW, int V) {
  constexpr int VEC = Sg<T>::VEC;
  const unsigned t0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned v = t0 % (unsigned)V, rstep = gridDim.x * 256u / (unsigned)V;
  for (unsigned row = t0 / (unsigned)V; row < nrows; row += rstep) {
    const unsigned n = row / HW;
    const size_t i = (size_t)row * V + v;
    float xv[VEC], o[VEC];
    Sg<T>::ld(x + i * VEC, xv);
    const float* ap = a + ((size_t)n * V + v) * VEC;
    // rounding points of the composed operators in fp16 storage: the per-sample factors and each product are fp16 tensors there
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = Sg<T>::rnd(xv[j] * Sg<T>::rnd(ap[j]));
    if (g) {
      float gv[VEC];
      Sg<T>::ld(g + i * VEC, gv);
      const float* bp = b + ((size_t)n * V + v) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = o[j] + Sg<T>::rnd(gv[j] * Sg<T>::rnd(bp[j]));
    }
    Sg<T>::st(u + i * VEC, o);
  }
}

// second-order pass through the activation / demodulation step of a layer's backward (dz = dy m, dc = dz d, dd = sum_p dz c):
//   cdy = (cdc * d + cdd[n][o] * c) * m        cc = cdd[n][o] * dz        sums[n][o] = sum_p cdc * dz
template <typename T>
__global__ __launch_bounds__(256) void sg2_act_bwd2_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ c,
                                                           const T* __restrict__ cdc, const float* __restrict__ d, const float* __restrict__ cdd,
                                                           T* __restrict__ cdy, T* __restrict__ cc, float* __restrict__ part, long HW, int V, int rpb,
                                                           int chunks, int act, float alpha, float gain, float clamp) {
  constexpr int VEC = Sg<T>::VEC;
  __shared__ float red[256 * VEC];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int v = threadIdx.x % V, rl = threadIdx.x / V, nrl = 256 / V;
  const long r0 = (long)chunk * rpb, r1 = min(r0 + rpb, HW);
  const int C = V * VEC;
  float acc[VEC], dv[VEC], ev[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    acc[j] = 0.f;
    dv[j] = d ? Sg<T>::rnd(d[((long)n * V + v) * VEC + j]) : 1.f;
    ev[j] = cdd ? cdd[((long)n * V + v) * VEC + j] : 0.f;
  }
  for (long r = r0 + rl; r < r1; r += nrl) {
    const long g = ((long)n * HW + r) * V + v;
    float gy[VEC], yv[VEC], cv[VEC], kv[VEC], o1[VEC], o2[VEC];
    Sg<T>::ld(dy + g * VEC, gy);
    Sg<T>::ld(y + g * VEC, yv);
    Sg<T>::ld(cdc + g * VEC, kv);
    if (c) Sg<T>::ld(c + g * VEC, cv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float m = gain * ((act == 3 && !(yv[j] > 0.f)) ? alpha : 1.f);
      if (clamp >= 0.f && !(yv[j] > -clamp && yv[j] < clamp)) m = 0.f;
      const float dz = Sg<T>::rnd(gy[j] * m);
      acc[j] += kv[j] * dz;
      o1[j] = (kv[j] * dv[j] + (c ? ev[j] * cv[j] : 0.f)) * m;
      o2[j] = ev[j] * dz;
    }
    Sg<T>::st(cdy + g * VEC, o1);
    if (cc) Sg<T>::st(cc + g * VEC, o2);
  }
  float* prow = part + (size_t)blockIdx.x * C;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[j];
  __syncthreads();
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    const int v2 = ch / VEC, j = ch % VEC;
    float s = 0.f;
    for (int q = 0; q < nrl; ++q) s += red[(q * V + v2) * VEC + j];
    prow[ch] = s;
  }
}

// part [N * chunks][ctot]  ->  per [N][ctot] (sum over the chunks of a sample) and tot [ctot] (sum over everything); fixed order.
// Block: 64 columns x 16 sample groups (one (sample, column) chain of <= 64 loads per thread at N = 16)
__global__ __launch_bounds__(1024) void sg2_rows_final_kernel(const float* __restrict__ part, int N, int chunks, int ctot, float* __restrict__ per,
                                                              float* __restrict__ tot) {
  __shared__ float red[1024];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cl;
  float t = 0.f;
  if (ch < ctot) {
    for (int n = grp; n < N; n += 16) {
      const float* p = part + (size_t)n * chunks * ctot + ch;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int k = 0;
      for (; k + 3 < chunks; k += 4) {
        s0 += p[(size_t)k * ctot]; s1 += p[(size_t)(k + 1) * ctot]; s2 += p[(size_t)(k + 2) * ctot]; s3 += p[(size_t)(k + 3) * ctot];
      }
      for (; k < chunks; ++k) s0 += p[(size_t)k * ctot];
      const float s = (s0 + s1) + (s2 + s3);
      if (per) per[(size_t)n * ctot + ch] = s;
      t += s;
    }
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (grp == 0 && ch < ctot && tot) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) a += red[g * 64 + cl];
    tot[ch] = a;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// style backward: t[n][o] = -dd d^3;  g[n][i] = ds_mod[n][i] + s[n][i] sum_o t[n][o] wsq[o][i];  pdot[n][blk] = sum_{i in blk} g s
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg2_style_bwd_kernel(const float* __restrict__ ds_mod, long ds_stride, const float* __restrict__ dd,
                                                            long dd_stride, const float* __restrict__ d, const float* __restrict__ s,
                                                            const float* __restrict__ wsq, int I, int O, float* __restrict__ g_out,
                                                            float* __restrict__ pdot, float* __restrict__ t_out) {
  extern __shared__ float sh_t[];             // [O] + [256]
  float* red = sh_t + O;
  const int n = blockIdx.x, by = blockIdx.y;
  const int il = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = by * 64 + il;
  if (dd) {
    for (int o = threadIdx.x; o < O; o += 256) {
      const float dv = d[(size_t)n * O + o];
      const float t = -dd[(size_t)n * dd_stride + o] * dv * dv * dv;
      sh_t[o] = t;
      if (by == 0) t_out[(size_t)n * O + o] = t;
    }
  }
  __syncthreads();
  float acc = 0.f;
  if (dd && i < I) {
    int o = grp;
    float a0 = 0.f, a1 = 0.f;
    for (; o + 4 < O; o += 8) { a0 += sh_t[o] * wsq[(size_t)o * I + i]; a1 += sh_t[o + 4] * wsq[(size_t)(o + 4) * I + i]; }
    for (; o < O; o += 4) a0 += sh_t[o] * wsq[(size_t)o * I + i];
    acc = a0 + a1;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  float gs = 0.f;
  if (grp == 0 && i < I) {
    const float sv = s[(size_t)n * I + i];
    const float g = ds_mod[(size_t)n * ds_stride + i] + sv * (red[il] + red[64 + il] + red[128 + il] + red[192 + il]);
    g_out[(size_t)n * I + i] = g;
    gs = g * sv;
  }
  if (grp == 0) {
    gs = wave_sum(gs);
    if (il == 0) pdot[(size_t)n * gridDim.y + by] = gs;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// small-batch fully-connected backward: dlin [N][I] (optionally through the style pre-normalisation) -> dW [I][K], db [I], dx [N][K]
// ------------------------------------------------------------------------------------------------------------------------------
struct SgNorm {                 // pre-normalisation s = s0 / |s0[arg]|: dlin = post_gain (g - [i == arg] sign(s0[arg]) P) / |s0[arg]|
  const float* smax;
  const int* sarg;
  const float* pdot;
  int npdot;
  float post_gain;
};
__device__ __forceinline__ float sg2_dlin(const float* __restrict__ g, const SgNorm& nm, int n, int i, int I) {
  float v = g[(size_t)n * I + i];
  if (nm.smax) {
    const float sm = nm.smax[n];
    if (i == nm.sarg[n]) {
      float P = 0.f;
      for (int k = 0; k < nm.npdot; ++k) P += nm.pdot[(size_t)n * nm.npdot + k];
      v -= (sm < 0.f ? -P : P);
    }
    v /= fabsf(sm);
  }
  return v * nm.post_gain;
}

constexpr int FCB_NMAX = 64;
// block: 8 rows of dW (i0 .. i0 + 7), all k
__global__ __launch_bounds__(256) void sg2_fc_bwd_dw_kernel(const float* __restrict__ g, SgNorm nm, const float* __restrict__ x, int N, int I, int K,
                                                            float wgain, float bias_gain, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float dl[FCB_NMAX][8];
  const int i0 = blockIdx.x * 8;
  for (int e = threadIdx.x; e < N * 8; e += 256) {
    const int n = e >> 3, j = e & 7;
    dl[n][j] = (i0 + j < I) ? sg2_dlin(g, nm, n, i0 + j, I) : 0.f;
  }
  __syncthreads();
  if (db && threadIdx.x < 8 && i0 + threadIdx.x < I) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dl[n][threadIdx.x];
    db[i0 + threadIdx.x] = s * bias_gain;
  }
  if (!dW) return;
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int n = 0; n < N; ++n) {
      const float xv = x[(size_t)n * K + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += dl[n][j] * xv;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j < I) dW[(size_t)(i0 + j) * K + k] = acc[j] * wgain;
  }
}
// block: 32 columns k of dx, rows n0 .. n0 + 15 (blockIdx.y); 8 groups split the i range
__global__ __launch_bounds__(256) void sg2_fc_bwd_dx_kernel(const float* __restrict__ g, SgNorm nm, const float* __restrict__ W, int N, int I, int K,
                                                            float wgain, float* __restrict__ dx) {
  __shared__ float dl[16][257];
  __shared__ float red[8][16][32];
  const int kl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + kl, n0 = blockIdx.y * 16;
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int i0 = 0; i0 < I; i0 += 256) {
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 256; e += 256) {
      const int q = e >> 8, il = e & 255;
      dl[q][il] = (n0 + q < N && i0 + il < I) ? sg2_dlin(g, nm, n0 + q, i0 + il, I) : 0.f;
    }
    __syncthreads();
    if (k < K) {
      const int iend = min(256, I - i0);
      int il = grp;
      for (; il + 8 < iend; il += 16) {
        const float w0 = W[(size_t)(i0 + il) * K + k], w1 = W[(size_t)(i0 + il + 8) * K + k];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += dl[q][il] * w0 + dl[q][il + 8] * w1;
      }
      for (; il < iend; il += 8) {
        const float wv = W[(size_t)(i0 + il) * K + k];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += dl[q][il] * wv;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[grp][q][kl] = acc[q];
  __syncthreads();
  if (k < K) {
    for (int q = grp; q < 16; q += 8)
      if (n0 + q < N) {
        float a = 0.f;
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) a += red[gq][q][kl];
        dx[(size_t)(n0 + q) * K + k] = a * wgain;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// weight gradient assembly: convolution weight gradient (HWIO or HWOI, fp32) + demodulation term, through the pre-normalisation / gain
// ------------------------------------------------------------------------------------------------------------------------------
struct SgWb {
  const float* dwc;       // layout 0: [RR][I][O]; 1: [RR][O][I]
  const float* Q;         // [O][I] or null: g += w scale Q[o][i]  (second-order passes: a general cotangent of the demodulation table)
  const float* t;         // [N][O] or null
  const float* s;         // [N][I]
  const float* w;         // [O][I][RR]
  const float* wscale;    // [O]
  float* dw;              // [O][I][RR]
  float* rowdot;          // [O][tiles_i] or null
  int layout, N, O, I, RR, round_f16, tiles_i;
};
__global__ __launch_bounds__(256) void sg2_weight_bwd_kernel(SgWb p) {
  extern __shared__ float sg2_wb[];           // g [32][32 RR + 1] | tt [N][32] | ss [N][32]
  const int RR = p.RR, rowlen = 32 * RR, ld = rowlen + 1;
  float* gt = sg2_wb;
  float* tt = gt + 32 * ld;
  float* ss = tt + p.N * 32;
  const int to = blockIdx.x / p.tiles_i, ti = blockIdx.x - to * p.tiles_i;
  const int o0 = to * 32, i0 = ti * 32;
  if (p.layout == 0) {
    for (int e = threadIdx.x; e < RR * 32 * 32; e += 256) {
      const int ol = e & 31, rest = e >> 5, il = rest & 31, k = rest >> 5;
      const int o = o0 + ol, i = i0 + il;
      float v = (o < p.O && i < p.I) ? p.dwc[((size_t)k * p.I + i) * p.O + o] : 0.f;
      if (p.round_f16) v = Sg<__half>::rnd(v);
      gt[ol * ld + il * RR + k] = v;
    }
  } else {
    for (int e = threadIdx.x; e < RR * 32 * 32; e += 256) {
      const int il = e & 31, rest = e >> 5, ol = rest & 31, k = rest >> 5;
      const int o = o0 + ol, i = i0 + il;
      float v = (o < p.O && i < p.I) ? p.dwc[((size_t)k * p.O + o) * p.I + i] : 0.f;
      if (p.round_f16) v = Sg<__half>::rnd(v);
      gt[ol * ld + il * RR + k] = v;
    }
  }
  if (p.t) {
    for (int e = threadIdx.x; e < p.N * 32; e += 256) {
      const int n = e >> 5, l = e & 31;
      tt[e] = (o0 + l < p.O) ? p.t[(size_t)n * p.O + o0 + l] : 0.f;
      const float sv = (i0 + l < p.I) ? p.s[(size_t)n * p.I + i0 + l] : 0.f;
      ss[e] = sv * sv;
    }
  }
  __syncthreads();
  // 8 threads per output channel walk its 32 RR contiguous elements
  const int ol = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const int o = o0 + ol;
  float dot = 0.f;
  if (o < p.O) {
    const float sc = p.wscale[o];
    for (int f = sub; f < rowlen; f += 8) {
      const int il = f / RR, i = i0 + il;
      if (i >= p.I) break;
      const size_t idx = ((size_t)o * p.I + i0) * RR + f;
      const float wv = p.w[idx];
      float g = gt[ol * ld + f];
      if (p.t) {
        float q = 0.f;
        for (int n = 0; n < p.N; ++n) q += tt[n * 32 + ol] * ss[n * 32 + il];
        g += wv * sc * q;
      }
      if (p.Q) g += wv * sc * p.Q[(size_t)o * p.I + i];
      dot += g * wv;
      p.dw[idx] = g * sc;
    }
  }
  if (p.rowdot) {
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (sub == 0 && o < p.O) p.rowdot[(size_t)o * p.tiles_i + ti] = dot;
  }
}
// scale[o] = c0 / m: d scale / d w[arg] = -sign(w[arg]) scale / m = -sign scale^2 / c0
__global__ __launch_bounds__(256) void sg2_weight_bwd_fix_kernel(const float* __restrict__ rowdot, int tiles_i, const float* __restrict__ w,
                                                                 const float* __restrict__ wscale, const int* __restrict__ warg, float c0, int O,
                                                                 int len, float* __restrict__ dw) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O) return;
  float D = 0.f;
  for (int k = 0; k < tiles_i; ++k) D += rowdot[(size_t)o * tiles_i + k];
  const size_t idx = (size_t)o * len + warg[o];
  const float sc = wscale[o];
  dw[idx] -= (w[idx] < 0.f ? -1.f : 1.f) * D * sc * sc / c0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// toRGB: a [pixels x C] x [C x 3] contraction with per-sample weights, HBM-bound on x
// ------------------------------------------------------------------------------------------------------------------------------
// L lanes per pixel (L = min(64, V)), each lane owns NV = V / L channel vectors
template <typename T, int NV>
__global__ __launch_bounds__(256) void sg2_torgb_fwd_kernel(const T* __restrict__ x, const float* __restrict__ s, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float clamp, const float* __restrict__ img_in,
                                                            float* __restrict__ img_out, T* __restrict__ y, long HW, int C, int L, int rpb, int chunks) {
  constexpr int VEC = Sg<T>::VEC;
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int l = threadIdx.x % L, rl = threadIdx.x / L, nrl = 256 / L;
  const long r0 = (long)chunk * rpb, r1 = min(r0 + rpb, HW);
  float sv[NV][VEC], wv[3][NV][VEC];
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int ch = (q * L + l) * VEC + j;
      sv[q][j] = Sg<T>::rnd(s[(size_t)n * C + ch]);
#pragma unroll
      for (int o = 0; o < 3; ++o) wv[o][q][j] = Sg<T>::rnd(w[o * C + ch]);
    }
  const float b0 = bias ? Sg<T>::rnd(bias[0]) : 0.f, b1 = bias ? Sg<T>::rnd(bias[1]) : 0.f, b2 = bias ? Sg<T>::rnd(bias[2]) : 0.f;
  for (long rb = r0; rb < r1; rb += nrl) {              // all lanes of a wave stay in the loop (shuffles below)
    const long r = rb + rl;
    const bool ok = r < r1;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (ok) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float xv[VEC];
        Sg<T>::ld(x + (((long)n * HW + r) * (NV * L) + q * L + l) * VEC, xv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xs = Sg<T>::rnd(xv[j] * sv[q][j]);
          a0 += xs * wv[0][q][j]; a1 += xs * wv[1][q][j]; a2 += xs * wv[2][q][j];
        }
      }
    }
    for (int off = L >> 1; off > 0; off >>= 1) {
      a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64);
    }
    if (ok && l == 0) {
      float o[3] = {Sg<T>::rnd(a0) + b0, Sg<T>::rnd(a1) + b1, Sg<T>::rnd(a2) + b2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (clamp >= 0.f) o[k] = fminf(fmaxf(o[k], -clamp), clamp);
        o[k] = Sg<T>::rnd(o[k]);
        Sg<T>::st1(y + ((long)n * HW + r) * 3 + k, o[k]);
        const size_t ip = ((size_t)n * 3 + k) * HW + r;          // the image is NCHW fp32 (networks.py:630: accumulated in fp32)
        img_out[ip] = img_in ? img_in[ip] + o[k] : o[k];
      }
    }
  }
}

// dz = dimg [|y| < clamp];  dxs = dz . w;  dx = dxs * s;  sums: ds[n][c] | dw[0..2][c] | db[0..2]
template <typename T, int NV>
__global__ __launch_bounds__(256) void sg2_torgb_bwd_kernel(const float* __restrict__ dimg, const T* __restrict__ y, const T* __restrict__ x,
                                                            const float* __restrict__ s, const float* __restrict__ w, float clamp, int mask_clamp,
                                                            T* __restrict__ dx, float* __restrict__ part, long HW, int C, int L, int rpb, int chunks) {
  constexpr int VEC = Sg<T>::VEC;
  __shared__ float red[256 * VEC];
  __shared__ float shb[4];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int l = threadIdx.x % L, rl = threadIdx.x / L, nrl = 256 / L;
  const long r0 = (long)chunk * rpb, r1 = min(r0 + rpb, HW);
  const int ctot = 4 * C + 3;
  float sv[NV][VEC], wv[3][NV][VEC], acc[4][NV][VEC], accb[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int ch = (q * L + l) * VEC + j;
      sv[q][j] = Sg<T>::rnd(s[(size_t)n * C + ch]);
#pragma unroll
      for (int o = 0; o < 3; ++o) wv[o][q][j] = Sg<T>::rnd(w[o * C + ch]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k][q][j] = 0.f;
    }
  for (long r = r0 + rl; r < r1; r += nrl) {
    float dz[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float g = Sg<T>::rnd(dimg[((size_t)n * 3 + k) * HW + r]);
      if (mask_clamp && clamp >= 0.f) {
        const float yv = Sg<T>::ld1(y + ((long)n * HW + r) * 3 + k);
        if (!(yv > -clamp && yv < clamp)) g = 0.f;
      }
      dz[k] = g;
      if (l == 0) accb[k] += g;
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const long g = (((long)n * HW + r) * (NV * L) + q * L + l) * VEC;
      float xv[VEC], o[VEC];
      Sg<T>::ld(x + g, xv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float dxs = Sg<T>::rnd(dz[0] * wv[0][q][j] + dz[1] * wv[1][q][j] + dz[2] * wv[2][q][j]);
        const float xs = Sg<T>::rnd(xv[j] * sv[q][j]);
        acc[0][q][j] += dxs * xv[j];
        acc[1][q][j] += dz[0] * xs; acc[2][q][j] += dz[1] * xs; acc[3][q][j] += dz[2] * xs;
        o[j] = Sg<T>::rnd(dxs * sv[q][j]);
      }
      if (dx) Sg<T>::st(dx + g, o);
    }
  }
  float* prow = part + (size_t)blockIdx.x * ctot;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[k][q][j];
      __syncthreads();
      for (int e = threadIdx.x; e < L * VEC; e += 256) {
        const int l2 = e / VEC, j = e % VEC;
        float sum = 0.f;
        for (int t = 0; t < nrl; ++t) sum += red[(t * L + l2) * VEC + j];
        prow[k * C + (q * L + l2) * VEC + j] = sum;
      }
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t = block_sum_256(accb[k], shb);
    if (threadIdx.x == 0) prow[4 * C + k] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// FIR pass (up = down = 1: the blur after a transposed convolution, conv2d_resample.py:163-186) with the layer epilogue on the filtered
// values: c = fir(x) (stored for the demodulation gradient), y = clamp(gain * act(c * d + noise + bias)).  A thread owns one 16-byte
// channel vector of a vertical strip of TY outputs, as the stand-alone upfirdn2d kernel does (csrc/stylegan_ops_typed.hip).
// ------------------------------------------------------------------------------------------------------------------------------
// TX output columns x TY output rows per thread: the (TY + fh - 1) x (TX + fw - 1) window is loaded once for TX TY outputs (TX = 2, TY = 4
// with the 4 x 4 filter of the networks: 35 loads per 8 outputs; the stand-alone upfirdn2d kernel does 28 per 4 and is bound by L1 traffic).
// FS > 0: filter size known at compile time (fully unrolled tap selection); FS = 0: any fh x fw <= 64 taps.
template <typename T, int TY, int TX, int FS>
__global__ __launch_bounds__(256) void sg2_fir_act_kernel(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ cout_,
                                                          T* __restrict__ y, const float* __restrict__ d, const float* __restrict__ noise,
                                                          long noise_bstride, const float* __restrict__ strength, const float* __restrict__ bias,
                                                          int N, int H, int W, int CV, int fh_, int fw_, int padx0, int pady0, int flip, float fgain,
                                                          int outH, int outW, int act, float alpha, float gain, float clamp) {
  constexpr int VEC = Sg<T>::VEC;
  const int fh = FS ? FS : fh_, fw = FS ? FS : fw_;
  __shared__ float fs[64];
  for (int i = threadIdx.x; i < fh * fw; i += blockDim.x) {
    const int ty = i / fw, tx = i - ty * fw;
    fs[i] = f[(flip ? ty : fh - 1 - ty) * fw + (flip ? tx : fw - 1 - tx)] * fgain;      // upfirdn2d: correlation with the flipped filter unless flip_filter
  }
  __syncthreads();
  const float st = (noise && strength) ? *strength : 1.f;
  const unsigned strips = (unsigned)((outH + TY - 1) / TY), cols = (unsigned)((outW + TX - 1) / TX);
  const unsigned total = (unsigned)N * strips * cols * (unsigned)CV;
  const unsigned t0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned cv = t0 % (unsigned)CV;                            // (grid stride is a multiple of CV: the thread keeps its channel vector)
  float bv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = bias ? Sg<T>::rnd(bias[cv * VEC + e]) : 0.f;
  for (unsigned i = t0; i < total; i += gridDim.x * 256u) {
    unsigned t = i / (unsigned)CV;
    const int ox0 = (int)(t % cols) * TX;
    t /= cols;
    const int ys = (int)(t % strips), n = (int)(t / strips);
    const int oy0 = ys * TY;
    const int bx = ox0 - padx0, by0 = oy0 - pady0;
    float acc[TY][TX][VEC];
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
      for (int q = 0; q < TX; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[j][q][e] = 0.f;
    const T* xp = x + ((size_t)n * H * W * CV + cv) * VEC;
    if (FS) {
#pragma unroll
      for (int r = 0; r < TY + FS - 1; ++r) {
        const int iy = by0 + r;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int cc = 0; cc < TX + FS - 1; ++cc) {
          const int ix = bx + cc;
          if (ix < 0 || ix >= W) continue;
          float v[VEC];
          Sg<T>::ld(xp + ((size_t)iy * W + ix) * CV * VEC, v);
#pragma unroll
          for (int j = 0; j < TY; ++j) {
            if (r - j < 0 || r - j >= FS) continue;                 // (compile-time after unrolling)
#pragma unroll
            for (int q = 0; q < TX; ++q) {
              if (cc - q < 0 || cc - q >= FS) continue;
              const float w = fs[(r - j) * FS + (cc - q)];
#pragma unroll
              for (int e = 0; e < VEC; ++e) acc[j][q][e] += v[e] * w;
            }
          }
        }
      }
    } else {
      const int ix0 = max(bx, 0), ix1 = min(bx + TX - 1 + fw, W);
      const int iy0 = max(by0, 0), iy1 = min(by0 + TY - 1 + fh, H);
      for (int iy = iy0; iy < iy1; ++iy) {
        for (int ix = ix0; ix < ix1; ++ix) {
          float v[VEC];
          Sg<T>::ld(xp + ((size_t)iy * W + ix) * CV * VEC, v);
#pragma unroll
          for (int j = 0; j < TY; ++j) {
            const int ty = iy - by0 - j;
            if (ty < 0 || ty >= fh) continue;
#pragma unroll
            for (int q = 0; q < TX; ++q) {
              const int tx = ix - bx - q;
              if (tx < 0 || tx >= fw) continue;
              const float w = fs[ty * fw + tx];
#pragma unroll
              for (int e = 0; e < VEC; ++e) acc[j][q][e] += v[e] * w;
            }
          }
        }
      }
    }
    float dv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) dv[e] = d ? Sg<T>::rnd(d[((size_t)n * CV + cv) * VEC + e]) : 1.f;
#pragma unroll
    for (int j = 0; j < TY; ++j) {
      if (oy0 + j >= outH) break;
#pragma unroll
      for (int q = 0; q < TX; ++q) {
        if (ox0 + q >= outW) break;
        const size_t o = ((((size_t)n * outH + oy0 + j) * outW + ox0 + q) * CV + cv) * VEC;
        const float nz = noise ? Sg<T>::rnd(noise[n * noise_bstride + (long)(oy0 + j) * outW + ox0 + q] * st) : 0.f;
        float cv_[VEC], yv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          cv_[e] = Sg<T>::rnd(acc[j][q][e]);
          float z = cv_[e];
          if (d) z = Sg<T>::rnd(__fmaf_rn(z, dv[e], nz));
          else if (noise) z = Sg<T>::rnd(z + nz);
          float o2 = sg2_act(act, z + bv[e], alpha) * gain;
          if (clamp >= 0.f) o2 = fminf(fmaxf(o2, -clamp), clamp);
          yv[e] = o2;
        }
        if (cout_) Sg<T>::st(cout_ + o, cv_);
        Sg<T>::st(y + o, yv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// fromRGB (networks.py:831-836, Conv2dLayer 1x1 over the 3-channel image + bias + lrelu + clamp): y[n][p][o] written once from the
// planar image -- the generic path runs an fp32 [pixels x 3] GEMM, a cast and the activation pass (three tensors of the size of y)
// ------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sg2_fromrgb_fwd_kernel(const T* __restrict__ x /*[N][3][HW]*/, const T* __restrict__ w /*[O][3]*/,
                                                              const float* __restrict__ bias, T* __restrict__ y, unsigned nrows, unsigned HW, int V,
                                                              int act, float alpha, float gain, float clamp) {
  constexpr int VEC = Sg<T>::VEC;
  const unsigned t0 = blockIdx.x * 256u + threadIdx.x;
  const unsigned v = t0 % (unsigned)V, rstep = gridDim.x * 256u / (unsigned)V;      // (the grid stride is a multiple of V: the thread keeps its channels)
  float wv[VEC][3], bv[VEC];
  {
    float wf[3 * VEC];                                  // the thread's 3 VEC weights are contiguous: three 16-byte loads
#pragma unroll
    for (int k = 0; k < 3; ++k) Sg<T>::ld(w + (size_t)v * VEC * 3 + k * VEC, wf + k * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      wv[j][0] = wf[3 * j]; wv[j][1] = wf[3 * j + 1]; wv[j][2] = wf[3 * j + 2];
      bv[j] = bias ? Sg<T>::rnd(bias[v * VEC + j]) : 0.f;
    }
  }
  for (unsigned row = t0 / (unsigned)V; row < nrows; row += rstep) {
    const unsigned n = row / HW, p = row - n * HW;
    const T* xp = x + (size_t)n * 3 * HW + p;
    const float x0 = Sg<T>::ld1(xp), x1 = Sg<T>::ld1(xp + HW), x2 = Sg<T>::ld1(xp + 2 * (size_t)HW);
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = Sg<T>::rnd(x0 * wv[j][0] + x1 * wv[j][1] + x2 * wv[j][2]);
      a = sg2_act(act, a + bv[j], alpha) * gain;
      if (clamp >= 0.f) a = fminf(fmaxf(a, -clamp), clamp);
      o[j] = a;
    }
    Sg<T>::st(y + ((size_t)row * V + v) * VEC, o);
  }
}

// dz = dy * gain * act'(y) [|y| < clamp];  sums per channel o: dz x0 | dz x1 | dz x2 | dz  (columns 4 o + k);  dimg[n][c][p] = sum_o dz w[o][c]
template <typename T>
__global__ __launch_bounds__(256) void sg2_fromrgb_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                              const T* __restrict__ w, T* __restrict__ dimg, float* __restrict__ part, long HW, int V,
                                                              int rpb, int chunks, int act, float alpha, float gain, float clamp) {
  constexpr int VEC = Sg<T>::VEC;
  __shared__ float red[256 * VEC];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int v = threadIdx.x % V, rl = threadIdx.x / V, nrl = 256 / V;
  const long r0 = (long)chunk * rpb, r1 = min(r0 + rpb, HW);
  const int C = V * VEC, ctot = 4 * C;
  float acc[4][VEC], wv[3][VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k][j] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) wv[k][j] = dimg ? Sg<T>::ld1(w + (size_t)(v * VEC + j) * 3 + k) : 0.f;
  }
  for (long rb = r0; rb < r1; rb += nrl) {                    // (all lanes stay in the loop: shuffles below)
    const long r = rb + rl;
    const bool ok = r < r1;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (ok) {
      const long g = ((long)n * HW + r) * V + v;
      float gv[VEC], yv[VEC];
      Sg<T>::ld(dy + g * VEC, gv);
      Sg<T>::ld(y + g * VEC, yv);
      const float x0 = Sg<T>::ld1(x + ((long)n * 3 + 0) * HW + r), x1 = Sg<T>::ld1(x + ((long)n * 3 + 1) * HW + r),
                  x2 = Sg<T>::ld1(x + ((long)n * 3 + 2) * HW + r);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float dz = gv[j] * (gain * ((act == 3 && !(yv[j] > 0.f)) ? alpha : 1.f));
        if (clamp >= 0.f && !(yv[j] > -clamp && yv[j] < clamp)) dz = 0.f;
        dz = Sg<T>::rnd(dz);
        acc[0][j] += dz * x0; acc[1][j] += dz * x1; acc[2][j] += dz * x2; acc[3][j] += dz;
        d0 += dz * wv[0][j]; d1 += dz * wv[1][j]; d2 += dz * wv[2][j];
      }
    }
    if (dimg) {
      for (int off = V >> 1; off > 0; off >>= 1) {
        d0 += __shfl_xor(d0, off, 64); d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64);
      }
      if (ok && v == 0) {
        Sg<T>::st1(dimg + ((long)n * 3 + 0) * HW + r, d0);
        Sg<T>::st1(dimg + ((long)n * 3 + 1) * HW + r, d1);
        Sg<T>::st1(dimg + ((long)n * 3 + 2) * HW + r, d2);
      }
    }
  }
  float* prow = part + (size_t)blockIdx.x * ctot;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[k][j];
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {
      const int v2 = ch / VEC, j = ch % VEC;
      float s = 0.f;
      for (int q = 0; q < nrl; ++q) s += red[(q * V + v2) * VEC + j];
      prow[4 * ch + k] = s;
    }
  }
}

// second-order pass through ToRGB's backward (dz = dimg m, dxs = dz . w, dx = dxs s, ds = sum_p dxs x, dw = sum dz xs): with
// u = x a[n][c] + cdx s (the cotangent of dxs):  cdimg[n][o][p] = m sum_c u_c w[o][c] (+ cim),  cx = dxs a,
// sums[n]: cot s[c] = sum_p cdx dxs | cot w[o][c] = sum_p dz_o u_c   (columns C | 3 C)
template <typename T, int NV>
__global__ __launch_bounds__(256) void sg2_torgb_bwd2_kernel(const float* __restrict__ dimg, const T* __restrict__ y, const T* __restrict__ x,
                                                             const float* __restrict__ s, const float* __restrict__ w, const float* __restrict__ a,
                                                             const T* __restrict__ cdx, const float* __restrict__ cim, float clamp, int mask_clamp,
                                                             float* __restrict__ cdimg, T* __restrict__ cx, float* __restrict__ part, long HW, int C,
                                                             int L, int rpb, int chunks) {
  constexpr int VEC = Sg<T>::VEC;
  __shared__ float red[256 * VEC];
  const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const int l = threadIdx.x % L, rl = threadIdx.x / L, nrl = 256 / L;
  const long r0 = (long)chunk * rpb, r1 = min(r0 + rpb, HW);
  const int ctot = 4 * C;
  float sv[NV][VEC], av[NV][VEC], wv[3][NV][VEC], acc[4][NV][VEC];
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int ch = (q * L + l) * VEC + j;
      sv[q][j] = Sg<T>::rnd(s[(size_t)n * C + ch]);
      av[q][j] = Sg<T>::rnd(a[(size_t)n * C + ch]);
#pragma unroll
      for (int o = 0; o < 3; ++o) wv[o][q][j] = Sg<T>::rnd(w[o * C + ch]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k][q][j] = 0.f;
    }
  for (long rb = r0; rb < r1; rb += nrl) {                  // (all lanes stay in the loop: shuffles below)
    const long r = rb + rl;
    const bool ok = r < r1;
    float dz[3] = {0.f, 0.f, 0.f}, mk[3] = {0.f, 0.f, 0.f}, t0 = 0.f, t1 = 0.f, t2 = 0.f;
    if (ok) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        mk[k] = 1.f;
        if (mask_clamp && clamp >= 0.f) {
          const float yv = Sg<T>::ld1(y + ((long)n * HW + r) * 3 + k);
          if (!(yv > -clamp && yv < clamp)) mk[k] = 0.f;
        }
        dz[k] = Sg<T>::rnd(dimg[((size_t)n * 3 + k) * HW + r]) * mk[k];
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const long g = (((long)n * HW + r) * (NV * L) + q * L + l) * VEC;
        float xv[VEC], gv[VEC], o[VEC];
        Sg<T>::ld(x + g, xv);
        if (cdx) Sg<T>::ld(cdx + g, gv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float dxs = Sg<T>::rnd(dz[0] * wv[0][q][j] + dz[1] * wv[1][q][j] + dz[2] * wv[2][q][j]);
          const float u = Sg<T>::rnd(Sg<T>::rnd(xv[j] * av[q][j]) + (cdx ? Sg<T>::rnd(gv[j] * sv[q][j]) : 0.f));      // (fp16 tensors in the composed graph)
          if (cdx) acc[0][q][j] += gv[j] * dxs;
          acc[1][q][j] += dz[0] * u; acc[2][q][j] += dz[1] * u; acc[3][q][j] += dz[2] * u;
          t0 += u * wv[0][q][j]; t1 += u * wv[1][q][j]; t2 += u * wv[2][q][j];
          o[j] = dxs * av[q][j];
        }
        if (cx) Sg<T>::st(cx + g, o);
      }
    }
    for (int off = L >> 1; off > 0; off >>= 1) {
      t0 += __shfl_xor(t0, off, 64); t1 += __shfl_xor(t1, off, 64); t2 += __shfl_xor(t2, off, 64);
    }
    if (ok && l == 0) {
      const float tt[3] = {t0, t1, t2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const size_t ip = ((size_t)n * 3 + k) * HW + r;
        cdimg[ip] = tt[k] * mk[k] + (cim ? cim[ip] : 0.f);
      }
    }
  }
  float* prow = part + (size_t)blockIdx.x * ctot;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[threadIdx.x * VEC + j] = acc[k][q][j];
      __syncthreads();
      for (int e = threadIdx.x; e < L * VEC; e += 256) {
        const int l2 = e / VEC, j = e % VEC;
        float sum = 0.f;
        for (int t = 0; t < nrl; ++t) sum += red[(t * L + l2) * VEC + j];
        prow[k * C + (q * L + l2) * VEC + j] = sum;
      }
    }
}

int rows_geometry(long HW, int V, int* rpb, int* chunks) {
  const int nrl = 256 / V;
  long r = icg_cdiv(HW, 64);
  if (r < nrl) r = nrl;
  r = icg_cdiv(r, nrl) * nrl;
  *rpb = (int)r;
  *chunks = (int)icg_cdiv(HW, r);
  return 0;
}
bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; }
// kernels whose threads load per-thread constants (weights, bias) before the loop: a fixed grid of <= 16 workgroups per CU, many rows per thread
unsigned ew_grid_capped(long nvec) {
  long b = icg_cdiv(nvec, 256);
  if (b > 4096) b = 4096;
  return (unsigned)(b < 1 ? 1 : b);
}
unsigned ew_grid(long nvec) {
  long b = icg_cdiv(nvec, 256);
  if (b > ICG_GRID_CAP) b = ICG_GRID_CAP;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

// ============================================================================================================================= C-ABI
extern "C" int icg_sg2_weight_prep_multi(const icg_sg2_weight* layers, int n, void* stream) {
  ICG_REQUIRE(layers && n >= 0);
  hipStream_t st = (hipStream_t)stream;
  int done = 0;
  while (done < n) {
    SgwPack rows, tiles;
    rows.n = tiles.n = 0;
    int rb = 0, tb = 0, maxRR = 1;
    while (done < n && rows.n < SGW_MAX) {
      const icg_sg2_weight& L = layers[done++];
      ICG_REQUIRE(L.w && L.w_fwd && L.wscale && L.O > 0 && L.I > 0 && L.R >= 1 && L.R <= 3 && (L.dtype == 0 || L.dtype == 1));
      ICG_REQUIRE(!L.prenorm || L.warg);
      rows.t[rows.n] = L; rows.blk0[rows.n] = rb; rb += L.O; rows.n++;
      tiles.t[tiles.n] = L; tiles.blk0[tiles.n] = tb; tb += (int)(icg_cdiv(L.O, 32) * icg_cdiv(L.I, 32)); tiles.n++;
      if (L.R * L.R > maxRR) maxRR = L.R * L.R;
    }
    rows.blk0[rows.n] = rb; tiles.blk0[tiles.n] = tb;
    hipLaunchKernelGGL(sg2_wprep_rows_kernel, dim3(rb), dim3(256), 0, st, rows);
    const size_t lds = (size_t)32 * (32 * maxRR + 1) * sizeof(float);
    hipLaunchKernelGGL(sg2_wprep_layout_kernel, dim3(tb), dim3(256), lds, st, tiles);
    const int rc = icg_check_launch();
    if (rc != ICG_OK) return rc;
  }
  return ICG_OK;
}

extern "C" int icg_sg2_style_prep(const float* lin, const float* bias, float bias_gain, float post_gain, const float* wsq, int N, int I, int O,
                                  int prenorm, float* s, float* smax, int* sarg, float* d, void* stream) {
  ICG_REQUIRE(lin && s && N > 0 && I > 0 && I <= 8192 && (!wsq || (d && O > 0)) && (!prenorm || (smax && sarg)));
  const dim3 grid(N, wsq ? (unsigned)icg_cdiv(O, 16) : 1);
  hipLaunchKernelGGL(sg2_style_prep_kernel, grid, dim3(256), (size_t)I * sizeof(float), (hipStream_t)stream, lin, bias, bias_gain, post_gain, wsq, I,
                     O, prenorm, s, smax, sarg, d);
  return icg_check_launch();
}

extern "C" int icg_sg2_rows_applies(int C, int dtype) {
  const int vec = dtype == 1 ? 8 : 4;
  if ((dtype != 0 && dtype != 1) || C < vec || C % vec != 0) return 0;
  const int V = C / vec;
  return (V <= 256 && pow2(V)) ? 1 : 0;
}

extern "C" int icg_sg2_modulate(const void* x, const float* s, void* xs, int N, int64_t HW, int C, int dtype, void* stream) {
  ICG_REQUIRE(x && s && xs && N > 0 && HW > 0 && icg_sg2_rows_applies(C, dtype) && al16(x) && al16(xs));
  const int V = C / (dtype == 1 ? 8 : 4);
  const long nvec = (long)N * HW * V;
  hipStream_t st = (hipStream_t)stream;
  ICG_REQUIRE((long)N * HW < 0x7fffffffL);
  if (dtype == 1)
    hipLaunchKernelGGL(sg2_modulate_kernel<__half>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const __half*)x, s, (__half*)xs, (unsigned)((long)N * HW), (unsigned)HW, V);
  else
    hipLaunchKernelGGL(sg2_modulate_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)x, s, (float*)xs, (unsigned)((long)N * HW), (unsigned)HW, V);
  return icg_check_launch();
}

extern "C" int icg_sg2_act_fwd(const void* c, const float* d, const float* noise, int64_t noise_bstride, const float* strength, const float* bias,
                               void* y, int N, int64_t HW, int O, int act, float alpha, float gain, float clamp, int dtype, void* stream) {
  ICG_REQUIRE(c && y && N > 0 && HW > 0 && icg_sg2_rows_applies(O, dtype) && (act == 1 || act == 3) && al16(c) && al16(y) && al16(d));
  ICG_REQUIRE((long)N * HW < 0x7fffffffL);
  const int V = O / (dtype == 1 ? 8 : 4);
  const long nvec = (long)N * HW * V;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1)
    hipLaunchKernelGGL(sg2_act_fwd_kernel<__half>, dim3(ew_grid_capped(nvec)), dim3(256), 0, st, (const __half*)c, d, noise, (long)noise_bstride, strength, bias,
                       (__half*)y, (unsigned)((long)N * HW), (unsigned)HW, V, act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL(sg2_act_fwd_kernel<float>, dim3(ew_grid_capped(nvec)), dim3(256), 0, st, (const float*)c, d, noise, (long)noise_bstride, strength, bias,
                       (float*)y, (unsigned)((long)N * HW), (unsigned)HW, V, act, alpha, gain, clamp);
  return icg_check_launch();
}

template <typename T>
static int fir_act_launch(const void* x, const float* f, void* c, void* y, const float* d, const float* noise, long noise_bstride, const float* strength,
                          const float* bias, int N, int C, int H, int W, int fh, int fw, int padx0, int pady0, int flip, float fgain, int outH, int outW,
                          int act, float alpha, float gain, float clamp, hipStream_t st) {
  constexpr int TY = 4, TX = 1;      // (TX = 2: 35 instead of 28 window loads per 8 outputs but 148 VGPRs -- measured 9 % slower)
  const int CV = C / Sg<T>::VEC;
  const long total = (long)N * ((outH + TY - 1) / TY) * ((outW + TX - 1) / TX) * CV;
  ICG_REQUIRE(total < 0x7fffffffL);
  if (fh == 4 && fw == 4)
    hipLaunchKernelGGL((sg2_fir_act_kernel<T, TY, TX, 4>), dim3(ew_grid(total)), dim3(256), 0, st, (const T*)x, f, (T*)c, (T*)y, d, noise, noise_bstride,
                       strength, bias, N, H, W, CV, fh, fw, padx0, pady0, flip, fgain, outH, outW, act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL((sg2_fir_act_kernel<T, TY, TX, 0>), dim3(ew_grid(total)), dim3(256), 0, st, (const T*)x, f, (T*)c, (T*)y, d, noise, noise_bstride,
                       strength, bias, N, H, W, CV, fh, fw, padx0, pady0, flip, fgain, outH, outW, act, alpha, gain, clamp);
  return icg_check_launch();
}
extern "C" int icg_sg2_fir_act_fwd(const void* x, const float* f, void* c, void* y, const float* d, const float* noise, int64_t noise_bstride,
                                   const float* strength, const float* bias, int N, int C, int H, int W, int fh, int fw, int padx0, int padx1,
                                   int pady0, int pady1, int flip, float fgain, int outH, int outW, int act, float alpha, float gain, float clamp,
                                   int dtype, void* stream) {
  ICG_REQUIRE(x && f && y && N > 0 && H > 0 && W > 0 && fh >= 1 && fw >= 1 && fh * fw <= 64 && (act == 1 || act == 3));
  ICG_REQUIRE(icg_sg2_rows_applies(C, dtype) && al16(x) && al16(y) && al16(c));
  ICG_REQUIRE(outW == W + padx0 + padx1 - fw + 1 && outH == H + pady0 + pady1 - fh + 1 && outW >= 1 && outH >= 1);
  if (dtype == 1)
    return fir_act_launch<__half>(x, f, c, y, d, noise, (long)noise_bstride, strength, bias, N, C, H, W, fh, fw, padx0, pady0, flip, fgain, outH, outW,
                                  act, alpha, gain, clamp, (hipStream_t)stream);
  return fir_act_launch<float>(x, f, c, y, d, noise, (long)noise_bstride, strength, bias, N, C, H, W, fh, fw, padx0, pady0, flip, fgain, outH, outW, act,
                               alpha, gain, clamp, (hipStream_t)stream);
}

extern "C" size_t icg_sg2_rows_workspace_bytes(int N, int64_t HW, int C, int ncols, int dtype) {
  if (!icg_sg2_rows_applies(C, dtype) || N <= 0 || HW <= 0) return 0;
  int rpb, chunks;
  rows_geometry((long)HW, C / (dtype == 1 ? 8 : 4), &rpb, &chunks);
  return (size_t)N * chunks * ncols * sizeof(float);
}

// sums [N][2 O + 1] (per sample: dz | dz c | dz noise) and tot [2 O + 1] (over the batch): db = tot[0 .. O), dd = sums[:, O .. 2 O),
// d strength = tot[2 O]
extern "C" int icg_sg2_act_bwd(const void* dy, const void* y, const void* c, const float* d, const float* noise, int64_t noise_bstride, void* dc,
                               float* sums, float* tot, int N, int64_t HW, int O, int act, float alpha, float gain, float clamp, int dtype,
                               void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dy && y && N > 0 && HW > 0 && icg_sg2_rows_applies(O, dtype) && (act == 1 || act == 3) && workspace && (sums || tot));
  ICG_REQUIRE(al16(dy) && al16(y) && al16(c) && al16(dc) && (!d || c));
  const int ctot = 2 * O + 1;
  ICG_REQUIRE(workspace_bytes >= icg_sg2_rows_workspace_bytes(N, HW, O, ctot, dtype));
  const int V = O / (dtype == 1 ? 8 : 4);
  int rpb, chunks;
  rows_geometry((long)HW, V, &rpb, &chunks);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == 1)
    hipLaunchKernelGGL((sg2_rows_kernel<__half, 0>), dim3(N * chunks), dim3(256), 0, st, (const __half*)dy, (const __half*)y, (const __half*)c, d, noise,
                       (long)noise_bstride, (__half*)dc, part, ctot, (long)HW, V, rpb, chunks, act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL((sg2_rows_kernel<float, 0>), dim3(N * chunks), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)c, d, noise,
                       (long)noise_bstride, (float*)dc, part, ctot, (long)HW, V, rpb, chunks, act, alpha, gain, clamp);
  hipLaunchKernelGGL(sg2_rows_final_kernel, dim3((unsigned)icg_cdiv(ctot, 64)), dim3(1024), 0, st, part, N, chunks, ctot, sums, tot);
  return icg_check_launch();
}

extern "C" int icg_sg2_modulate_bwd(const void* dxs, const void* x, const float* s, void* dx, float* ds, int N, int64_t HW, int C, int dtype,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dxs && x && s && ds && N > 0 && HW > 0 && icg_sg2_rows_applies(C, dtype) && workspace && al16(dxs) && al16(x) && al16(dx));
  ICG_REQUIRE(workspace_bytes >= icg_sg2_rows_workspace_bytes(N, HW, C, C, dtype));
  const int V = C / (dtype == 1 ? 8 : 4);
  int rpb, chunks;
  rows_geometry((long)HW, V, &rpb, &chunks);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == 1)
    hipLaunchKernelGGL((sg2_rows_kernel<__half, 1>), dim3(N * chunks), dim3(256), 0, st, (const __half*)dxs, (const __half*)x, (const __half*)nullptr, s,
                       (const float*)nullptr, 0L, (__half*)dx, part, C, (long)HW, V, rpb, chunks, 1, 0.f, 1.f, -1.f);
  else
    hipLaunchKernelGGL((sg2_rows_kernel<float, 1>), dim3(N * chunks), dim3(256), 0, st, (const float*)dxs, (const float*)x, (const float*)nullptr, s,
                       (const float*)nullptr, 0L, (float*)dx, part, C, (long)HW, V, rpb, chunks, 1, 0.f, 1.f, -1.f);
  hipLaunchKernelGGL(sg2_rows_final_kernel, dim3((unsigned)icg_cdiv(C, 64)), dim3(1024), 0, st, part, N, chunks, C, ds, (float*)nullptr);
  return icg_check_launch();
}

extern "C" int icg_sg2_mod2(const void* x, const float* a, const void* g, const float* b, void* u, int N, int64_t HW, int C, int dtype, void* stream) {
  ICG_REQUIRE(x && a && u && (!g || b) && N > 0 && HW > 0 && icg_sg2_rows_applies(C, dtype) && al16(x) && al16(g) && al16(u) && (long)N * HW < 0x7fffffffL);
  const int V = C / (dtype == 1 ? 8 : 4);
  const long nvec = (long)N * HW * V;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1)
    hipLaunchKernelGGL(sg2_mod2_kernel<__half>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const __half*)x, a, (const __half*)g, b, (__half*)u,
                       (unsigned)((long)N * HW), (unsigned)HW, V);
  else
    hipLaunchKernelGGL(sg2_mod2_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)x, a, (const float*)g, b, (float*)u,
                       (unsigned)((long)N * HW), (unsigned)HW, V);
  return icg_check_launch();
}

// sums [N][O] = sum_p cdc * dz;  workspace: icg_sg2_rows_workspace_bytes(N, HW, O, O, dtype)
extern "C" int icg_sg2_act_bwd2(const void* dy, const void* y, const void* c, const void* cdc, const float* d, const float* cdd, void* cdy, void* cc,
                                float* sums, int N, int64_t HW, int O, int act, float alpha, float gain, float clamp, int dtype, void* workspace,
                                size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dy && y && cdc && cdy && sums && N > 0 && HW > 0 && icg_sg2_rows_applies(O, dtype) && (act == 1 || act == 3) && workspace);
  ICG_REQUIRE((!cdd || c) && al16(dy) && al16(y) && al16(c) && al16(cdc) && al16(cdy) && al16(cc));
  ICG_REQUIRE(workspace_bytes >= icg_sg2_rows_workspace_bytes(N, HW, O, O, dtype));
  const int V = O / (dtype == 1 ? 8 : 4);
  int rpb, chunks;
  rows_geometry((long)HW, V, &rpb, &chunks);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == 1)
    hipLaunchKernelGGL(sg2_act_bwd2_kernel<__half>, dim3(N * chunks), dim3(256), 0, st, (const __half*)dy, (const __half*)y, (const __half*)c,
                       (const __half*)cdc, d, cdd, (__half*)cdy, (__half*)cc, part, (long)HW, V, rpb, chunks, act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL(sg2_act_bwd2_kernel<float>, dim3(N * chunks), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)c,
                       (const float*)cdc, d, cdd, (float*)cdy, (float*)cc, part, (long)HW, V, rpb, chunks, act, alpha, gain, clamp);
  hipLaunchKernelGGL(sg2_rows_final_kernel, dim3((unsigned)icg_cdiv(O, 64)), dim3(1024), 0, st, part, N, chunks, O, sums, (float*)nullptr);
  return icg_check_launch();
}

extern "C" int icg_sg2_style_bwd(const float* ds_mod, int64_t ds_stride, const float* dd, int64_t dd_stride, const float* d, const float* s,
                                 const float* wsq, int N, int I, int O, float* g, float* pdot, float* t, void* stream) {
  ICG_REQUIRE(ds_mod && s && g && pdot && N > 0 && I > 0 && (!dd || (d && wsq && t && O > 0 && O <= 8192)));
  const dim3 grid(N, (unsigned)icg_cdiv(I, 64));
  const size_t lds = ((size_t)(dd ? O : 0) + 256) * sizeof(float);
  hipLaunchKernelGGL(sg2_style_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, ds_mod, (long)ds_stride, dd, (long)dd_stride, d, s, wsq, I,
                     dd ? O : 0, g, pdot, t);
  return icg_check_launch();
}

extern "C" int icg_sg2_fc_bwd(const float* g, const float* smax, const int* sarg, const float* pdot, int npdot, float post_gain, const float* x,
                              const float* W, int N, int I, int K, float wgain, float bias_gain, float* dW, float* db, float* dx, void* stream) {
  ICG_REQUIRE(g && N > 0 && N <= FCB_NMAX && I > 0 && K > 0 && (!smax || (sarg && pdot && npdot > 0)) && (!dW || x) && (!dx || W));
  SgNorm nm{smax, sarg, pdot, npdot, post_gain};
  hipStream_t st = (hipStream_t)stream;
  if (dW || db)
    hipLaunchKernelGGL(sg2_fc_bwd_dw_kernel, dim3((unsigned)icg_cdiv(I, 8)), dim3(256), 0, st, g, nm, x, N, I, K, wgain, bias_gain, dW, db);
  if (dx)
    hipLaunchKernelGGL(sg2_fc_bwd_dx_kernel, dim3((unsigned)icg_cdiv(K, 32), (unsigned)icg_cdiv(N, 16)), dim3(256), 0, st, g, nm, W, N, I, K, wgain, dx);
  return icg_check_launch();
}

extern "C" size_t icg_sg2_weight_bwd_workspace_bytes(int O, int I) { return (size_t)O * icg_cdiv(I, 32) * sizeof(float); }

extern "C" int icg_sg2_weight_bwd(const float* dw_conv, int layout, const float* t, const float* s, int N, const float* w, const float* wscale,
                                  const int* warg, int prenorm, float c0, int round_f16, float* dw, int O, int I, int R, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return icg_sg2_weight_bwd_q(dw_conv, layout, t, s, N, nullptr, w, wscale, warg, prenorm, c0, round_f16, dw, O, I, R, workspace, workspace_bytes, stream);
}

extern "C" int icg_sg2_weight_bwd_q(const float* dw_conv, int layout, const float* t, const float* s, int N, const float* Q, const float* w,
                                    const float* wscale, const int* warg, int prenorm, float c0, int round_f16, float* dw, int O, int I, int R,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dw_conv && w && wscale && dw && O > 0 && I > 0 && R >= 1 && R <= 3 && (layout == 0 || layout == 1) && (!t || (s && N > 0 && N <= 64)));
  ICG_REQUIRE(!prenorm || (warg && workspace && workspace_bytes >= icg_sg2_weight_bwd_workspace_bytes(O, I)));
  SgWb p{dw_conv, Q, t, s, w, wscale, dw, prenorm ? (float*)workspace : nullptr, layout, t ? N : 0, O, I, R * R, round_f16, (int)icg_cdiv(I, 32)};
  const size_t lds = ((size_t)32 * (32 * p.RR + 1) + (size_t)p.N * 64) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sg2_weight_bwd_kernel, dim3((unsigned)(icg_cdiv(O, 32) * p.tiles_i)), dim3(256), lds, st, p);
  if (prenorm)
    hipLaunchKernelGGL(sg2_weight_bwd_fix_kernel, dim3((unsigned)icg_cdiv(O, 256)), dim3(256), 0, st, (const float*)workspace, p.tiles_i, w, wscale, warg,
                       c0, O, I * p.RR, dw);
  return icg_check_launch();
}

// y [N][HW][O] = clamp(gain * act(x . w + bias)) from the planar image x [N][3][HW] (same storage type), w [O][3] (prepared: gain folded, storage type)
extern "C" int icg_sg2_fromrgb_applies(int O, int dtype) {
  const int vec = dtype == 1 ? 8 : 4;
  if (!icg_sg2_rows_applies(O, dtype)) return 0;
  return (O / vec) <= 64 ? 1 : 0;               // a pixel's channel vectors sit in one wavefront (the image gradient is reduced by shuffles)
}
extern "C" int icg_sg2_fromrgb_fwd(const void* x, const void* w, const float* bias, void* y, int N, int64_t HW, int O, int act, float alpha,
                                   float gain, float clamp, int dtype, void* stream) {
  ICG_REQUIRE(x && w && y && N > 0 && HW > 0 && icg_sg2_fromrgb_applies(O, dtype) && (act == 1 || act == 3) && al16(y) && al16(w) && (long)N * HW < 0x7fffffffL);
  const int V = O / (dtype == 1 ? 8 : 4);
  const long nvec = (long)N * HW * V;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1)
    hipLaunchKernelGGL(sg2_fromrgb_fwd_kernel<__half>, dim3(ew_grid_capped(nvec)), dim3(256), 0, st, (const __half*)x, (const __half*)w, bias, (__half*)y,
                       (unsigned)((long)N * HW), (unsigned)HW, V, act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL(sg2_fromrgb_fwd_kernel<float>, dim3(ew_grid_capped(nvec)), dim3(256), 0, st, (const float*)x, (const float*)w, bias, (float*)y,
                       (unsigned)((long)N * HW), (unsigned)HW, V, act, alpha, gain, clamp);
  return icg_check_launch();
}
// tot [4 O]: (d w[o][0..2], d bias[o]) at 4 o + k, with respect to the PREPARED weight;  dimg [N][3][HW] (may be NULL).
// workspace: icg_sg2_rows_workspace_bytes(N, HW, O, 4 O, dtype)
extern "C" int icg_sg2_fromrgb_bwd(const void* dy, const void* y, const void* x, const void* w, void* dimg, float* tot, int N, int64_t HW, int O,
                                   int act, float alpha, float gain, float clamp, int dtype, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  ICG_REQUIRE(dy && y && x && w && tot && N > 0 && HW > 0 && icg_sg2_fromrgb_applies(O, dtype) && (act == 1 || act == 3) && workspace);
  ICG_REQUIRE(al16(dy) && al16(y) && workspace_bytes >= icg_sg2_rows_workspace_bytes(N, HW, O, 4 * O, dtype));
  const int V = O / (dtype == 1 ? 8 : 4);
  int rpb, chunks;
  rows_geometry((long)HW, V, &rpb, &chunks);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (dtype == 1)
    hipLaunchKernelGGL(sg2_fromrgb_bwd_kernel<__half>, dim3(N * chunks), dim3(256), 0, st, (const __half*)dy, (const __half*)y, (const __half*)x,
                       (const __half*)w, (__half*)dimg, part, (long)HW, V, rpb, chunks, act, alpha, gain, clamp);
  else
    hipLaunchKernelGGL(sg2_fromrgb_bwd_kernel<float>, dim3(N * chunks), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)x,
                       (const float*)w, (float*)dimg, part, (long)HW, V, rpb, chunks, act, alpha, gain, clamp);
  hipLaunchKernelGGL(sg2_rows_final_kernel, dim3((unsigned)icg_cdiv(4 * O, 64)), dim3(1024), 0, st, part, N, chunks, 4 * O, (float*)nullptr, tot);
  return icg_check_launch();
}

extern "C" int icg_sg2_torgb_applies(int C, int dtype) {
  const int vec = dtype == 1 ? 8 : 4;
  if ((dtype != 0 && dtype != 1) || C < vec || C % vec != 0) return 0;
  const int V = C / vec;
  if (!pow2(V)) return 0;
  return (V <= 64 || V == 128 || V == 256) ? 1 : 0;
}

template <typename T>
static int torgb_fwd_launch(const void* x, const float* s, const float* w, const float* bias, float clamp, const float* img_in, float* img_out, void* y,
                            int N, long HW, int C, hipStream_t st) {
  const int V = C / Sg<T>::VEC, L = V < 64 ? V : 64, NV = V / L;
  int rpb, chunks;
  rows_geometry(HW, L, &rpb, &chunks);
  const dim3 grid(N * chunks), block(256);
#define ICG_TORGB_F(NVV)                                                                                                                             \
  hipLaunchKernelGGL((sg2_torgb_fwd_kernel<T, NVV>), grid, block, 0, st, (const T*)x, s, w, bias, clamp, img_in, img_out, (T*)y, HW, C, L, rpb, chunks)
  if (NV == 1) ICG_TORGB_F(1);
  else if (NV == 2) ICG_TORGB_F(2);
  else ICG_TORGB_F(4);
#undef ICG_TORGB_F
  return icg_check_launch();
}
extern "C" int icg_sg2_torgb_fwd(const void* x, const float* s, const float* w, const float* bias, float clamp, const float* img_in, float* img_out,
                                 void* y, int N, int64_t HW, int C, int dtype, void* stream) {
  ICG_REQUIRE(x && s && w && img_out && y && N > 0 && HW > 0 && icg_sg2_torgb_applies(C, dtype) && al16(x));
  if (dtype == 1) return torgb_fwd_launch<__half>(x, s, w, bias, clamp, img_in, img_out, y, N, (long)HW, C, (hipStream_t)stream);
  return torgb_fwd_launch<float>(x, s, w, bias, clamp, img_in, img_out, y, N, (long)HW, C, (hipStream_t)stream);
}

extern "C" size_t icg_sg2_torgb_bwd_workspace_bytes(int N, int64_t HW, int C, int dtype) {
  if (!icg_sg2_torgb_applies(C, dtype) || N <= 0 || HW <= 0) return 0;
  const int V = C / (dtype == 1 ? 8 : 4), L = V < 64 ? V : 64;
  int rpb, chunks;
  rows_geometry((long)HW, L, &rpb, &chunks);
  return (size_t)N * chunks * (4 * C + 3) * sizeof(float);
}
template <typename T>
static int torgb_bwd_launch(const float* dimg, const void* y, const void* x, const float* s, const float* w, float clamp, int mask_clamp, void* dx,
                            float* sums, float* tot, int N, long HW, int C, float* part, hipStream_t st) {
  const int V = C / Sg<T>::VEC, L = V < 64 ? V : 64, NV = V / L;
  int rpb, chunks;
  rows_geometry(HW, L, &rpb, &chunks);
  const dim3 grid(N * chunks), block(256);
#define ICG_TORGB_B(NVV)                                                                                                                             \
  hipLaunchKernelGGL((sg2_torgb_bwd_kernel<T, NVV>), grid, block, 0, st, dimg, (const T*)y, (const T*)x, s, w, clamp, mask_clamp, (T*)dx, part, HW, C, L, \
                     rpb, chunks)
  if (NV == 1) ICG_TORGB_B(1);
  else if (NV == 2) ICG_TORGB_B(2);
  else ICG_TORGB_B(4);
#undef ICG_TORGB_B
  const int ctot = 4 * C + 3;
  hipLaunchKernelGGL(sg2_rows_final_kernel, dim3((unsigned)icg_cdiv(ctot, 64)), dim3(1024), 0, st, part, N, chunks, ctot, sums, tot);
  return icg_check_launch();
}
template <typename T>
static int torgb_bwd2_launch(const float* dimg, const void* y, const void* x, const float* s, const float* w, const float* a, const void* cdx,
                             const float* cim, float clamp, int mask_clamp, float* cdimg, void* cx, float* sums, float* tot, int N, long HW, int C,
                             float* part, hipStream_t st) {
  const int V = C / Sg<T>::VEC, L = V < 64 ? V : 64, NV = V / L;
  int rpb, chunks;
  rows_geometry(HW, L, &rpb, &chunks);
  const dim3 grid(N * chunks), block(256);
#define ICG_TORGB_B2(NVV)                                                                                                                            \
  hipLaunchKernelGGL((sg2_torgb_bwd2_kernel<T, NVV>), grid, block, 0, st, dimg, (const T*)y, (const T*)x, s, w, a, (const T*)cdx, cim, clamp, mask_clamp, \
                     cdimg, (T*)cx, part, HW, C, L, rpb, chunks)
  if (NV == 1) ICG_TORGB_B2(1);
  else if (NV == 2) ICG_TORGB_B2(2);
  else ICG_TORGB_B2(4);
#undef ICG_TORGB_B2
  hipLaunchKernelGGL(sg2_rows_final_kernel, dim3((unsigned)icg_cdiv(4 * C, 64)), dim3(1024), 0, st, part, N, chunks, 4 * C, sums, tot);
  return icg_check_launch();
}
// sums [N][4 C] (cot s = sums[:, 0 .. C)), tot [4 C] (cot w[o][c] = tot[(1 + o) C + c]); workspace: icg_sg2_torgb_bwd_workspace_bytes
extern "C" int icg_sg2_torgb_bwd2(const float* dimg, const void* y, const void* x, const float* s, const float* w, const float* a, const void* cdx,
                                  const float* cim, float clamp, int mask_clamp, float* cdimg, void* cx, float* sums, float* tot, int N, int64_t HW,
                                  int C, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dimg && y && x && s && w && a && cdimg && sums && tot && N > 0 && HW > 0 && icg_sg2_torgb_applies(C, dtype) && al16(x) && al16(cdx) &&
              al16(cx) && workspace);
  ICG_REQUIRE(workspace_bytes >= icg_sg2_torgb_bwd_workspace_bytes(N, HW, C, dtype));
  if (dtype == 1)
    return torgb_bwd2_launch<__half>(dimg, y, x, s, w, a, cdx, cim, clamp, mask_clamp, cdimg, cx, sums, tot, N, (long)HW, C, (float*)workspace,
                                     (hipStream_t)stream);
  return torgb_bwd2_launch<float>(dimg, y, x, s, w, a, cdx, cim, clamp, mask_clamp, cdimg, cx, sums, tot, N, (long)HW, C, (float*)workspace,
                                  (hipStream_t)stream);
}

// sums [N][4 C + 3]: ds = sums[:, 0 .. C);  tot [4 C + 3]: dw[o][c] = tot[(1 + o) C + c], db[o] = tot[4 C + o]
extern "C" int icg_sg2_torgb_bwd(const float* dimg, const void* y, const void* x, const float* s, const float* w, float clamp, int mask_clamp, void* dx,
                                 float* sums, float* tot, int N, int64_t HW, int C, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dimg && y && x && s && w && sums && tot && N > 0 && HW > 0 && icg_sg2_torgb_applies(C, dtype) && al16(x) && al16(dx) && workspace);
  ICG_REQUIRE(workspace_bytes >= icg_sg2_torgb_bwd_workspace_bytes(N, HW, C, dtype));
  if (dtype == 1)
    return torgb_bwd_launch<__half>(dimg, y, x, s, w, clamp, mask_clamp, dx, sums, tot, N, (long)HW, C, (float*)workspace, (hipStream_t)stream);
  return torgb_bwd_launch<float>(dimg, y, x, s, w, clamp, mask_clamp, dx, sums, tot, N, (long)HW, C, (float*)workspace, (hipStream_t)stream);
}
