// HBM-bound glue of the IC-GAN BigGAN step that is not folded into a convolution prologue/epilogue:
// layout change at the module boundary, tanh tail (BigGAN.py:386), 2x2 avg/max pooling (BigGAN.py:528,
// layers.py:230-231), the attention row softmax (layers.py:237) with wavefront-shuffle reductions,
// relu+sum-pool head (BigGAN.py:625), gamma*o + x (layers.py:244) and column sums for bias gradients.
// Everything is NHWC fp32; kernels are grid-stride with 16-byte-per-lane accesses where C % 4 == 0.
#include "icg_common.h"

// a FULL grid (one block-iteration per workgroup) up to 2^22 workgroups: measured with hand-written copies (tools/hbm_bench.py) a
// one-iteration grid streams at 6.35 TB/s on MI355X, grid-stride loops over 2048 - 8192 workgroups at 4.8 - 5.5
#define GRID_1D(n, per) ((unsigned)(icg_cdiv((n), (per)) > ICG_GRID_CAP ? ICG_GRID_CAP : (icg_cdiv((n), (per)) < 1 ? 1 : icg_cdiv((n), (per)))))

// ---------------------------------------------------------------- batched transpose  [b][R][S] -> [b][S][R]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R,
                                                        int S) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (long)b * R * S;
  float* yb = y + (long)b * R * S;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, s = s0 + tx;
    if (r < R && s < S) tile[ty + 8 * k][tx] = xb[(long)r * S + s];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int s = s0 + ty + 8 * k, r = r0 + tx;
    if (r < R && s < S) yb[(long)s * R + r] = tile[tx][ty + 8 * k];
  }
}

static int transpose_batched(const float* x, float* y, int batch, int R, int S, hipStream_t st) {
  if (batch <= 0 || R <= 0 || S <= 0 || batch > 65535) return ICG_ERR_ARG;
  const long gy = icg_cdiv(R, 32), gx = icg_cdiv(S, 32);
  if (gy > 65535) return ICG_ERR_ARG;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy, batch), dim3(256), 0, st, x, y, R, S);
  return icg_check_launch();
}

extern "C" int icg_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  ICG_REQUIRE(x && y);
  return transpose_batched(x, y, B, C, H * W, (hipStream_t)stream);
}
extern "C" int icg_nhwc_to_nchw(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  ICG_REQUIRE(x && y);
  return transpose_batched(x, y, B, H * W, C, (hipStream_t)stream);
}

// ---------------------------------------------------------------- elementwise
template <int OP>
__device__ __forceinline__ float4 ew_apply(const float4 va, const float4 vb) {
  float4 o;
  if (OP == 0) {  // tanh
    o.x = tanhf(va.x); o.y = tanhf(va.y); o.z = tanhf(va.z); o.w = tanhf(va.w);
  } else if (OP == 1) {  // tanh bwd: a = y, b = dy
    o.x = vb.x * (1.f - va.x * va.x); o.y = vb.y * (1.f - va.y * va.y);
    o.z = vb.z * (1.f - va.z * va.z); o.w = vb.w * (1.f - va.w * va.w);
  } else if (OP == 2) {  // relu bwd: a = x, b = dy
    o.x = va.x > 0.f ? vb.x : 0.f; o.y = va.y > 0.f ? vb.y : 0.f;
    o.z = va.z > 0.f ? vb.z : 0.f; o.w = va.w > 0.f ? vb.w : 0.f;
  } else if (OP == 3) {  // relu
    o.x = fmaxf(va.x, 0.f); o.y = fmaxf(va.y, 0.f); o.z = fmaxf(va.z, 0.f); o.w = fmaxf(va.w, 0.f);
  } else {  // add
    o.x = va.x + vb.x; o.y = va.y + vb.y; o.z = va.z + vb.z; o.w = va.w + vb.w;
  }
  return o;
}

// EW_U float4 per thread and block-iteration, every load of the iteration issued before the first store: one 16-byte load per
// thread in flight (the round-1 form) keeps ~32 KB per CU on the wire, short of what 6 TB/s x ~2 us of latency needs
// (tools/hbm_bench.py: hand-written copies x1 / x4)
#define EW_U 4
template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ y, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long n4 = n >> 2;
  const long step = stride * EW_U;
  long base = (long)blockIdx.x * blockDim.x * EW_U + threadIdx.x;
  for (; base + (long)(EW_U - 1) * 256 < n4; base += step) {
    float4 va[EW_U], vb[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) {
      va[u] = reinterpret_cast<const float4*>(a)[base + u * 256];
      vb[u] = (OP != 0 && OP != 3) ? reinterpret_cast<const float4*>(b)[base + u * 256] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < EW_U; ++u) reinterpret_cast<float4*>(y)[base + u * 256] = ew_apply<OP>(va[u], vb[u]);
  }
  for (int u = 0; u < EW_U; ++u) {          // ragged end of the float4 range
    const long i = base + u * 256;
    if (i < n4) {
      const float4 va = reinterpret_cast<const float4*>(a)[i];
      const float4 vb = (OP != 0 && OP != 3) ? reinterpret_cast<const float4*>(b)[i] : make_float4(0, 0, 0, 0);
      reinterpret_cast<float4*>(y)[i] = ew_apply<OP>(va, vb);
    }
  }
  // tail
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float va = a[i];
    const float vb = (OP != 0 && OP != 3) ? b[i] : 0.f;
    float o;
    if (OP == 0) o = tanhf(va);
    else if (OP == 1) o = vb * (1.f - va * va);
    else if (OP == 2) o = va > 0.f ? vb : 0.f;
    else if (OP == 3) o = fmaxf(va, 0.f);
    else o = va + vb;
    y[i] = o;
  }
}

template <int OP>
static int launch_ew(const float* a, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !y || n <= 0) return ICG_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b)) & 15)
    return ICG_ERR_ARG;
  hipLaunchKernelGGL(ew_kernel<OP>, dim3(GRID_1D(n, 1024 * EW_U)), dim3(256), 0, (hipStream_t)stream, a, b, y, (long)n);
  return icg_check_launch();
}

extern "C" int icg_tanh_fwd(const float* x, float* y, int64_t n, void* stream) { return launch_ew<0>(x, nullptr, y, n, stream); }
extern "C" int icg_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream) {
  ICG_REQUIRE(dy);
  return launch_ew<1>(y, dy, dx, n, stream);
}
extern "C" int icg_relu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  ICG_REQUIRE(dy);
  return launch_ew<2>(x, dy, dx, n, stream);
}
extern "C" int icg_relu_fwd(const float* x, float* y, int64_t n, void* stream) { return launch_ew<3>(x, nullptr, y, n, stream); }
extern "C" int icg_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  ICG_REQUIRE(b);
  return launch_ew<4>(a, b, y, n, stream);
}

// ---------------------------------------------------------------- 2x2 pooling (NHWC, scalar over c: C may be any)
// MODE 0: avg fwd (+add)  1: avg bwd  2: max fwd  3: max bwd
template <int MODE>
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x, const float* __restrict__ aux,
                                                   const float* __restrict__ dy, float* __restrict__ out, int B, int H,
                                                   int W, int C, float scale) {
  // (H, W) = full-resolution dims; pooled dims are H/2, W/2.  One thread per pooled element.
  const int Hp = H >> 1, Wp = W >> 1;
  const long total = (long)B * Hp * Wp * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wp = (int)(t % Wp);
    t /= Wp;
    const int hp = (int)(t % Hp);
    const long b = t / Hp;
    const long base = ((b * H + 2 * hp) * W + 2 * wp) * C + c;
    const long o01 = C, o10 = (long)W * C, o11 = (long)W * C + C;
    if (MODE == 0) {
      float v = ((x[base] + x[base + o01]) + (x[base + o10] + x[base + o11])) * scale;
      if (aux) v += aux[i];
      out[i] = v;
    } else if (MODE == 1) {
      const float g = scale * dy[i];
      if (aux) {      // + running gradient at the full resolution (icg_avgpool2_bwd_add)
        out[base] = g + aux[base]; out[base + o01] = g + aux[base + o01];
        out[base + o10] = g + aux[base + o10]; out[base + o11] = g + aux[base + o11];
      } else {
        out[base] = g; out[base + o01] = g; out[base + o10] = g; out[base + o11] = g;
      }
    } else if (MODE == 2) {
      float m = x[base];
      m = fmaxf(m, x[base + o01]);
      m = fmaxf(m, x[base + o10]);
      m = fmaxf(m, x[base + o11]);
      out[i] = m;
    } else {
      const float v0 = x[base], v1 = x[base + o01], v2 = x[base + o10], v3 = x[base + o11];
      int arg = 0;
      float m = v0;
      if (v1 > m) { m = v1; arg = 1; }
      if (v2 > m) { m = v2; arg = 2; }
      if (v3 > m) { m = v3; arg = 3; }
      const float g = dy[i];
      out[base] = arg == 0 ? g : 0.f;
      out[base + o01] = arg == 1 ? g : 0.f;
      out[base + o10] = arg == 2 ? g : 0.f;
      out[base + o11] = arg == 3 ? g : 0.f;
    }
  }
}

template <int MODE>
static int launch_pool(const float* x, const float* aux, const float* dy, float* out, int B, int H, int W, int C,
                       void* stream, float scale = 0.25f) {
  if (!out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return ICG_ERR_ARG;
  const long total = (long)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(pool_kernel<MODE>, dim3(GRID_1D(total, 256)), dim3(256), 0, (hipStream_t)stream, x, aux, dy, out,
                     B, H, W, C, scale);
  return icg_check_launch();
}

extern "C" int icg_sumpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
  ICG_REQUIRE(x);
  return launch_pool<0>(x, nullptr, nullptr, y, B, H, W, C, stream, 1.0f);
}
extern "C" int icg_avgpool2_fwd(const float* x, const float* add, float* y, int B, int H, int W, int C, void* stream) {
  ICG_REQUIRE(x);
  return launch_pool<0>(x, add, nullptr, y, B, H, W, C, stream);
}
extern "C" int icg_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
  ICG_REQUIRE(dy);
  return launch_pool<1>(nullptr, nullptr, dy, dx, B, H, W, C, stream);
}
extern "C" int icg_avgpool2_bwd_add(const float* dy, const float* carry, float* dx, int B, int H, int W, int C, void* stream) {
  ICG_REQUIRE(dy && carry);
  return launch_pool<1>(nullptr, carry, dy, dx, B, H, W, C, stream);
}
extern "C" int icg_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
  ICG_REQUIRE(x);
  return launch_pool<2>(x, nullptr, nullptr, y, B, H, W, C, stream);
}
extern "C" int icg_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
  ICG_REQUIRE(x && dy);
  return launch_pool<3>(x, nullptr, dy, dx, B, H, W, C, stream);
}

// ---------------------------------------------------------------- row softmax: one wavefront per row
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                          int cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < rows; r += nwaves) {
    const float* xr = x + r * cols;
    float* yr = y + r * cols;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, xr[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) s += expf(xr[j] - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int j = lane; j < cols; j += 64) yr[j] = expf(xr[j] - m) * inv;
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < rows; r += nwaves) {
    const float* yr = y + r * cols;
    const float* gr = dy + r * cols;
    float* dr = dx + r * cols;
    float s = 0.f;
    for (int j = lane; j < cols; j += 64) s = fmaf(yr[j], gr[j], s);
    s = wave_sum(s);
    for (int j = lane; j < cols; j += 64) dr[j] = yr[j] * (gr[j] - s);
  }
}

// Register-resident rows: cols == 256 * NV, one wavefront per row, every element read once and written once with
// 16-byte accesses; max / sum by wave shuffles (the attention maps of the 64x64 blocks have cols = 1024 -> NV = 4).
template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_reg_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                              long rows) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < rows; r += nwaves) {
    const float4* xr = x + r * (64 * NV);
    float4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k] = xr[k * 64 + lane];
      m = fmaxf(fmaxf(m, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k].x = expf(v[k].x - m); v[k].y = expf(v[k].y - m); v[k].z = expf(v[k].z - m); v[k].w = expf(v[k].w - m);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    float4* yr = y + r * (64 * NV);
#pragma unroll
    for (int k = 0; k < NV; ++k)
      yr[k * 64 + lane] = make_float4(v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_reg_kernel(const float4* __restrict__ y, const float4* __restrict__ dy,
                                                              float4* __restrict__ dx, long rows) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  for (long r = wave; r < rows; r += nwaves) {
    const float4* yr = y + r * (64 * NV);
    const float4* gr = dy + r * (64 * NV);
    float4 a[NV], g[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      a[k] = yr[k * 64 + lane];
      g[k] = gr[k * 64 + lane];
      s += (a[k].x * g[k].x + a[k].y * g[k].y) + (a[k].z * g[k].z + a[k].w * g[k].w);
    }
    s = wave_sum(s);
    float4* dr = dx + r * (64 * NV);
#pragma unroll
    for (int k = 0; k < NV; ++k)
      dr[k * 64 + lane] = make_float4(a[k].x * (g[k].x - s), a[k].y * (g[k].y - s), a[k].z * (g[k].z - s),
                                      a[k].w * (g[k].w - s));
  }
}

static bool softmax_reg_ok(const void* a, const void* b, const void* c, int cols) {
  return cols % 256 == 0 && cols / 256 >= 1 && cols / 256 <= 8 && ((cols / 256) & (cols / 256 - 1)) == 0 &&
         ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

extern "C" int icg_softmax_fwd(const float* x, float* y, int64_t rows, int cols, void* stream) {
  ICG_REQUIRE(x && y && rows > 0 && cols > 0);
  long blocks = icg_cdiv(rows, 4);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
  if (softmax_reg_ok(x, y, nullptr, cols)) {
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (cols / 256) {
      case 1: hipLaunchKernelGGL(softmax_fwd_reg_kernel<1>, grid, blk, 0, st, (const float4*)x, (float4*)y, (long)rows); break;
      case 2: hipLaunchKernelGGL(softmax_fwd_reg_kernel<2>, grid, blk, 0, st, (const float4*)x, (float4*)y, (long)rows); break;
      case 4: hipLaunchKernelGGL(softmax_fwd_reg_kernel<4>, grid, blk, 0, st, (const float4*)x, (float4*)y, (long)rows); break;
      default: hipLaunchKernelGGL(softmax_fwd_reg_kernel<8>, grid, blk, 0, st, (const float4*)x, (float4*)y, (long)rows); break;
    }
    return icg_check_launch();
  }
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, (long)rows,
                     cols);
  return icg_check_launch();
}
extern "C" int icg_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int cols, void* stream) {
  ICG_REQUIRE(y && dy && dx && rows > 0 && cols > 0);
  long blocks = icg_cdiv(rows, 4);
  if (blocks > ICG_GRID_CAP) blocks = ICG_GRID_CAP;
  if (softmax_reg_ok(y, dy, dx, cols)) {
    const dim3 grid((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (cols / 256) {
      case 1: hipLaunchKernelGGL(softmax_bwd_reg_kernel<1>, grid, blk, 0, st, (const float4*)y, (const float4*)dy, (float4*)dx, (long)rows); break;
      case 2: hipLaunchKernelGGL(softmax_bwd_reg_kernel<2>, grid, blk, 0, st, (const float4*)y, (const float4*)dy, (float4*)dx, (long)rows); break;
      case 4: hipLaunchKernelGGL(softmax_bwd_reg_kernel<4>, grid, blk, 0, st, (const float4*)y, (const float4*)dy, (float4*)dx, (long)rows); break;
      default: hipLaunchKernelGGL(softmax_bwd_reg_kernel<8>, grid, blk, 0, st, (const float4*)y, (const float4*)dy, (float4*)dx, (long)rows); break;
    }
    return icg_check_launch();
  }
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, dy, dx,
                     (long)rows, cols);
  return icg_check_launch();
}

// ---------------------------------------------------------------- relu + spatial sum pool
__global__ void relu_sumpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C) return;
  const int c = (int)(i % C);
  const long b = i / C;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += fmaxf(x[(b * HW + p) * C + c], 0.f);
  y[i] = s;
}
__global__ void relu_sumpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                        float* __restrict__ dx, int B, int HW, int C) {
  const long total = (long)B * HW * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const long b = i / ((long)HW * C);
    dx[i] = x[i] > 0.f ? dy[b * C + c] : 0.f;
  }
}
extern "C" int icg_relu_sumpool_fwd(const float* x, float* y, int B, int HW, int C, void* stream) {
  ICG_REQUIRE(x && y && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(relu_sumpool_fwd_kernel, dim3((unsigned)icg_cdiv((long)B * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, B, HW, C);
  return icg_check_launch();
}
extern "C" int icg_relu_sumpool_bwd(const float* x, const float* dy, float* dx, int B, int HW, int C, void* stream) {
  ICG_REQUIRE(x && dy && dx && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(relu_sumpool_bwd_kernel, dim3(GRID_1D((long)B * HW * C, 256)), dim3(256), 0, (hipStream_t)stream,
                     x, dy, dx, B, HW, C);
  return icg_check_launch();
}

// ---------------------------------------------------------------- out = gamma*o + x
__global__ void scale_add_fwd_kernel(const float* __restrict__ gamma, const float* __restrict__ o,
                                     const float* __restrict__ x, float* __restrict__ out, long n) {
  const float g = gamma[0];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fmaf(g, o[i], x[i]);
}
// 16-byte loads when the three arrays allow it (two quads in flight per thread); the per-thread sum stays fp64 and the order fixed
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const float* __restrict__ gamma,
                                                            const float* __restrict__ o,
                                                            const float* __restrict__ dout, float* __restrict__ d_o,
                                                            double* __restrict__ part, long n, int vec) {
  __shared__ double red[4];
  const float g = gamma[0];
  const long stride = (long)gridDim.x * blockDim.x;
  const long g0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (vec) {
    const long n4 = n >> 2;
    const float4* __restrict__ o4 = reinterpret_cast<const float4*>(o);
    const float4* __restrict__ d4 = reinterpret_cast<const float4*>(dout);
    float4* __restrict__ r4 = reinterpret_cast<float4*>(d_o);
    long i = g0;
    for (; i + stride < n4; i += 2 * stride) {
      const float4 da = d4[i], db = d4[i + stride], oa = o4[i], ob = o4[i + stride];
      acc += ((double)da.x * (double)oa.x + (double)da.y * (double)oa.y) + ((double)da.z * (double)oa.z + (double)da.w * (double)oa.w);
      acc += ((double)db.x * (double)ob.x + (double)db.y * (double)ob.y) + ((double)db.z * (double)ob.z + (double)db.w * (double)ob.w);
      r4[i] = make_float4(g * da.x, g * da.y, g * da.z, g * da.w);
      r4[i + stride] = make_float4(g * db.x, g * db.y, g * db.z, g * db.w);
    }
    if (i < n4) {
      const float4 da = d4[i], oa = o4[i];
      acc += ((double)da.x * (double)oa.x + (double)da.y * (double)oa.y) + ((double)da.z * (double)oa.z + (double)da.w * (double)oa.w);
      r4[i] = make_float4(g * da.x, g * da.y, g * da.z, g * da.w);
    }
  } else {
    for (long i = g0; i < n; i += stride) {
      const float d = dout[i];
      acc += (double)d * (double)o[i];
      d_o[i] = g * d;
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// out[0] = sum of n block partials: one wave, lane l takes partials l, l + 64, ... in order, then the wave fold
__global__ __launch_bounds__(64) void sum_parts_kernel(const double* __restrict__ part, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += 64) s += part[k];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)s;
}
extern "C" int icg_scale_add_fwd(const float* gamma, const float* o, const float* x, float* out, int64_t n,
                                 void* stream) {
  ICG_REQUIRE(gamma && o && x && out && n > 0);
  hipLaunchKernelGGL(scale_add_fwd_kernel, dim3(GRID_1D(n, 256)), dim3(256), 0, (hipStream_t)stream, gamma, o, x, out,
                     (long)n);
  return icg_check_launch();
}
extern "C" int icg_scale_add_bwd(const float* gamma, const float* o, const float* dout, float* d_o, float* dgamma,
                                 int64_t n, void* scratch, size_t scratch_bytes, void* stream) {
  ICG_REQUIRE(gamma && o && dout && d_o && dgamma && scratch && n > 0);
  if (scratch_bytes < 512 * sizeof(double)) return ICG_ERR_WORKSPACE;
  long cap = (long)(scratch_bytes / sizeof(double));       // 512 partials at least; up to 2048 when the scratch holds them
  if (cap > 2048) cap = 2048;
  const int blocks = (int)(icg_cdiv(n, 2048) > cap ? cap : icg_cdiv(n, 2048));
  const int vec = ((n & 3) == 0) && ((((uintptr_t)o) | ((uintptr_t)dout) | ((uintptr_t)d_o)) & 15) == 0;
  hipLaunchKernelGGL(scale_add_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gamma, o, dout, d_o,
                     (double*)scratch, (long)n, vec);
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)scratch, blocks,
                     dgamma);
  return icg_check_launch();
}

// ---------------------------------------------------------------- column sums (bias gradients), any C
// Every thread walks the flat [rows*C] array with a stride that is a multiple of C, so it always sees one
// channel; the per-block combine is a deterministic ordered sum.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long n, int C,
                                                             float* __restrict__ part) {
  __shared__ float buf[256];
  const long T = (long)gridDim.x * 256;  // host guarantees T % C == 0
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  for (long i = g; i < n; i += T) s += x[i];
  buf[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    // channel handled by this thread = threadIdx.x; members are threads t with (block*256 + t) % C == c
    const int c = threadIdx.x;
    const int first = (int)(((long)c - ((long)blockIdx.x * 256) % C + C) % C);
    float acc = 0.f;
    for (int t = first; t < 256; t += C) acc += buf[t];
    part[(long)blockIdx.x * C + c] = acc;
  }
}
// 16-byte version (C % 4 == 0): a thread always sees the same channel quad; 4 independent accumulation chains per thread
// keep enough loads in flight to stream at HBM rate; order of every sum is fixed (deterministic).
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float4* __restrict__ x, long n4, int C4,
                                                              float* __restrict__ part) {
  __shared__ float4 buf[256];
  const long T = (long)gridDim.x * 256;  // host guarantees T % C4 == 0
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  long i = g;
  for (; i + 3 * T < n4; i += 4 * T) {
    const float4 v0 = x[i], v1 = x[i + T], v2 = x[i + 2 * T], v3 = x[i + 3 * T];
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; i < n4; i += T) {
    const float4 v0 = x[i];
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
  }
  buf[threadIdx.x] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                 (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (threadIdx.x < C4) {
    const int q = threadIdx.x;           // channel quad; members are threads t with (block*256 + t) % C4 == q
    const int first = (int)(((long)q - ((long)blockIdx.x * 256) % C4 + C4) % C4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = first; t < 256; t += C4) {
      const float4 v = buf[t];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(part + (long)blockIdx.x * C4 * 4)[q] = acc;
  }
}

static int colsum4_blocks(int64_t rows, int C) {
  const int C4 = C / 4;
  long want = icg_cdiv(rows * C4, 256 * 8);
  if (want > 2048) want = 2048;
  if (want < 1) want = 1;
  int a = 256, b = C4;
  while (b) { int t = a % b; a = b; b = t; }
  const int q = C4 / a;
  want = icg_cdiv(want, q) * q;
  return (int)want;
}

// out[c] = sum_k part[k][c]; block = 32 channels x 32 slices of k (deterministic order)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nblocks, int C,
                                                            float* __restrict__ out) {
  __shared__ double red[32][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < C)
    for (int k = sl; k < nblocks; k += 32) s += (double)part[(long)k * C + c];
  red[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 32; ++k) s += red[k][cl];
    out[c] = (float)s;
  }
}

static int colsum_blocks(int64_t rows, int C) {
  long n = rows * C;
  long want = icg_cdiv(n, 256 * 16);
  if (want > 1024) want = 1024;
  if (want < 1) want = 1;
  // T = 256*blocks must be a multiple of C: round blocks up to a multiple of C / gcd(256, C)
  int a = 256, b = C;
  while (b) { int t = a % b; a = b; b = t; }
  const int q = C / a;
  want = icg_cdiv(want, q) * q;
  return (int)want;
}

// wide case (C > 256, few rows): one thread per column, row chunks over blockIdx.y
__global__ __launch_bounds__(256) void colsum_wide_partial_kernel(const float* __restrict__ x, long rows, int C,
                                                                  long rows_per_chunk, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  for (long r = r0; r < r1; ++r) s += x[r * C + c];
  part[(long)blockIdx.y * C + c] = s;
}

static int colsum_wide_chunks(int64_t rows) {
  long ch = icg_cdiv(rows, 8);
  if (ch > 256) ch = 256;
  if (ch < 1) ch = 1;
  return (int)ch;
}

extern "C" size_t icg_colsum_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  if (C > 256) return (size_t)colsum_wide_chunks(rows) * C * sizeof(float);
  const int b4 = (C % 4 == 0) ? colsum4_blocks(rows, C) : 0, b1 = colsum_blocks(rows, C);
  return (size_t)(b4 > b1 ? b4 : b1) * C * sizeof(float);
}

extern "C" int icg_colsum(const float* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes,
                          void* stream) {
  ICG_REQUIRE(x && out && workspace && rows > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  if (C > 256) {
    const int ch = colsum_wide_chunks(rows);
    if (workspace_bytes < (size_t)ch * C * sizeof(float)) return ICG_ERR_WORKSPACE;
    const long rpc = icg_cdiv(rows, ch);
    const int chunks = (int)icg_cdiv(rows, rpc);
    hipLaunchKernelGGL(colsum_wide_partial_kernel, dim3((unsigned)icg_cdiv(C, 256), chunks), dim3(256), 0, st, x,
                       (long)rows, C, rpc, (float*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(1024), 0, st,
                       (const float*)workspace, chunks, C, out);
    return icg_check_launch();
  }
  if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
    const int blocks = colsum4_blocks(rows, C);
    if (workspace_bytes < (size_t)blocks * C * sizeof(float)) return ICG_ERR_WORKSPACE;
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)x, (long)rows * (C / 4),
                       C / 4, (float*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(1024), 0, st, (const float*)workspace,
                       blocks, C, out);
    return icg_check_launch();
  }
  const int blocks = colsum_blocks(rows, C);
  if (workspace_bytes < (size_t)blocks * C * sizeof(float)) return ICG_ERR_WORKSPACE;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), 0, st, x, (long)rows * C, C, (float*)workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(1024), 0, st, (const float*)workspace,
                     blocks, C, out);
  return icg_check_launch();
}
