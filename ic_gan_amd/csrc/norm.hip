// BatchNorm / class+instance-conditional BatchNorm (ccbn) statistics and backward for NHWC fp32.
//
// Replaces F.batch_norm + `out * gain + bias` of ccbn.forward (BigGAN_PyTorch/layers.py:398-437), bn.forward
// (layers.py:485-503) and their autograd backward; the normalise/affine/ReLU *apply* itself lives in the
// prologue of the following convolution (gemm_conv.hip), so the forward costs ONE read of x here.
// The cross-replica variant (sync_batchnorm/batchnorm.py:61-193: sum / sum-of-squares exchange) plugs in
// between icg_bn_reduce_partials and icg_bn_finalize (forward) and around icg_bn_bwd_channel_sums (backward):
// the caller all-reduces the small double[2][C] buffers over RCCL.
//
// All kernels are HBM-bound column reductions over [rows][C] with C contiguous: a thread owns one float4
// channel quad and walks rows, so every wave issues full 16-byte-per-lane coalesced loads; partial sums are
// fp32 with short serial chains (shifted by the running mean against cancellation) and are combined in fp64.
#include "icg_common.h"

struct ColPlan {
  int cv;        // float4 columns (C/4)
  int cc;        // float4 columns per block
  int ncol;      // column chunks
  int ny;        // row lanes per block (256 / cc)
  int nchunks;   // row chunks
  long rows_per_chunk;
};

static ColPlan col_plan(long rows, int C, long max_blocks) {
  ColPlan pl;
  pl.cv = C / 4;
  if (pl.cv <= 256) {
    pl.cc = pl.cv;
    pl.ncol = 1;
  } else {
    pl.ncol = (int)icg_cdiv(pl.cv, 256);
    pl.cc = (int)icg_cdiv(pl.cv, pl.ncol);
  }
  pl.ny = 256 / pl.cc;
  if (pl.ny < 1) pl.ny = 1;
  long want = icg_cdiv(rows, (long)pl.ny * 8);
  long cap = max_blocks / pl.ncol;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  pl.rows_per_chunk = icg_cdiv(rows, want);
  pl.nchunks = (int)icg_cdiv(rows, pl.rows_per_chunk);
  return pl;
}

__device__ __forceinline__ float4 ld4n(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------- forward statistics
// partial[chunk][2][C]: shifted sum and shifted sum of squares
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ kshift,
                                                         long rows, int C, ColPlan pl, float* __restrict__ partial) {
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x;
  const int tx = tid % pl.cc, ty = tid / pl.cc;
  const int col4 = blockIdx.y * pl.cc + tx;
  const bool active = (ty < pl.ny) && (col4 < pl.cv);
  const long r0 = (long)blockIdx.x * pl.rows_per_chunk;
  const long r1 = min(rows, r0 + pl.rows_per_chunk);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  if (active) {
    float4 k = make_float4(0, 0, 0, 0);
    if (kshift) k = ld4n(kshift + 4 * col4);
    for (long r = r0 + ty; r < r1; r += pl.ny) {
      float4 v = ld4n(x + r * C + 4 * col4);
      v.x -= k.x; v.y -= k.y; v.z -= k.z; v.w -= k.w;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y);
      s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
    }
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (active && ty == 0) {
    for (int y = 1; y < pl.ny; ++y) {
      float4 a = red[0][y * pl.cc + tx], b = red[1][y * pl.cc + tx];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    float* o = partial + (long)blockIdx.x * 2 * C;
    *reinterpret_cast<float4*>(o + 4 * col4) = s1;
    *reinterpret_cast<float4*>(o + C + 4 * col4) = s2;
  }
}

// sums[2][C] (double) = sum over chunks of partial[chunk][2][C]; block = 32 channels x 32 chunk slices (8 slices until round 4: with
// C / 32 = 3 ... 6 workgroups for the 96- / 192-channel layers each thread walked 128 of the 1024 chunks serially: 35 us per launch,
// 26 launches per step)
__global__ __launch_bounds__(1024) void bn_reduce_partials_kernel(const float* __restrict__ partial, int nchunks, int C,
                                                                  double* __restrict__ sums) {
  constexpr int SL = 32;
  __shared__ double red[2][SL][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    for (int k = sl; k < nchunks; k += SL) {
      a += (double)partial[(long)k * 2 * C + c];
      b += (double)partial[(long)k * 2 * C + C + c];
    }
  }
  red[0][sl][cl] = a;
  red[1][sl][cl] = b;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < SL; ++k) {
      a += red[0][k][cl];
      b += red[1][k][cl];
    }
    sums[c] = a;
    sums[C + c] = b;
  }
}

// block = 32 channels x 8 row slices: the per-sample rows of scale / shift (gb_rows = B for conditional BN) are written by 8 threads
// per channel instead of one walking all B rows (29 us per launch at B = 64, 26 launches per step)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums, const float* kshift, double count,
                                   float* running_mean, float* running_var, float momentum, float eps, int training,
                                   const float* __restrict__ gain, const float* __restrict__ bias, int gb_rows,
                                   float gain_offset, int C, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                   float* __restrict__ scale, float* __restrict__ shift) {
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < C;
  float mean = 0.f, invstd = 0.f;
  double mu = 0.0, var = 0.0;
  if (live) {
    if (training) {
      const double k = kshift ? (double)kshift[c] : 0.0;   // read (by every slice) before the running mean is overwritten below
      if (count <= 0.0) count = sums[2 * C];                // packed cross-replica payload [sum, sumsq, n] (icg_bn_sync_pack)
      const double m1 = sums[c] / count;
      var = sums[C + c] / count - m1 * m1;
      if (var < 0.0) var = 0.0;
      mu = k + m1;
      mean = (float)mu;
      invstd = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      mean = running_mean[c];
      invstd = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
    }
  }
  __syncthreads();                                          // kshift may BE running_mean
  if (live && sl == 0) {
    if (training && running_mean) {
      const double unb = (count > 1.0) ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
    mean_o[c] = mean;
    invstd_o[c] = invstd;
  }
  if (!live) return;
  for (int b = sl; b < gb_rows; b += 8) {
    const float g = gain_offset + (gain ? gain[(long)b * C + c] : 0.f);
    const float be = bias ? bias[(long)b * C + c] : 0.f;
    const float sc = invstd * g;
    scale[(long)b * C + c] = sc;
    shift[(long)b * C + c] = be - mean * sc;
  }
}

// icg_bn_reduce_partials + icg_bn_finalize of the training-mode forward in ONE launch (26 of each per cfg3 step otherwise): a block
// owns 32 channels as both kernels do; its 32 slices of 32 lanes sum the partial rows in the order of bn_reduce_partials_kernel,
// the 8 first slices then run bn_finalize_kernel's arithmetic on the block's own sums -- bit-identical to the two-kernel path.
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ partial, int nchunks, const float* kshift,
                                                                  double count, float* running_mean, float* running_var,
                                                                  float momentum, float eps, const float* __restrict__ gain,
                                                                  const float* __restrict__ bias, int gb_rows, float gain_offset,
                                                                  int C, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                                  float* __restrict__ scale, float* __restrict__ shift) {
  constexpr int SL = 32;
  __shared__ double red[2][SL][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < C;
  double a = 0.0, b = 0.0;
  if (live) {
    for (int k = sl; k < nchunks; k += SL) {
      a += (double)partial[(long)k * 2 * C + c];
      b += (double)partial[(long)k * 2 * C + C + c];
    }
  }
  red[0][sl][cl] = a;
  red[1][sl][cl] = b;
  __syncthreads();
  if (sl == 0) {
    for (int k = 1; k < SL; ++k) {
      a += red[0][k][cl];
      b += red[1][k][cl];
    }
    red[0][0][cl] = a;
    red[1][0][cl] = b;
  }
  __syncthreads();
  const bool fin = sl < 8;                                  // the 8 x 32 threads bn_finalize_kernel runs
  float mean = 0.f, invstd = 0.f;
  double mu = 0.0, var = 0.0;
  if (live && fin) {
    const double k = kshift ? (double)kshift[c] : 0.0;      // read by every slice before the running mean is overwritten below
    const double m1 = red[0][0][cl] / count;
    var = red[1][0][cl] / count - m1 * m1;
    if (var < 0.0) var = 0.0;
    mu = k + m1;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();                                          // kshift may BE running_mean
  if (live && sl == 0) {
    if (running_mean) {
      const double unb = (count > 1.0) ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
    mean_o[c] = mean;
    invstd_o[c] = invstd;
  }
  if (!live || !fin) return;
  for (int r = sl; r < gb_rows; r += 8) {
    const float g = gain_offset + (gain ? gain[(long)r * C + c] : 0.f);
    const float be = bias ? bias[(long)r * C + c] : 0.f;
    const float sc = invstd * g;
    scale[(long)r * C + c] = sc;
    shift[(long)r * C + c] = be - mean * sc;
  }
}

extern "C" size_t icg_bn_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || (C % 4) != 0) return 0;
  ColPlan pl = col_plan(rows, C, 1024);
  return (size_t)pl.nchunks * 2 * C * sizeof(float);
}

extern "C" int icg_bn_partial_stats(const float* x, const float* shift_k, int64_t rows, int C, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && workspace && rows > 0 && C > 0 && (C % 4) == 0);
  ColPlan pl = col_plan(rows, C, 1024);
  if (workspace_bytes < (size_t)pl.nchunks * 2 * C * sizeof(float)) return ICG_ERR_WORKSPACE;
  hipLaunchKernelGGL(bn_partial_kernel, dim3(pl.nchunks, pl.ncol), dim3(256), 0, (hipStream_t)stream, x, shift_k,
                     (long)rows, C, pl, (float*)workspace);
  return icg_check_launch();
}

extern "C" int icg_bn_reduce_partials(const void* workspace, int64_t rows, int C, double* sums, void* stream) {
  ICG_REQUIRE(workspace && sums && rows > 0 && C > 0 && (C % 4) == 0);
  ColPlan pl = col_plan(rows, C, 1024);
  hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, pl.nchunks, C, sums);
  return icg_check_launch();
}

extern "C" int icg_bn_finalize(const double* sums, const float* shift_k, double count, float* running_mean,
                               float* running_var, float momentum, float eps, int training, const float* gain,
                               const float* bias, int gb_rows, float gain_offset, int C, float* mean, float* invstd,
                               float* scale, float* shift, void* stream) {
  ICG_REQUIRE(C > 0 && mean && invstd && scale && shift && gb_rows >= 1);
  if (training) ICG_REQUIRE(sums != nullptr);          // count <= 0: the element count is sums[2*C] (device side)
  else ICG_REQUIRE(running_mean && running_var);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, sums,
                     shift_k, count, running_mean, running_var, momentum, eps, training, gain, bias, gb_rows,
                     gain_offset, C, mean, invstd, scale, shift);
  return icg_check_launch();
}

extern "C" int icg_bn_reduce_finalize(const void* workspace, int64_t rows, int C, const float* shift_k, float* running_mean,
                                      float* running_var, float momentum, float eps, const float* gain, const float* bias,
                                      int gb_rows, float gain_offset, float* mean, float* invstd, float* scale, float* shift,
                                      void* stream) {
  ICG_REQUIRE(workspace && rows > 0 && C > 0 && (C % 4) == 0 && mean && invstd && scale && shift && gb_rows >= 1);
  ColPlan pl = col_plan(rows, C, 1024);
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, pl.nchunks, shift_k, (double)rows, running_mean, running_var, momentum, eps, gain, bias,
                     gb_rows, gain_offset, C, mean, invstd, scale, shift);
  return icg_check_launch();
}

// Cross-replica BN payload: the per-replica sums are taken about each replica's own shift k (its running mean), so they are
// moved to the common origin 0 in fp64 before the all-reduce, and the element count travels with them:
//   payload = [ sum x (C) | sum x^2 (C) | n ],   sum x = S1 + n k,  sum x^2 = S2 + 2 k S1 + n k^2
// (fp64: the cancellation in var = E[x^2] - E[x]^2 costs ~1e-16 * mean^2/var, far below fp32 resolution)
__global__ void bn_sync_pack_kernel(const double* __restrict__ sums, const float* __restrict__ kshift, double n, int C,
                                    double* __restrict__ payload) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) payload[2 * C] = n;
  if (c >= C) return;
  const double k = kshift ? (double)kshift[c] : 0.0;
  const double s1 = sums[c], s2 = sums[C + c];
  payload[c] = s1 + n * k;
  payload[C + c] = s2 + 2.0 * k * s1 + n * k * k;
}

extern "C" int icg_bn_sync_pack(const double* sums, const float* shift_k, double local_count, int C, double* payload,
                                void* stream) {
  ICG_REQUIRE(sums && payload && C > 0 && local_count > 0);
  hipLaunchKernelGGL(bn_sync_pack_kernel, dim3((unsigned)icg_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sums,
                     shift_k, local_count, C, payload);
  return icg_check_launch();
}

// ---------------------------------------------------------------- backward
// dy = (sum over the 2x2 window when `up`) da * mask,  mask = (x*scale+shift > 0) when relu.
__device__ __forceinline__ float4 bwd_dy(const float* __restrict__ da, long b, int hs, int ws, int Hs, int Ws, int C,
                                         int c, int up, float4 xv, float4 sc, float4 sh, int affine, int relu) {
  float4 d;
  if (up) {
    const int H = 2 * Hs, W = 2 * Ws;
    const float* p00 = da + (((long)b * H + 2 * hs) * W + 2 * ws) * C + c;
    const float4 a = ld4n(p00), b4 = ld4n(p00 + C), c4 = ld4n(p00 + (long)W * C), d4 = ld4n(p00 + (long)W * C + C);
    d.x = (a.x + b4.x) + (c4.x + d4.x);
    d.y = (a.y + b4.y) + (c4.y + d4.y);
    d.z = (a.z + b4.z) + (c4.z + d4.z);
    d.w = (a.w + b4.w) + (c4.w + d4.w);
  } else {
    d = ld4n(da + (((long)b * Hs + hs) * Ws + ws) * C + c);
  }
  if (relu) {
    float4 y = xv;
    if (affine) {
      y.x = fmaf(xv.x, sc.x, sh.x); y.y = fmaf(xv.y, sc.y, sh.y);
      y.z = fmaf(xv.z, sc.z, sh.z); y.w = fmaf(xv.w, sc.w, sh.w);
    }
    d.x = y.x > 0.f ? d.x : 0.f;
    d.y = y.y > 0.f ? d.y : 0.f;
    d.z = y.z > 0.f ? d.z : 0.f;
    d.w = y.w > 0.f ? d.w : 0.f;
  }
  return d;
}

struct BwdPlan {
  int cv, cc, ncol, ny, nchunks_img;
  int rows_per_chunk;
};

static BwdPlan bwd_plan(int B, int HW, int C) {
  BwdPlan pl;
  pl.cv = C / 4;
  if (pl.cv <= 256) {
    pl.cc = pl.cv;
    pl.ncol = 1;
  } else {
    pl.ncol = (int)icg_cdiv(pl.cv, 256);
    pl.cc = (int)icg_cdiv(pl.cv, pl.ncol);
  }
  pl.ny = 256 / pl.cc;
  if (pl.ny < 1) pl.ny = 1;
  long want = icg_cdiv(HW, (long)pl.ny * 8);
  long cap = 2048 / ((long)B * pl.ncol);
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  pl.rows_per_chunk = (int)icg_cdiv(HW, want);
  pl.nchunks_img = (int)icg_cdiv(HW, pl.rows_per_chunk);
  return pl;
}

// partial[b][chunk][2][C]:  sum dy  and  sum dy*(x - mean[c])
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, long ss_bstride,
                                                             const float* __restrict__ mean, int Hs, int Ws, int C,
                                                             int up, int affine, int relu, BwdPlan pl,
                                                             float* __restrict__ partial) {
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x;
  const int tx = tid % pl.cc, ty = tid / pl.cc;
  const int col4 = blockIdx.y * pl.cc + tx;
  const bool active = (ty < pl.ny) && (col4 < pl.cv);
  const int b = blockIdx.z;
  const int HW = Hs * Ws;
  const int r0 = blockIdx.x * pl.rows_per_chunk;
  const int r1 = min(HW, r0 + pl.rows_per_chunk);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  if (active) {
    const int c = 4 * col4;
    float4 sc = make_float4(0, 0, 0, 0), sh = sc, mu = sc;
    if (affine) {
      sc = ld4n(scale + (long)b * ss_bstride + c);
      sh = ld4n(shift + (long)b * ss_bstride + c);
    }
    if (mean) mu = ld4n(mean + c);
    for (int r = r0 + ty; r < r1; r += pl.ny) {
      const int hs = r / Ws, ws = r - hs * Ws;
      const float4 xv = ld4n(x + ((long)b * HW + r) * C + c);
      const float4 d = bwd_dy(da, b, hs, ws, Hs, Ws, C, c, up, xv, sc, sh, affine, relu);
      s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
      s2.x = fmaf(d.x, xv.x - mu.x, s2.x); s2.y = fmaf(d.y, xv.y - mu.y, s2.y);
      s2.z = fmaf(d.z, xv.z - mu.z, s2.z); s2.w = fmaf(d.w, xv.w - mu.w, s2.w);
    }
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (active && ty == 0) {
    for (int y = 1; y < pl.ny; ++y) {
      float4 a = red[0][y * pl.cc + tx], b4 = red[1][y * pl.cc + tx];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b4.x; s2.y += b4.y; s2.z += b4.z; s2.w += b4.w;
    }
    float* o = partial + ((long)b * pl.nchunks_img + blockIdx.x) * 2 * C;
    *reinterpret_cast<float4*>(o + 4 * col4) = s1;
    *reinterpret_cast<float4*>(o + C + 4 * col4) = s2;
  }
}

// sum_dy[b][c], sum_dyx[b][c] = sum over chunks
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ partial, int nchunks, int C, float* __restrict__ sum_dy,
                                     float* __restrict__ sum_dyx) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C) return;
  double a = 0.0, d = 0.0;
  const float* p = partial + (long)b * nchunks * 2 * C;
  for (int k = 0; k < nchunks; ++k) {
    a += (double)p[(long)k * 2 * C + c];
    d += (double)p[(long)k * 2 * C + C + c];
  }
  sum_dy[(long)b * C + c] = (float)a;
  sum_dyx[(long)b * C + c] = (float)d;
}

// chan_sums[0][c] = sum_b g[b][c]*Sd[b][c];  chan_sums[1][c] = sum_b g[b][c]*invstd[c]*Sxc[b][c]
// (this kernel and the next: block = 32 channels x 8 batch slices folded through LDS, instead of one thread per channel walking
// the whole batch in fp64 -- 43 / 50 us per launch at B = 64, 13 launches each per step)
__global__ __launch_bounds__(256) void bn_bwd_chan_kernel(const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx,
                                   const float* __restrict__ gain, int gb_rows, float gain_offset,
                                   const float* __restrict__ invstd, int B, int C, double* __restrict__ chan) {
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, d = 0.0;
  if (c < C) {
    const double is = (double)invstd[c];
    for (int b = sl; b < B; b += 8) {
      const int gb = (gb_rows == 1) ? 0 : b;
      const double g = (double)gain_offset + (gain ? (double)gain[(long)gb * C + c] : 0.0);
      a += g * (double)sum_dy[(long)b * C + c];
      d += g * is * (double)sum_dyx[(long)b * C + c];
    }
  }
  red[0][sl][cl] = a;
  red[1][sl][cl] = d;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 8; ++k) {
      a += red[0][k][cl];
      d += red[1][k][cl];
    }
    chan[c] = a;
    chan[C + c] = d;
  }
}

// dgain[gb][c] = invstd*Sxc, dbias[gb][c] = Sd (summed over b when gb_rows == 1);
// coefA[c] = invstd*mean(dxhat), coefB[c] = invstd^2*mean(dxhat*xhat)  (0 when !batch_stats)
__global__ __launch_bounds__(256) void bn_bwd_coefs_kernel(const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx,
                                    const double* __restrict__ chan, const float* __restrict__ invstd, double count,
                                    int batch_stats, int gb_rows, int B, int C, float* __restrict__ dgain,
                                    float* __restrict__ dbias, float* __restrict__ coefA, float* __restrict__ coefB) {
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < C;
  const float is = live ? invstd[c] : 0.f;
  double a = 0.0, d = 0.0;
  if (live) {
    if (gb_rows == 1) {
      for (int b = sl; b < B; b += 8) {
        a += (double)sum_dy[(long)b * C + c];
        d += (double)sum_dyx[(long)b * C + c];
      }
    } else {
      for (int b = sl; b < B; b += 8) {
        if (dgain) dgain[(long)b * C + c] = is * sum_dyx[(long)b * C + c];
        if (dbias) dbias[(long)b * C + c] = sum_dy[(long)b * C + c];
      }
    }
  }
  red[0][sl][cl] = a;
  red[1][sl][cl] = d;
  __syncthreads();
  if (sl != 0 || !live) return;
  if (gb_rows == 1) {
    for (int k = 1; k < 8; ++k) {
      a += red[0][k][cl];
      d += red[1][k][cl];
    }
    if (dgain) dgain[c] = (float)((double)is * d);
    if (dbias) dbias[c] = (float)a;
  }
  float ca = 0.f, cb = 0.f;
  if (batch_stats) {
    if (count <= 0.0) count = chan[2 * C];               // global element count kept on the device (cross-replica BN)
    ca = (float)((double)is * chan[c] / count);
    cb = (float)((double)is * (double)is * chan[C + c] / count);
  }
  coefA[c] = ca;
  coefB[c] = cb;
}

// dx = dy*scale[b][c] - coefA[c] - coefB[c]*(x - mean[c])
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, long ss_bstride,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ coefA,
                                                           const float* __restrict__ coefB, int B, int Hs, int Ws, int C,
                                                           int up, int affine, int relu, float* __restrict__ dx) {
  const int cv = C / 4;
  const long total = (long)B * Hs * Ws * cv;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = 4 * (int)(i % cv);
    const long pix = i / cv;
    const int ws = (int)(pix % Ws);
    const long t = pix / Ws;
    const int hs = (int)(t % Hs);
    const long b = t / Hs;
    const float4 xv = ld4n(x + pix * C + c);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0, 0, 0, 0);
    if (affine) {
      sc = ld4n(scale + b * ss_bstride + c);
      sh = ld4n(shift + b * ss_bstride + c);
    }
    const float4 d = bwd_dy(da, b, hs, ws, Hs, Ws, C, c, up, xv, sc, sh, affine, relu);
    float4 o;
    o.x = d.x * sc.x; o.y = d.y * sc.y; o.z = d.z * sc.z; o.w = d.w * sc.w;
    if (coefA) {
      const float4 ca = ld4n(coefA + c), cb = ld4n(coefB + c), mu = ld4n(mean + c);
      o.x -= ca.x + cb.x * (xv.x - mu.x);
      o.y -= ca.y + cb.y * (xv.y - mu.y);
      o.z -= ca.z + cb.z * (xv.z - mu.z);
      o.w -= ca.w + cb.w * (xv.w - mu.w);
    }
    *reinterpret_cast<float4*>(dx + pix * C + c) = o;
  }
}

extern "C" size_t icg_bn_bwd_workspace_bytes(int B, int Hs, int Ws, int C) {
  if (B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || (C % 4) != 0) return 0;
  BwdPlan pl = bwd_plan(B, Hs * Ws, C);
  return (size_t)B * pl.nchunks_img * 2 * C * sizeof(float);
}

extern "C" int icg_bn_bwd_reduce(const float* x, const float* da, const float* scale, const float* shift,
                                 int64_t ss_bstride, const float* mean, int B, int Hs, int Ws, int C, unsigned flags,
                                 void* workspace, size_t workspace_bytes, float* sum_dy, float* sum_dyx,
                                 void* stream) {
  ICG_REQUIRE(x && da && workspace && sum_dy && sum_dyx && B > 0 && Hs > 0 && Ws > 0 && C > 0 && (C % 4) == 0);
  ICG_REQUIRE(B <= 65535);
  const int affine = (flags & ICG_PRE_AFFINE) ? 1 : 0, relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  const int up = (flags & ICG_UPSAMPLE2X) ? 1 : 0;
  if (affine) ICG_REQUIRE(scale && shift && (ss_bstride % 4) == 0);
  BwdPlan pl = bwd_plan(B, Hs * Ws, C);
  if (workspace_bytes < (size_t)B * pl.nchunks_img * 2 * C * sizeof(float)) return ICG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(pl.nchunks_img, pl.ncol, B), dim3(256), 0, st, x, da, scale, shift,
                     (long)ss_bstride, mean, Hs, Ws, C, up, affine, relu, pl, (float*)workspace);
  int rc = icg_check_launch();
  if (rc != ICG_OK) return rc;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)icg_cdiv(C, 128), B), dim3(128), 0, st,
                     (const float*)workspace, pl.nchunks_img, C, sum_dy, sum_dyx);
  return icg_check_launch();
}

extern "C" int icg_bn_bwd_channel_sums(const float* sum_dy, const float* sum_dyx, const float* gain, int gb_rows,
                                       float gain_offset, const float* invstd, int B, int C, double* chan_sums,
                                       void* stream) {
  ICG_REQUIRE(sum_dy && sum_dyx && invstd && chan_sums && B > 0 && C > 0 && gb_rows >= 1);
  hipLaunchKernelGGL(bn_bwd_chan_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, sum_dy,
                     sum_dyx, gain, gb_rows, gain_offset, invstd, B, C, chan_sums);
  return icg_check_launch();
}

extern "C" int icg_bn_bwd_coefs(const float* sum_dy, const float* sum_dyx, const double* chan_sums,
                                const float* invstd, double count, int batch_stats, int gb_rows, int B, int C,
                                float* dgain, float* dbias, float* coefA, float* coefB, void* stream) {
  ICG_REQUIRE(sum_dy && sum_dyx && invstd && coefA && coefB && B > 0 && C > 0 && gb_rows >= 1);
  if (batch_stats) ICG_REQUIRE(chan_sums != nullptr);
  hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3((unsigned)icg_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, sum_dy,
                     sum_dyx, chan_sums, invstd, count, batch_stats, gb_rows, B, C, dgain, dbias, coefA, coefB);
  return icg_check_launch();
}

extern "C" int icg_bn_bwd_apply(const float* x, const float* da, const float* scale, const float* shift,
                                int64_t ss_bstride, const float* mean, const float* coefA, const float* coefB, int B,
                                int Hs, int Ws, int C, unsigned flags, float* dx, void* stream) {
  ICG_REQUIRE(x && da && dx && B > 0 && Hs > 0 && Ws > 0 && C > 0 && (C % 4) == 0);
  const int affine = (flags & ICG_PRE_AFFINE) ? 1 : 0, relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  const int up = (flags & ICG_UPSAMPLE2X) ? 1 : 0;
  if (affine) ICG_REQUIRE(scale && shift && (ss_bstride % 4) == 0);
  if (coefA) ICG_REQUIRE(coefB && mean);
  const long total = (long)B * Hs * Ws * (C / 4);
  long blocks = icg_cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, da, scale,
                     shift, (long)ss_bstride, mean, coefA, coefB, B, Hs, Ws, C, up, affine, relu, dx);
  return icg_check_launch();
}

// ---------------------------------------------------------------- stand-alone apply (API parity: ccbn / bn called
// outside a fused block).  y = relu?(x*scale[b][c] + shift[b][c])
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, long ss_bstride, long HW, int C,
                                                       long total4, int relu, float* __restrict__ y) {
  const int cv = C / 4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int c = 4 * (int)(i % cv);
    const long pix = i / cv;
    const long b = pix / HW;
    const float4 v = ld4n(x + pix * C + c);
    const float4 sc = ld4n(scale + b * ss_bstride + c), sh = ld4n(shift + b * ss_bstride + c);
    float4 o;
    o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + pix * C + c) = o;
  }
}

extern "C" int icg_bn_apply(const float* x, const float* scale, const float* shift, int64_t ss_bstride, int B,
                            int64_t HW, int C, unsigned flags, float* y, void* stream) {
  ICG_REQUIRE(x && scale && shift && y && B > 0 && HW > 0 && C > 0 && (C % 4) == 0 && (ss_bstride % 4) == 0);
  const long total4 = (long)B * HW * (C / 4);
  long blocks = icg_cdiv(total4, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                     (long)ss_bstride, (long)HW, C, total4, (flags & ICG_PRE_RELU) ? 1 : 0, y);
  return icg_check_launch();
}
