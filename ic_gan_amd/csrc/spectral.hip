// Spectral normalisation: one power-iteration step + W/sigma, and its backward.
// Replaces power_iteration / SN.W_ (BigGAN_PyTorch/layers.py:39-61, 98-112) and the autograd graph
//   sigma = (v W^T) u'^T  (u', v constant)  ->  dW = (dW_ - <dW_, W_> u'^T v) / sigma.
// HBM-bound GEMV-shaped work: W is read three times (u W, W v^T, W/sigma) with coalesced row-major accesses;
// the normalised weight is emitted directly in the two layouts the MFMA kernels consume:
//   OHWI  [Cout][R][R][Cin]            (fprop:  B operand rows are K-contiguous)
//   dgrad [Cin][R][R][Cout], taps flipped (data gradient = the same fprop kernel on dy)
#include "icg_common.h"

// part[yc][j] = sum_{i in row chunk yc} u[i] * w[i][j]
// HBM-bound GEMV: a thread owns 4 consecutive columns (16-byte loads, a wavefront reads 1 KiB of a row per instruction) and
// walks its row chunk four rows at a time (four independent loads in flight); scalar columns when cols % 4 != 0 (Cin = 3).
#define SN_MAX_CHUNKS 64      // row chunks per layer: enough blocks to fill the chip on the 1536 x 13824 layers
__device__ __forceinline__ void sn_uw_partial_body(int bx, int by, int gx, const float* __restrict__ w, const float* __restrict__ u,
                                                            int rows, int cols, int rows_per_chunk,
                                                            float* __restrict__ part) {
  const int yc = by;
  const int i0 = yc * rows_per_chunk, i1 = min(rows, i0 + rows_per_chunk);
  if ((cols & 3) == 0) {
    const int j = (bx * 256 + threadIdx.x) * 4;
    if (j >= cols) return;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int i = i0;
    for (; i + 1 < i1; i += 2) {
      const float4 a = *reinterpret_cast<const float4*>(w + (long)i * cols + j);
      const float4 b = *reinterpret_cast<const float4*>(w + (long)(i + 1) * cols + j);
      const float ua = u[i], ub = u[i + 1];
      s0.x = fmaf(ua, a.x, s0.x); s0.y = fmaf(ua, a.y, s0.y); s0.z = fmaf(ua, a.z, s0.z); s0.w = fmaf(ua, a.w, s0.w);
      s1.x = fmaf(ub, b.x, s1.x); s1.y = fmaf(ub, b.y, s1.y); s1.z = fmaf(ub, b.z, s1.z); s1.w = fmaf(ub, b.w, s1.w);
    }
    if (i < i1) {
      const float4 a = *reinterpret_cast<const float4*>(w + (long)i * cols + j);
      const float ua = u[i];
      s0.x = fmaf(ua, a.x, s0.x); s0.y = fmaf(ua, a.y, s0.y); s0.z = fmaf(ua, a.z, s0.z); s0.w = fmaf(ua, a.w, s0.w);
    }
    *reinterpret_cast<float4*>(part + (long)yc * cols + j) = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    return;
  }
  for (int q = 0; q < 4; ++q) {            // the same block-to-column map (4 columns per thread slot), one at a time
    const int j = (bx * 256 + threadIdx.x) * 4 + q;
    if (j >= cols) return;
    float s = 0.f;
    for (int i = i0; i < i1; ++i) s = fmaf(u[i], w[(long)i * cols + j], s);
    part[(long)yc * cols + j] = s;
  }
}
__global__ __launch_bounds__(256) void sn_uw_partial_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                            int rows, int cols, int rows_per_chunk,
                                                            float* __restrict__ part) {
  sn_uw_partial_body(blockIdx.x, blockIdx.y, gridDim.x, w, u, rows, cols, rows_per_chunk, part);
}

// scratch layout of one layer (floats unless noted): part[SN_MAX_CHUNKS][cols] | t[cols] | s[rows] | (8-byte aligned) ssq[SN_VBLOCKS] doubles
#define SN_VBLOCKS 64
struct SnScratch { float* part; float* t; float* s; double* ssq; };
__host__ __device__ __forceinline__ SnScratch sn_scratch(void* base, int rows, int cols) {
  SnScratch r;
  r.part = (float*)base;
  r.t = r.part + (long)SN_MAX_CHUNKS * cols;
  r.s = r.t + cols;
  const long nf = (((long)SN_MAX_CHUNKS * cols + cols + rows + 1) / 2) * 2;
  r.ssq = (double*)(r.part + nf);
  return r;
}

// t[j] = sum_yc part[yc][j] (chunk order), ssq[block] = sum over the block's columns of t^2.  Many blocks: the single-block
// form of round 1 walked 64 x 13824 partials with 1024 threads (~90 us of the 0.37 ms a 1536 x 13824 layer took).
__device__ __forceinline__ void sn_vsum_body(int bx, int gx, const float* __restrict__ part, int ychunks, int cols,
                                             float* __restrict__ t, double* __restrict__ ssq) {
  __shared__ double red[4];
  double ss = 0.0;
  for (int j = bx * 256 + threadIdx.x; j < cols; j += gx * 256) {
    float a = 0.f;
    for (int y = 0; y < ychunks; ++y) a += part[(long)y * cols + j];
    t[j] = a;
    ss += (double)a * (double)a;
  }
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) ssq[bx] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sn_vsum_kernel(const float* __restrict__ part, int ychunks, int cols,
                                                      float* __restrict__ t, double* __restrict__ ssq) {
  sn_vsum_body(blockIdx.x, gridDim.x, part, ychunks, cols, t, ssq);
}

// 1 / max(||t||, eps) from the block partials (every wavefront recomputes it: <= 64 doubles, fixed order)
__device__ __forceinline__ float sn_inv_norm(const double* __restrict__ ssq, int nblk, float eps) {
  const int lane = threadIdx.x & 63;
  double a = lane < nblk ? ssq[lane] : 0.0;
  a = wave_sum_d(a);
  return 1.0f / fmaxf((float)sqrt(a), eps);
}

// v = t / max(||t||, eps) (written by the blocks in a grid-stride sweep) and s[i] = sum_j w[i][j] v[j] = inv * sum_j w[i][j] t[j]:
// one wavefront per row, 16-byte loads (two in flight) when the row length allows
__device__ __forceinline__ void sn_wv_body(int bx, int gx, const float* __restrict__ w, const float* __restrict__ t,
                                           const double* __restrict__ ssq, int nblk, float eps, int rows, int cols,
                                           float* __restrict__ v_out, float* __restrict__ s) {
  const int lane = threadIdx.x & 63;
  const float inv = sn_inv_norm(ssq, nblk, eps);
  for (int j = bx * 256 + threadIdx.x; j < cols; j += gx * 256) v_out[j] = t[j] * inv;
  const int i = bx * 4 + (threadIdx.x >> 6);
  if (i >= rows) return;
  const float* wr = w + (long)i * cols;
  float acc = 0.f;
  if ((cols & 3) == 0) {
    float acc2 = 0.f;
    const int c4 = cols >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(wr);
    const float4* v4 = reinterpret_cast<const float4*>(t);
    int j = lane;
    for (; j + 64 < c4; j += 128) {
      const float4 a = w4[j], b = w4[j + 64], x = v4[j], y = v4[j + 64];
      acc = fmaf(a.x, x.x, acc); acc = fmaf(a.y, x.y, acc); acc = fmaf(a.z, x.z, acc); acc = fmaf(a.w, x.w, acc);
      acc2 = fmaf(b.x, y.x, acc2); acc2 = fmaf(b.y, y.y, acc2); acc2 = fmaf(b.z, y.z, acc2); acc2 = fmaf(b.w, y.w, acc2);
    }
    if (j < c4) {
      const float4 a = w4[j], x = v4[j];
      acc = fmaf(a.x, x.x, acc); acc = fmaf(a.y, x.y, acc); acc = fmaf(a.z, x.z, acc); acc = fmaf(a.w, x.w, acc);
    }
    acc += acc2;
  } else {
    for (int j = lane; j < cols; j += 64) acc = fmaf(wr[j], t[j], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) s[i] = acc * inv;
}
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ w, const float* __restrict__ t,
                                                    const double* __restrict__ ssq, int nblk, float eps, int rows, int cols,
                                                    float* __restrict__ v_out, float* __restrict__ s) {
  sn_wv_body(blockIdx.x, gridDim.x, w, t, ssq, nblk, eps, rows, cols, v_out, s);
}

// single block: u' = s / max(||s||, eps); sigma = s . u'
__device__ __forceinline__ void sn_u_body(int bx, int by, int gx, const float* __restrict__ s, int rows, float eps, int training,
                                                    float* __restrict__ u, float* __restrict__ sv,
                                                    float* __restrict__ u_out, float* __restrict__ sigma_out) {
  __shared__ double red[16];
  __shared__ float s_inv;
  double ss = 0.0;
  for (int i = threadIdx.x; i < rows; i += 1024) ss += (double)s[i] * (double)s[i];
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int k = 0; k < 16; ++k) tot += red[k];
    s_inv = 1.0f / fmaxf((float)sqrt(tot), eps);
  }
  __syncthreads();
  const float inv = s_inv;
  double dot = 0.0;
  for (int i = threadIdx.x; i < rows; i += 1024) {
    const float un = s[i] * inv;
    u_out[i] = un;
    if (training) u[i] = un;
    dot += (double)s[i] * (double)un;
  }
  __syncthreads();
  dot = wave_sum_d(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int k = 0; k < 16; ++k) tot += red[k];
    sigma_out[0] = (float)tot;
    if (training && sv) sv[0] = (float)tot;
  }
}
__global__ __launch_bounds__(1024) void sn_u_kernel(const float* __restrict__ s, int rows, float eps, int training,
                                                    float* __restrict__ u, float* __restrict__ sv,
                                                    float* __restrict__ u_out, float* __restrict__ sigma_out) {
  sn_u_body(blockIdx.x, blockIdx.y, gridDim.x, s, rows, eps, training, u, sv, u_out, sigma_out);
}

// w_ohwi[co][r][s][ci] = w[co][ci][r][s]/sigma ; w_dgrad[ci][R-1-r][R-1-s][co] = same
// Both outputs are transposes of the parameter layout: a 32(co) x 32(ci) x R*R tile goes through LDS so that the read
// (R*R*32 contiguous floats per co) and both writes (32 contiguous floats per (co, tap) / (ci, tap)) are coalesced; the
// direct form wrote w_dgrad with a stride of R*R*rows floats between neighbouring lanes.
// (RRT: R * R as a compile-time constant for the 1x1 and 3x3 layers -- the index decompositions below are divisions by it, which made
// this kernel ALU-bound with a run-time divisor: 1.2 TB/s; 0 = run-time R)
template <int RRT>
__device__ __forceinline__ void sn_scale_body_t(int bx, int gx, const float* __restrict__ w, const float* __restrict__ sigma,
                                                int rows, int Cin, int R, float* __restrict__ w_ohwi, float* __restrict__ w_dgrad,
                                                float* __restrict__ tile) {
  const int RR = RRT ? RRT : R * R;
  const float sg = sigma[0];
  const int tco = (rows + 31) >> 5, tci = (Cin + 31) >> 5;
  const int per = 32 * 32 * RR;
  for (int t = bx; t < tco * tci; t += gx) {
    const int co0 = (t / tci) << 5, ci0 = (t % tci) << 5;
    __syncthreads();
    for (int e = threadIdx.x; e < per; e += blockDim.x) {          // read: (co_l, ci_l, tap), tap fastest
      const int co_l = e / (32 * RR), rem = e - co_l * 32 * RR;
      const int ci_l = rem / RR, tap = rem - ci_l * RR;
      float v = 0.f;
      if (co0 + co_l < rows && ci0 + ci_l < Cin) v = w[((long)(co0 + co_l) * Cin + ci0) * RR + rem] / sg;
      tile[(co_l * RR + tap) * 33 + ci_l] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < per; e += blockDim.x) {          // OHWI: (co_l, tap, ci_l), ci fastest
      const int ci_l = e & 31, q = e >> 5;
      const int tap = q % RR, co_l = q / RR;
      if (co0 + co_l < rows && ci0 + ci_l < Cin)
        w_ohwi[((long)(co0 + co_l) * RR + tap) * Cin + ci0 + ci_l] = tile[(co_l * RR + tap) * 33 + ci_l];
    }
    if (w_dgrad) {
      for (int e = threadIdx.x; e < per; e += blockDim.x) {        // dgrad: (ci_l, tap, co_l), co fastest, taps flipped
        const int co_l = e & 31, q = e >> 5;
        const int tap = q % RR, ci_l = q / RR;
        if (co0 + co_l < rows && ci0 + ci_l < Cin)
          w_dgrad[((long)(ci0 + ci_l) * RR + (RR - 1 - tap)) * rows + co0 + co_l] = tile[(co_l * RR + tap) * 33 + ci_l];
      }
    }
  }
}
__device__ __forceinline__ void sn_scale_body(int bx, int by, int gx, const float* __restrict__ w, const float* __restrict__ sigma,
                                              int rows, int Cin, int R, float* __restrict__ w_ohwi, float* __restrict__ w_dgrad) {
  __shared__ float tile[32 * 9 * 33];
  if (R == 3) sn_scale_body_t<9>(bx, gx, w, sigma, rows, Cin, R, w_ohwi, w_dgrad, tile);
  else if (R == 1) sn_scale_body_t<1>(bx, gx, w, sigma, rows, Cin, R, w_ohwi, w_dgrad, tile);
  else sn_scale_body_t<0>(bx, gx, w, sigma, rows, Cin, R, w_ohwi, w_dgrad, tile);
}
__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                                       int rows, int Cin, int R, float* __restrict__ w_ohwi,
                                                       float* __restrict__ w_dgrad) {
  sn_scale_body(blockIdx.x, blockIdx.y, gridDim.x, w, sigma, rows, Cin, R, w_ohwi, w_dgrad);
}

// phase weights of the upsample-fused 3x3 conv (gemm_conv.hip, icg_conv2d_up_*), one thread per (co, ci):
//   wp[al][be][co][u][v][ci] = sum_{r in S(al,u), s in S(be,v)} w[co][ci][r][s]/sigma
//        S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
//   vd[ci][P][Q][co]         = sum_{r in T(P), s in T(Q)} w[co][ci][r][s]/sigma,   T = {2}, {1,2}, {0,1}, {0}
// phase / pooled layouts of the downsample-fused conv (icg_conv2d_down_*): with c[P] = sum_{a+r=P} w[r]
// (c0 = w0, c1 = w0+w1, c2 = w1+w2, c3 = w2) per dimension,
//   vdn[co][P][Q][ci] = 0.25 * c[P] (x) c[Q] / sigma
//   wq[al][be][ci][u][v][co] = vdn[co][Pd(al,u)][Pd(be,v)][ci],   Pd(0,0) = 3, Pd(0,1) = 1, Pd(1,0) = 2, Pd(1,1) = 0
__device__ __forceinline__ void sn_up_layouts_body(int bx, int by, int gx, const float* __restrict__ w, const float* __restrict__ sigma,
                                                            int rows, int Cin, float* __restrict__ wp,
                                                            float* __restrict__ vd, float* __restrict__ vdn,
                                                            float* __restrict__ wq) {
  const long idx = (long)bx * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * Cin) return;
  const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
  const float inv = 1.0f / sigma[0];
  float k[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) k[r][s] = w[((long)co * Cin + ci) * 9 + r * 3 + s];
  // row / column partial sums over the tap sets  {0}, {1,2}, {0,1}, {2}
  auto rowsum = [&](int set, int s) -> float {
    return set == 0 ? k[0][s] : set == 1 ? k[1][s] + k[2][s] : set == 2 ? k[0][s] + k[1][s] : k[2][s];
  };
  auto both = [&](int rset, int cset) -> float {
    const float c0 = rowsum(rset, 0), c1 = rowsum(rset, 1), c2 = rowsum(rset, 2);
    return cset == 0 ? c0 : cset == 1 ? c1 + c2 : cset == 2 ? c0 + c1 : c2;
  };
  // S(al,u): (0,0)->set0 {0}, (0,1)->set1 {1,2}, (1,0)->set2 {0,1}, (1,1)->set3 {2}   => set = 2*al + u
  if (wp) {
#pragma unroll
    for (int al = 0; al < 2; ++al)
#pragma unroll
      for (int be = 0; be < 2; ++be)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < 2; ++v)
            wp[((((long)(al * 2 + be) * rows + co) * 2 + u) * 2 + v) * Cin + ci] = both(2 * al + u, 2 * be + v) * inv;
  }
  // T(P): P=0 {2} = set3, P=1 {1,2} = set1, P=2 {0,1} = set2, P=3 {0} = set0
  if (vd) {
    const int tset[4] = {3, 1, 2, 0};
#pragma unroll
    for (int P = 0; P < 4; ++P)
#pragma unroll
      for (int Q = 0; Q < 4; ++Q)
        vd[(((long)ci * 4 + P) * 4 + Q) * rows + co] = both(tset[P], tset[Q]) * inv;
  }
  if (vdn || wq) {
    // c-sets in the numbering of rowsum(): c0 = {0} = set0, c1 = {0,1} = set2, c2 = {1,2} = set1, c3 = {2} = set3
    const int cset[4] = {0, 2, 1, 3};
    float vv[4][4];
#pragma unroll
    for (int P = 0; P < 4; ++P)
#pragma unroll
      for (int Q = 0; Q < 4; ++Q) vv[P][Q] = 0.25f * both(cset[P], cset[Q]) * inv;
    if (vdn) {
#pragma unroll
      for (int P = 0; P < 4; ++P)
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) vdn[(((long)co * 4 + P) * 4 + Q) * Cin + ci] = vv[P][Q];
    }
    if (wq) {
      const int pd[2][2] = {{3, 1}, {2, 0}};
#pragma unroll
      for (int al = 0; al < 2; ++al)
#pragma unroll
        for (int be = 0; be < 2; ++be)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
              wq[((((long)(al * 2 + be) * Cin + ci) * 2 + u) * 2 + v) * rows + co] = vv[pd[al][u]][pd[be][v]];
    }
  }
}
__global__ __launch_bounds__(256) void sn_up_layouts_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                                            int rows, int Cin, float* __restrict__ wp,
                                                            float* __restrict__ vd, float* __restrict__ vdn,
                                                            float* __restrict__ wq) {
  sn_up_layouts_body(blockIdx.x, blockIdx.y, gridDim.x, w, sigma, rows, Cin, wp, vd, vdn, wq);
}

// rows per chunk: 24 rows of the widest layer (13824 columns) = 1.3 MB streamed per block column; at least 16 rows per chunk
static int sn_vblocks(int cols) {
  const long b = icg_cdiv(cols, 256);
  return (int)(b > SN_VBLOCKS ? SN_VBLOCKS : b);
}

static void sn_chunk_plan(int rows, int* ychunks, int* rpc) {
  int yc = (int)icg_cdiv(rows, 24);
  if (yc > SN_MAX_CHUNKS) yc = SN_MAX_CHUNKS;
  if (yc < 1) yc = 1;
  *rpc = (int)icg_cdiv(rows, yc);
  *ychunks = (int)icg_cdiv(rows, *rpc);
}

extern "C" size_t icg_sn_scratch_bytes(int rows, int Cin, int R) {
  const long cols = (long)Cin * R * R;
  return (size_t)((long)SN_MAX_CHUNKS * cols + cols + rows + 2) * sizeof(float) + SN_VBLOCKS * sizeof(double) + 256;
}

extern "C" int icg_sn_forward(const float* w, float* u, float* sv, int rows, int Cin, int R, float eps, int training,
                              float* v_out, float* u_out, float* sigma_out, float* w_ohwi, float* w_dgrad,
                              float* w_up_fprop, float* w_up_dgrad, float* w_down_fprop, float* w_down_dgrad,
                              void* scratch, size_t scratch_bytes, void* stream) {
  ICG_REQUIRE(w && u && v_out && u_out && sigma_out && w_ohwi && scratch);
  ICG_REQUIRE(rows > 0 && Cin > 0 && R >= 1);
  if (scratch_bytes < icg_sn_scratch_bytes(rows, Cin, R)) return ICG_ERR_WORKSPACE;
  const int cols = Cin * R * R;
  hipStream_t st = (hipStream_t)stream;
  const SnScratch sc = sn_scratch(scratch, rows, cols);
  int ychunks, rpc;
  sn_chunk_plan(rows, &ychunks, &rpc);
  const int nblk = sn_vblocks(cols);
  hipLaunchKernelGGL(sn_uw_partial_kernel, dim3((unsigned)icg_cdiv(cols, 1024), ychunks), dim3(256), 0, st, w, u, rows,
                     cols, rpc, sc.part);
  hipLaunchKernelGGL(sn_vsum_kernel, dim3(nblk), dim3(256), 0, st, (const float*)sc.part, ychunks, cols, sc.t, sc.ssq);
  hipLaunchKernelGGL(sn_wv_kernel, dim3((unsigned)icg_cdiv(rows, 4)), dim3(256), 0, st, w, (const float*)sc.t,
                     (const double*)sc.ssq, nblk, eps, rows, cols, v_out, sc.s);
  hipLaunchKernelGGL(sn_u_kernel, dim3(1), dim3(1024), 0, st, (const float*)sc.s, rows, eps, training, u, sv, u_out,
                     sigma_out);
  const long total = (long)rows * cols;
  long blocks = icg_cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sn_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, (const float*)sigma_out, rows, Cin,
                     R, w_ohwi, w_dgrad);
  if (w_up_fprop || w_up_dgrad || w_down_fprop || w_down_dgrad) {
    if (R != 3) return ICG_ERR_ARG;
    hipLaunchKernelGGL(sn_up_layouts_kernel, dim3((unsigned)icg_cdiv((long)rows * Cin, 256)), dim3(256), 0, st, w,
                       (const float*)sigma_out, rows, Cin, w_up_fprop, w_up_dgrad, w_down_fprop, w_down_dgrad);
  }
  return icg_check_launch();
}

// ---------------------------------------------------------------- all layers of a network in one pass
// The per-layer kernels above are launch-bound (76 layers x 5-6 launches x 4 forwards per training step).  The multi
// versions run one stage for up to ICG_SN_PACK layers per launch: blockIdx.z selects the layer (descriptors travel in the
// kernel arguments), blockIdx.x/y index inside the layer and blocks beyond a layer's own extent exit at once.  Same
// arithmetic in the same order as the single-layer path: results are bit-identical.
struct SnPack {
  icg_sn_layer l[ICG_SN_PACK];
  int ychunks[ICG_SN_PACK], rpc[ICG_SN_PACK], vblocks[ICG_SN_PACK];
  int n;
};

__global__ __launch_bounds__(256) void sn_uw_partial_multi_kernel(SnPack p) {
  const icg_sn_layer& L = p.l[blockIdx.z];
  const int cols = L.Cin * L.R * L.R;
  if ((int)blockIdx.x * 1024 >= cols || (int)blockIdx.y >= p.ychunks[blockIdx.z]) return;
  sn_uw_partial_body(blockIdx.x, blockIdx.y, 0, L.w, L.u, L.rows, cols, p.rpc[blockIdx.z], sn_scratch(L.scratch, L.rows, cols).part);
}
__global__ __launch_bounds__(256) void sn_vsum_multi_kernel(SnPack p) {
  const icg_sn_layer& L = p.l[blockIdx.z];
  const int cols = L.Cin * L.R * L.R;
  const int nblk = p.vblocks[blockIdx.z];
  if ((int)blockIdx.x >= nblk) return;
  const SnScratch sc = sn_scratch(L.scratch, L.rows, cols);
  sn_vsum_body(blockIdx.x, nblk, sc.part, p.ychunks[blockIdx.z], cols, sc.t, sc.ssq);
}
__global__ __launch_bounds__(256) void sn_wv_multi_kernel(SnPack p, float eps) {
  const icg_sn_layer& L = p.l[blockIdx.z];
  const int cols = L.Cin * L.R * L.R;
  const int gx = (L.rows + 3) / 4;                         // the single-layer launch geometry
  if ((int)blockIdx.x >= gx) return;
  const SnScratch sc = sn_scratch(L.scratch, L.rows, cols);
  sn_wv_body(blockIdx.x, gx, L.w, sc.t, sc.ssq, p.vblocks[blockIdx.z], eps, L.rows, cols, L.v_out, sc.s);
}
__global__ __launch_bounds__(1024) void sn_u_multi_kernel(SnPack p, float eps, int training) {
  const icg_sn_layer& L = p.l[blockIdx.z];
  const int cols = L.Cin * L.R * L.R;
  sn_u_body(0, 0, 1, sn_scratch(L.scratch, L.rows, cols).s, L.rows, eps, training, L.u, L.sv, L.u_out, L.sigma_out);
}
__global__ __launch_bounds__(256) void sn_scale_multi_kernel(SnPack p) {
  const icg_sn_layer& L = p.l[blockIdx.z];
  const long total = (long)L.rows * L.Cin * L.R * L.R;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;                        // the single-layer launch geometry (same element -> thread map)
  if ((long)blockIdx.x >= blocks) return;
  sn_scale_body(blockIdx.x, 0, (int)blocks, L.w, (const float*)L.sigma_out, L.rows, L.Cin, L.R, L.w_ohwi, L.w_dgrad);
}
__global__ __launch_bounds__(256) void sn_up_layouts_multi_kernel(SnPack p) {
  const icg_sn_layer& L = p.l[blockIdx.z];
  if (!(L.w_up_fprop || L.w_up_dgrad || L.w_down_fprop || L.w_down_dgrad)) return;
  if ((long)blockIdx.x * 256 >= (long)L.rows * L.Cin) return;
  sn_up_layouts_body(blockIdx.x, 0, 0, L.w, (const float*)L.sigma_out, L.rows, L.Cin, L.w_up_fprop, L.w_up_dgrad,
                     L.w_down_fprop, L.w_down_dgrad);
}

extern "C" int icg_sn_forward_multi(const icg_sn_layer* layers, int n, float eps, int training, void* stream) {
  ICG_REQUIRE(layers && n > 0);
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < n; base += ICG_SN_PACK) {
    SnPack p{};
    p.n = (n - base < ICG_SN_PACK) ? n - base : ICG_SN_PACK;
    unsigned gx_uw = 1, gy_uw = 1, gx_wv = 1, gx_sc = 1, gx_up = 0;
    for (int i = 0; i < p.n; ++i) {
      const icg_sn_layer& L = layers[base + i];
      ICG_REQUIRE(L.w && L.u && L.v_out && L.u_out && L.sigma_out && L.w_ohwi && L.scratch);
      ICG_REQUIRE(L.rows > 0 && L.Cin > 0 && L.R >= 1);
      if (L.scratch_bytes < icg_sn_scratch_bytes(L.rows, L.Cin, L.R)) return ICG_ERR_WORKSPACE;
      const bool lay = L.w_up_fprop || L.w_up_dgrad || L.w_down_fprop || L.w_down_dgrad;
      if (lay && L.R != 3) return ICG_ERR_ARG;
      p.l[i] = L;
      const int cols = L.Cin * L.R * L.R;
      int ychunks, rpc;
      sn_chunk_plan(L.rows, &ychunks, &rpc);
      p.ychunks[i] = ychunks; p.rpc[i] = rpc; p.vblocks[i] = sn_vblocks(cols);
      gx_uw = max(gx_uw, (unsigned)icg_cdiv(cols, 1024));
      gy_uw = max(gy_uw, (unsigned)ychunks);
      gx_wv = max(gx_wv, (unsigned)icg_cdiv(L.rows, 4));
      long sb = icg_cdiv((long)L.rows * cols, 256);
      if (sb > 2048) sb = 2048;
      gx_sc = max(gx_sc, (unsigned)sb);
      if (lay) gx_up = max(gx_up, (unsigned)icg_cdiv((long)L.rows * L.Cin, 256));
    }
    const unsigned nz = (unsigned)p.n;
    hipLaunchKernelGGL(sn_uw_partial_multi_kernel, dim3(gx_uw, gy_uw, nz), dim3(256), 0, st, p);
    hipLaunchKernelGGL(sn_vsum_multi_kernel, dim3(SN_VBLOCKS, 1, nz), dim3(256), 0, st, p);
    hipLaunchKernelGGL(sn_wv_multi_kernel, dim3(gx_wv, 1, nz), dim3(256), 0, st, p, eps);
    hipLaunchKernelGGL(sn_u_multi_kernel, dim3(1, 1, nz), dim3(1024), 0, st, p, eps, training);
    hipLaunchKernelGGL(sn_scale_multi_kernel, dim3(gx_sc, 1, nz), dim3(256), 0, st, p);
    if (gx_up) hipLaunchKernelGGL(sn_up_layouts_multi_kernel, dim3(gx_up, 1, nz), dim3(256), 0, st, p);
  }
  return icg_check_launch();
}

// ---------------------------------------------------------------- backward
// element idx of the PARAMETER layout [co][ci][tap]  ->  g = dw_hwio[tap][ci][co] + dw_ohwi[co][tap][ci]
// dw_up: phase-form weight gradient [al][be][u][v][ci][co] of the upsample-fused conv; tap r receives the phase taps
// (al,u) with r in S(al,u):  r=0: (0,0),(1,0)   r=1: (0,1),(1,0)   r=2: (0,1),(1,1)
// dw_down: gradient w.r.t. the 4x4 pooled kernel [P][Q][ci][co]; tap (r,s) receives 0.25 * sum_{a,b in {0,1}} [r+a][s+b]
__device__ __forceinline__ float sn_gather_g(const float* __restrict__ dw_hwio, const float* __restrict__ dw_ohwi,
                                              const float* __restrict__ dw_up, const float* __restrict__ dw_down,
                                              int co, int ci, int tap, int rows, int Cin, int RR) {
  float g = 0.f;
  if (dw_hwio) g += dw_hwio[((long)tap * Cin + ci) * rows + co];
  if (dw_ohwi) g += dw_ohwi[((long)co * RR + tap) * Cin + ci];
  if (dw_up) {
    const int r = tap / 3, s = tap - 3 * r;
    const int ru[2] = {r == 0 ? 0 : 1, r == 2 ? 1 : 0};     // u of the pair contributed by phase al = 0 / 1
    const int su[2] = {s == 0 ? 0 : 1, s == 2 ? 1 : 0};     // v of the pair contributed by phase be = 0 / 1
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int al = a, u = ru[a], be = b, v = su[b];
        g += dw_up[(((((long)(al * 2 + be)) * 2 + u) * 2 + v) * Cin + ci) * rows + co];
      }
  }
  if (dw_down) {
    const int r = tap / 3, s = tap - 3 * r;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc += dw_down[((((long)(r + a)) * 4 + (s + b)) * Cin + ci) * rows + co];
    g += 0.25f * acc;
  }
  return g;
}

// (the three stages are written as bodies over (block index bx, blocks gx) so that the per-layer kernels and the many-layer kernels
// of icg_sn_backward_multi run the same arithmetic in the same order)
__device__ __forceinline__ void sn_bwd_dot_body(int bx, int gx, const float* __restrict__ dw_hwio,
                                                const float* __restrict__ dw_ohwi, const float* __restrict__ dw_up,
                                                const float* __restrict__ dw_down, const float* __restrict__ w_ohwi, int rows,
                                                int Cin, int RR, double* __restrict__ part) {
  __shared__ double red[4];
  const long total = (long)rows * Cin * RR;
  const long stride = (long)gx * blockDim.x;
  double acc = 0.0;
  // idx enumerates OHWI; (row q = co * RR + tap, ci) advance with the loop (no 64-bit divisions per element)
  long idx = (long)bx * blockDim.x + threadIdx.x;
  long q = idx / Cin;
  int ci = (int)(idx - q * Cin);
  const long qs = stride / Cin;
  const int rs = (int)(stride - qs * Cin);
  for (; idx < total; idx += stride) {
    const int qi = (int)q;
    const int co = RR == 9 ? qi / 9 : (RR == 1 ? qi : qi / RR), tap = qi - co * RR;
    const float g = sn_gather_g(dw_hwio, dw_ohwi, dw_up, dw_down, co, ci, tap, rows, Cin, RR);
    acc += (double)g * (double)w_ohwi[idx];
    q += qs;
    ci += rs;
    if (ci >= Cin) { ci -= Cin; ++q; }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[bx] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const float* __restrict__ dw_hwio, const float* __restrict__ dw_ohwi,
                                                         const float* __restrict__ dw_up, const float* __restrict__ dw_down,
                                                         const float* __restrict__ w_ohwi, int rows, int Cin, int RR,
                                                         double* __restrict__ part) {
  sn_bwd_dot_body(blockIdx.x, gridDim.x, dw_hwio, dw_ohwi, dw_up, dw_down, w_ohwi, rows, Cin, RR, part);
}

__device__ __forceinline__ void sn_bwd_apply_body(int bx, int gx, const float* __restrict__ dw_hwio,
                                                  const float* __restrict__ dw_ohwi, const float* __restrict__ dw_up,
                                                  const float* __restrict__ dw_down, const float* __restrict__ u,
                                                  const float* __restrict__ v, const float* __restrict__ sigma,
                                                  const double* __restrict__ part, int nparts, int rows, int Cin, int RR,
                                                  float* __restrict__ dw, int accumulate) {
  __shared__ double s_red[4];
  {   // every block re-reduces the per-block partial dots: all lanes, fixed order (deterministic)
    double d = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) d += part[k];
    d = wave_sum_d(d);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = d;
  }
  __syncthreads();
  const float dot = (float)((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  const float inv_sigma = 1.0f / sigma[0];
  const long total = (long)rows * Cin * RR;
  const long stride = (long)gx * blockDim.x;
  // idx enumerates the parameter layout [co][ci][tap] (coalesced stores); column j = ci*RR + tap.  (co, j) advance with the loop
  // instead of being divided out of a 64-bit index per element -- the divisions were this kernel's whole run time
  const long CR = (long)Cin * RR;
  long idx = (long)bx * blockDim.x + threadIdx.x;
  int co = (int)(idx / CR), j = (int)(idx - (long)co * CR);
  const int qs = (int)(stride / CR), rs = (int)(stride - (long)qs * CR);
  for (; idx < total; idx += stride) {
    const int ci = RR == 9 ? j / 9 : (RR == 1 ? j : j / RR), tap = j - ci * RR;
    const float g = sn_gather_g(dw_hwio, dw_ohwi, dw_up, dw_down, co, ci, tap, rows, Cin, RR);
    const float corr = (u != nullptr && v != nullptr) ? dot * u[co] * v[j] : 0.f;
    const float val = (g - corr) * inv_sigma;
    dw[idx] = accumulate ? dw[idx] + val : val;
    co += qs;
    j += rs;
    if (j >= (int)CR) { j -= (int)CR; ++co; }
  }
}
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ dw_hwio, const float* __restrict__ dw_ohwi,
                                                           const float* __restrict__ dw_up, const float* __restrict__ dw_down,
                                                           const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ sigma, const double* __restrict__ part,
                                                           int nparts, int rows, int Cin, int RR, float* __restrict__ dw,
                                                           int accumulate) {
  sn_bwd_apply_body(blockIdx.x, gridDim.x, dw_hwio, dw_ohwi, dw_up, dw_down, u, v, sigma, part, nparts, rows, Cin, RR, dw, accumulate);
}

// Coalesced form of the gather + dot: the HWIO-ordered sources (dw_hwio and the phase / pooled gradients, all with co as
// the fastest index) are read 32x32 tiles at a time with co across lanes, transposed through LDS and written to
// g[co][tap][ci] (OHWI, the order of w_ohwi and dw_ohwi) with (tap, ci) across lanes; <g, w_ohwi> is accumulated on the way.
__device__ __forceinline__ void sn_bwd_gather_body(int bx, int gx, const float* __restrict__ dw_hwio,
                                                   const float* __restrict__ dw_ohwi, const float* __restrict__ dw_up,
                                                   const float* __restrict__ dw_down, const float* __restrict__ w_ohwi, int rows,
                                                   int Cin, int RR, float* __restrict__ g_out, double* __restrict__ part) {
  __shared__ float tile[32][33];
  __shared__ double red[4];
  const int K = Cin * RR;
  const int tk = (K + 31) / 32, tc = (rows + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  double acc = 0.0;
  for (int tno = bx; tno < tk * tc; tno += gx) {
    const int k0 = (tno % tk) * 32, c0 = (tno / tk) * 32;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = k0 + ty + 8 * r, co = c0 + tx;
      float v = 0.f;
      if (kk < K && co < rows) {
        const int tap = kk / Cin, ci = kk - tap * Cin;
        v = sn_gather_g(dw_hwio, nullptr, dw_up, dw_down, co, ci, tap, rows, Cin, RR);
      }
      tile[ty + 8 * r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = c0 + ty + 8 * r, kk = k0 + tx;
      if (kk < K && co < rows) {
        const long o = (long)co * K + kk;
        float g = tile[tx][ty + 8 * r];
        if (dw_ohwi) g += dw_ohwi[o];
        g_out[o] = g;
        acc += (double)g * (double)w_ohwi[o];
      }
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[bx] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sn_bwd_gather_kernel(const float* __restrict__ dw_hwio, const float* __restrict__ dw_ohwi,
                                                            const float* __restrict__ dw_up, const float* __restrict__ dw_down,
                                                            const float* __restrict__ w_ohwi, int rows, int Cin, int RR,
                                                            float* __restrict__ g_out, double* __restrict__ part) {
  sn_bwd_gather_body(blockIdx.x, gridDim.x, dw_hwio, dw_ohwi, dw_up, dw_down, w_ohwi, rows, Cin, RR, g_out, part);
}

#define SN_BWD_PARTS 2048     // blocks of the gather / dot pass (8 per CU: the pass is a latency-bound transpose at 1 per CU)
extern "C" size_t icg_sn_backward_scratch_bytes(int rows, int Cin, int R) {
  return SN_BWD_PARTS * sizeof(double) + (size_t)rows * Cin * R * R * sizeof(float);
}

extern "C" int icg_sn_backward(const float* dw_hwio, const float* dw_ohwi, const float* dw_up, const float* dw_down,
                               const float* w_ohwi, const float* u_saved,
                               const float* v_saved, const float* sigma, int rows, int Cin, int R, float* dw,
                               int accumulate, void* scratch, size_t scratch_bytes, void* stream) {
  ICG_REQUIRE((dw_hwio || dw_ohwi || dw_up || dw_down) && w_ohwi && sigma && dw && scratch);
  if (dw_up || dw_down) ICG_REQUIRE(R == 3);
  ICG_REQUIRE(rows > 0 && Cin > 0 && R >= 1);
  if (scratch_bytes < SN_BWD_PARTS * sizeof(double)) return ICG_ERR_WORKSPACE;
  const int RR = R * R;
  const long total = (long)rows * Cin * RR;
  int nparts = (int)(icg_cdiv(total, 1024) > SN_BWD_PARTS ? SN_BWD_PARTS : icg_cdiv(total, 1024));
  hipStream_t st = (hipStream_t)stream;
  if ((dw_hwio || dw_up || dw_down) && scratch_bytes >= icg_sn_backward_scratch_bytes(rows, Cin, R)) {
    // two coalesced passes: transpose-gather into OHWI order (+ dot), then the correction in parameter order
    float* g = reinterpret_cast<float*>(reinterpret_cast<double*>(scratch) + SN_BWD_PARTS);
    hipLaunchKernelGGL(sn_bwd_gather_kernel, dim3(nparts), dim3(256), 0, st, dw_hwio, dw_ohwi, dw_up, dw_down, w_ohwi, rows,
                       Cin, RR, g, (double*)scratch);
    long blocks2 = icg_cdiv(total, 256);
    if (blocks2 > 2048) blocks2 = 2048;
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3((unsigned)blocks2), dim3(256), 0, st, (const float*)nullptr,
                       (const float*)g, (const float*)nullptr, (const float*)nullptr, u_saved, v_saved, sigma,
                       (const double*)scratch, nparts, rows, Cin, RR, dw, accumulate);
    return icg_check_launch();
  }
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(nparts), dim3(256), 0, st, dw_hwio, dw_ohwi, dw_up, dw_down, w_ohwi, rows, Cin, RR,
                     (double*)scratch);
  long blocks = icg_cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dw_hwio, dw_ohwi, dw_up, dw_down, u_saved,
                     v_saved, sigma, (const double*)scratch, nparts, rows, Cin, RR, dw, accumulate);
  return icg_check_launch();
}

// ---------------------------------------------------------------- backward of many layers in two launches
// icg_sn_backward per layer is two launches of a few microseconds each, 150 per training step, issued from inside the autograd
// nodes of the layers.  The autograd graph of ic_gan_amd.ops routes the raw weight gradients of a GROUP of layers through one node
// (ops.SNGroupFn), whose backward calls this: stage 1 (transpose-gather + <g, W/sigma> partial sums, or the dot alone for gradients
// that arrive in OHWI order) and stage 2 (the rank-one correction in parameter order) for up to ICG_SN_PACK layers per launch pair;
// blockIdx.z selects the layer, blocks past a layer's own geometry exit at once.  Same bodies, same per-layer geometry: bit-identical
// to icg_sn_backward per layer.
struct SnBwdPack {
  icg_sn_bwd_item l[ICG_SN_PACK];
  int nparts[ICG_SN_PACK], blocks2[ICG_SN_PACK], gather[ICG_SN_PACK];
  int n;
};

__global__ __launch_bounds__(256) void sn_bwd_stage1_multi_kernel(SnBwdPack p) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x >= p.nparts[z]) return;
  const icg_sn_bwd_item& L = p.l[z];
  const int RR = L.R * L.R;
  double* part = reinterpret_cast<double*>(L.scratch);
  if (p.gather[z])
    sn_bwd_gather_body(blockIdx.x, p.nparts[z], L.dw_hwio, L.dw_ohwi, L.dw_up, L.dw_down, L.w_ohwi, L.rows, L.Cin, RR,
                       reinterpret_cast<float*>(part + SN_BWD_PARTS), part);
  else
    sn_bwd_dot_body(blockIdx.x, p.nparts[z], L.dw_hwio, L.dw_ohwi, L.dw_up, L.dw_down, L.w_ohwi, L.rows, L.Cin, RR, part);
}
__global__ __launch_bounds__(256) void sn_bwd_stage2_multi_kernel(SnBwdPack p) {
  const int z = blockIdx.z;
  if ((int)blockIdx.x >= p.blocks2[z]) return;
  const icg_sn_bwd_item& L = p.l[z];
  const int RR = L.R * L.R;
  const double* part = reinterpret_cast<const double*>(L.scratch);
  if (p.gather[z])
    sn_bwd_apply_body(blockIdx.x, p.blocks2[z], nullptr, reinterpret_cast<const float*>(part + SN_BWD_PARTS), nullptr, nullptr, L.u, L.v,
                      L.sigma, part, p.nparts[z], L.rows, L.Cin, RR, L.dw, L.accumulate);
  else
    sn_bwd_apply_body(blockIdx.x, p.blocks2[z], L.dw_hwio, L.dw_ohwi, L.dw_up, L.dw_down, L.u, L.v, L.sigma, part, p.nparts[z], L.rows,
                      L.Cin, RR, L.dw, L.accumulate);
}

extern "C" int icg_sn_backward_multi(const icg_sn_bwd_item* items, int n, void* stream) {
  ICG_REQUIRE(items && n > 0);
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < n; base += ICG_SN_PACK) {
    SnBwdPack p{};
    p.n = (n - base < ICG_SN_PACK) ? n - base : ICG_SN_PACK;
    unsigned g1 = 1, g2 = 1;
    for (int i = 0; i < p.n; ++i) {
      const icg_sn_bwd_item& L = items[base + i];
      ICG_REQUIRE((L.dw_hwio || L.dw_ohwi || L.dw_up || L.dw_down) && L.w_ohwi && L.sigma && L.dw && L.scratch);
      if (L.dw_up || L.dw_down) ICG_REQUIRE(L.R == 3);
      ICG_REQUIRE(L.rows > 0 && L.Cin > 0 && L.R >= 1);
      if (L.scratch_bytes < SN_BWD_PARTS * sizeof(double)) return ICG_ERR_WORKSPACE;
      const long total = (long)L.rows * L.Cin * L.R * L.R;
      p.l[i] = L;
      p.nparts[i] = (int)(icg_cdiv(total, 1024) > SN_BWD_PARTS ? SN_BWD_PARTS : icg_cdiv(total, 1024));
      p.gather[i] = ((L.dw_hwio || L.dw_up || L.dw_down) && L.scratch_bytes >= icg_sn_backward_scratch_bytes(L.rows, L.Cin, L.R)) ? 1 : 0;
      long b2 = icg_cdiv(total, 256);
      if (b2 > 2048) b2 = 2048;
      p.blocks2[i] = (int)b2;
      g1 = max(g1, (unsigned)p.nparts[i]);
      g2 = max(g2, (unsigned)b2);
    }
    hipLaunchKernelGGL(sn_bwd_stage1_multi_kernel, dim3(g1, 1, (unsigned)p.n), dim3(256), 0, st, p);
    hipLaunchKernelGGL(sn_bwd_stage2_multi_kernel, dim3(g2, 1, (unsigned)p.n), dim3(256), 0, st, p);
  }
  return icg_check_launch();
}
