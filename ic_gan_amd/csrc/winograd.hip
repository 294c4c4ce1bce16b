// Winograd F(2x2, 3x3) for the wide 3x3 stride-1 convolutions (forward and data gradient):
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 2x2 output tile / 4x4 input tile, summed over input channels
// turns the convolution into 16 independent [tiles x Cin] x [Cin x Cout] GEMMs with 16/36 of the multiply-adds.  The GEMMs
// run on the MFMA kernel (icg_gemm_batched, K = Cin); the input transform (with the fused BN-affine / ReLU prologue and the
// zero padding) and the output transform (with bias / residual epilogue) are HBM-bound passes over 4x the activation volume,
// so the form only pays for wide layers (Cin, Cout >= 256), where the GEMM time dominates; ops.py applies that rule.
// Same mathematical result as the direct convolution; rounding differs at the 1e-6 level (fp32 transforms).
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
#include "icg_common.h"
#include <stdlib.h>
#include <vector>
#include <atomic>
#include <mutex>

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// ICG_RES_RELU_MASK: v where the ReLU input r was positive, else 0
__device__ __forceinline__ float4 f4relu_mask(float4 v, float4 r) {
  return make_float4(r.x > 0.f ? v.x : 0.f, r.y > 0.f ? v.y : 0.f, r.z > 0.f ? v.z : 0.f, r.w > 0.f ? v.w : 0.f);
}
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// V[xi][t][c] = (B^T d B)[xi],  d = act(x) on the 4x4 window of tile t (zero outside the image), xi = 4*i + j
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, long ssb, float* __restrict__ V,
                                                         int B, int H, int W, int C4, int affine, int relu) {
  const int th = H >> 1, tw = W >> 1;
  const long T = (long)B * th * tw;
  const long total = T * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long t2 = t / tw;
    const int ty = (int)(t2 % th);
    const int b = (int)(t2 / th);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (affine) {
      sc = *reinterpret_cast<const float4*>(scale + (long)b * ssb + 4 * c4);
      sh = *reinterpret_cast<const float4*>(shift + (long)b * ssb + 4 * c4);
    }
    const float4* xp = reinterpret_cast<const float4*>(x) + (long)b * H * W * C4 + c4;
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 2 * ty - 1 + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int w = 2 * tx - 1 + s;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
          v = xp[((long)h * W + w) * C4];
          if (affine) {
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
          }
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        d[r][s] = v;
      }
    }
    float4 u[4][4];      // B^T d
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u[0][s] = f4sub(d[0][s], d[2][s]);
      u[1][s] = f4add(d[1][s], d[2][s]);
      u[2][s] = f4sub(d[2][s], d[1][s]);
      u[3][s] = f4sub(d[1][s], d[3][s]);
    }
    float4* vp = reinterpret_cast<float4*>(V) + t * C4 + c4;
    const long plane = T * C4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vp[(4 * r + 0) * plane] = f4sub(u[r][0], u[r][2]);
      vp[(4 * r + 1) * plane] = f4add(u[r][1], u[r][2]);
      vp[(4 * r + 2) * plane] = f4sub(u[r][2], u[r][1]);
      vp[(4 * r + 3) * plane] = f4sub(u[r][1], u[r][3]);
    }
  }
}

// y[b, 2ty+a, 2tx+c, co] = (A^T m A)[a][c] + bias[co] + residual
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, const float* __restrict__ bias,
                                                          const float* __restrict__ res, int res_up, float alpha,
                                                          float* __restrict__ y, int B, int H, int W, int C4) {
  const int th = H >> 1, tw = W >> 1;
  const long T = (long)B * th * tw;
  const long total = T * C4;
  const long plane = T * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long t2 = t / tw;
    const int ty = (int)(t2 % th);
    const long b = t2 / th;
    const float4* mp = reinterpret_cast<const float4*>(Mb) + t * C4 + c4;
    float4 s[2][4];      // A^T m
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 m0 = mp[(0 + c) * plane], m1 = mp[(4 + c) * plane], m2 = mp[(8 + c) * plane], m3 = mp[(12 + c) * plane];
      s[0][c] = f4add(f4add(m0, m1), m2);
      s[1][c] = f4sub(f4sub(m1, m2), m3);
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * c4);
    float4 rlow = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res && res_up == 1) rlow = reinterpret_cast<const float4*>(res)[((b * th + ty) * tw + tx) * C4 + c4];   // one source pixel per tile
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float4 o0 = f4add(f4add(s[a][0], s[a][1]), s[a][2]);
      const float4 o1 = f4sub(f4sub(s[a][1], s[a][2]), s[a][3]);
      const long p0 = ((b * H + (2 * ty + a)) * W + 2 * tx) * C4 + c4;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4 o = c ? o1 : o0;
        float4 v = make_float4(alpha * o.x + bv.x, alpha * o.y + bv.y, alpha * o.z + bv.z, alpha * o.w + bv.w);
        if (res) {
          const float4 r = (res_up == 1) ? rlow : reinterpret_cast<const float4*>(res)[p0 + (long)c * C4];
          v = (res_up == 2) ? f4relu_mask(v, r) : f4add(v, r);           // 2: ICG_RES_RELU_MASK
        }
        reinterpret_cast<float4*>(y)[p0 + (long)c * C4] = v;
      }
    }
  }
}

// U[xi][n][k] = (G g G^T)[xi] for g = w[n][.][.][k]   (w: [N][3][3][K], the OHWI or the dgrad layout)
// (blk, nblk: this block's index and the block count of ITS tensor -- the single-tensor kernel passes blockIdx.x / gridDim.x,
// the batched one the position inside the tensor's block range)
__device__ __forceinline__ void wino_weight_body(const float* __restrict__ w, float* __restrict__ U, int N, int K, unsigned blk,
                                                 unsigned nblk) {
  const long total = (long)N * K;
  const long gstride = (long)nblk * blockDim.x;
  for (long i = (long)blk * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int k = (int)(i % K);
    const long n = i / K;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) g[r][s] = w[((n * 3 + r) * 3 + s) * K + k];
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      U[(4 * r + 0) * total + i] = t[r][0];
      U[(4 * r + 1) * total + i] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      U[(4 * r + 2) * total + i] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      U[(4 * r + 3) * total + i] = t[r][2];
    }
  }
}
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int N, int K) {
  wino_weight_body(w, U, N, K, blockIdx.x, gridDim.x);
}

// DY[xi][t][c] = (A dy A^T)[xi] on the 2x2 output tile t,  A = [1 0; 1 1; 1 -1; 0 -1]     (weight-gradient side)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ DY, int B, int H,
                                                      int W, int C4) {
  const int th = H >> 1, tw = W >> 1;
  const long T = (long)B * th * tw;
  const long total = T * C4, plane = T * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long t2 = t / tw;
    const int ty = (int)(t2 % th);
    const long b = t2 / th;
    const float4* gp = reinterpret_cast<const float4*>(dy) + ((b * H + 2 * ty) * W + 2 * tx) * C4 + c4;
    const float4 d00 = gp[0], d01 = gp[C4], d10 = gp[(long)W * C4], d11 = gp[(long)W * C4 + C4];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    // rows of A dy: r0 = d0, r1 = d0 + d1, r2 = d0 - d1, r3 = -d1   (each a pair over the two columns)
    const float4 r[4][2] = {{d00, d01}, {f4add(d00, d10), f4add(d01, d11)}, {f4sub(d00, d10), f4sub(d01, d11)},
                            {f4sub(z, d10), f4sub(z, d11)}};
    float4* op = reinterpret_cast<float4*>(DY) + t * C4 + c4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      op[(4 * a + 0) * plane] = r[a][0];
      op[(4 * a + 1) * plane] = f4add(r[a][0], r[a][1]);
      op[(4 * a + 2) * plane] = f4sub(r[a][0], r[a][1]);
      op[(4 * a + 3) * plane] = f4sub(z, r[a][1]);
    }
  }
}

// dw[r][s][ci][co] = (G^T dU G)[r][s],  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]      (HWIO, like the direct weight gradient)
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, long n) {
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride) {
    float u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) u[a][b] = dU[(long)(4 * a + b) * n + i];
    float t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
      t[1][b] = 0.5f * (u[1][b] - u[2][b]);
      t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      dw[(long)(3 * r + 0) * n + i] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      dw[(long)(3 * r + 1) * n + i] = 0.5f * (t[r][1] - t[r][2]);
      dw[(long)(3 * r + 2) * n + i] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
  }
}

void icg_gemm_mark_planes(int on);       // gemm_conv.hip: launch the following GEMMs as icg_gemm_planes_kernel (profile name)
extern "C" int icg_gemm_last_variant(int* out4);

// measurement hook for bench.py: HIP events on the launch stream around plane-GEMM launches.  Off by default -- the product path
// then pays one relaxed atomic load per plane-GEMM call and touches no shared state; when enabled, the tables below are updated
// under a mutex, so launches from several threads are safe.  icg_planes_timing(P): every launch is COUNTED per distinct shape
// (M, N, K, planes, NN / TN), every P-th launch of a shape is TIMED (P = 1: all of them); a shape's launches do the same work, so
// its total time is estimated as (mean of its timed launches) x (its launch count) -- stratified sampling that keeps the event
// pairs (each one drains the stream's launch pipeline for a few microseconds) out of three launches in four.
struct PlanesShape { double M, N, K; int planes, tn_form; int amode, tn, levels; long launches, timed; double flops, bytes; };
struct PlanesRecord { hipEvent_t e0, e1; int shape; };
static std::vector<PlanesShape> g_planes_shapes;
static std::vector<PlanesRecord> g_planes_records;
static std::mutex g_planes_mutex;
static std::atomic<int> g_planes_timing{0};

struct PlanesScope {
  hipStream_t st;
  int shape = -1;
  hipEvent_t e0 = nullptr;
  // M x N x K per plane; tn_form: weight-gradient (TN) launch
  PlanesScope(void* stream, int planes, double M, double N, double K, int tn_form = 0) : st((hipStream_t)stream) {
    icg_gemm_mark_planes(1);
    const int period = g_planes_timing.load(std::memory_order_relaxed);
    if (period <= 0) return;
    bool timed = false;
    {
      std::lock_guard<std::mutex> lock(g_planes_mutex);
      for (size_t i = 0; i < g_planes_shapes.size(); ++i) {
        const PlanesShape& h = g_planes_shapes[i];
        if (h.M == M && h.N == N && h.K == K && h.planes == planes && h.tn_form == tn_form) { shape = (int)i; break; }
      }
      if (shape < 0) {
        g_planes_shapes.push_back(PlanesShape{M, N, K, planes, tn_form, 0, 0, 2, 0, 0, 2.0 * planes * M * N * K,
                                              4.0 * planes * (M * K + K * N + M * N)});
        shape = (int)g_planes_shapes.size() - 1;
      }
      timed = (g_planes_shapes[shape].launches++ % period) == 0;
    }
    if (timed && hipEventCreate(&e0) == hipSuccess) hipEventRecord(e0, st);
  }
  ~PlanesScope() {
    icg_gemm_mark_planes(0);
    if (shape < 0) return;
    int v[4] = {0, 0, 0, 0};
    icg_gemm_last_variant(v);
    hipEvent_t e1 = nullptr;
    if (e0 && hipEventCreate(&e1) == hipSuccess) hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lock(g_planes_mutex);
    if ((size_t)shape >= g_planes_shapes.size()) {        // tables were reset while this launch was in flight
      if (e0) hipEventDestroy(e0);
      if (e1) hipEventDestroy(e1);
      return;
    }
    PlanesShape& h = g_planes_shapes[shape];
    h.amode = v[0]; h.tn = v[2]; h.levels = (v[3] == 4) ? 1 : 2;
    if (e0 && e1) g_planes_records.push_back(PlanesRecord{e0, e1, shape});
    else if (e0) hipEventDestroy(e0);
  }
};

extern "C" int icg_planes_timing(int enable) {
  std::lock_guard<std::mutex> lock(g_planes_mutex);
  for (auto& r : g_planes_records) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  g_planes_records.clear();
  g_planes_shapes.clear();
  g_planes_timing.store(enable > 0 ? enable : 0);
  return ICG_OK;
}

// out[i] = {amode, tn + 10 * (levels == 1), planes, launches, total ms, total executed flops, total operand bytes} per distinct
// (amode, tn, levels, planes) -- levels: accumulation levels of the kernel that ran (icg_gemm_planes1_kernel / icg_gemm_planes_kernel);
// total ms = sum over shapes of (mean timed launch) x launches (exact when the sampling period is 1);
// returns the number of rows written (synchronises on the recorded events)
extern "C" int icg_planes_timing_drain(double* out, int max_rows) {
  ICG_REQUIRE(out && max_rows > 0);
  std::lock_guard<std::mutex> lock(g_planes_mutex);
  std::vector<double> ms_sum(g_planes_shapes.size(), 0.0);
  for (auto& h : g_planes_shapes) h.timed = 0;
  for (auto& r : g_planes_records) {
    if ((size_t)r.shape >= g_planes_shapes.size() || hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    ms_sum[r.shape] += ms;
    g_planes_shapes[r.shape].timed += 1;
  }
  int rows = 0;
  for (size_t i = 0; i < g_planes_shapes.size(); ++i) {
    const PlanesShape& h = g_planes_shapes[i];
    if (h.timed == 0 || h.launches == 0) continue;
    const double code = h.tn + (h.levels == 1 ? 10 : 0);
    int k = 0;
    for (; k < rows; ++k)
      if ((int)out[7 * k] == h.amode && out[7 * k + 1] == code && (int)out[7 * k + 2] == h.planes) break;
    if (k == rows) {
      if (rows == max_rows) continue;
      out[7 * k] = h.amode; out[7 * k + 1] = code; out[7 * k + 2] = h.planes;
      out[7 * k + 3] = out[7 * k + 4] = out[7 * k + 5] = out[7 * k + 6] = 0.0;
      ++rows;
    }
    out[7 * k + 3] += (double)h.launches;
    out[7 * k + 4] += ms_sum[i] / (double)h.timed * (double)h.launches;
    out[7 * k + 5] += h.flops * (double)h.launches;
    out[7 * k + 6] += h.bytes * (double)h.launches;
  }
  return rows;
}
extern "C" int icg_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int transA, int transB,
                                int64_t strideA, int64_t strideB, int64_t strideC, int batch, float alpha, void* stream);

extern "C" int icg_plane_gemm(const float* A, const float* B, float* C, int M, int N, int K, int planes, float alpha,
                              void* stream) {
  ICG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && planes > 0);
  PlanesScope ps(stream, planes, M, N, K);
  return icg_gemm_batched(A, B, C, M, N, K, 0, 1, (int64_t)M * K, (int64_t)N * K, (int64_t)M * N, planes, alpha, stream);
}

extern "C" size_t icg_gemm_tn_batched_workspace_bytes(int M, int N, int K, int batch);
extern "C" int icg_gemm_tn_batched(const float* A, const float* B, float* C, int M, int N, int K, int64_t strideA,
                                   int64_t strideB, int64_t strideC, int batch, void* workspace, size_t workspace_bytes,
                                   void* stream);

extern "C" size_t icg_plane_gemm_tn_workspace_bytes(int M, int N, int K, int planes) {
  return icg_gemm_tn_batched_workspace_bytes(M, N, K, planes);
}

extern "C" int icg_plane_gemm_tn(const float* A, const float* B, float* C, int M, int N, int K, int planes, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && planes > 0);
  PlanesScope ps(stream, planes, M, N, K, 1);
  return icg_gemm_tn_batched(A, B, C, M, N, K, (int64_t)K * M, (int64_t)K * N, (int64_t)M * N, planes, workspace, workspace_bytes,
                             stream);
}

static size_t wino_al(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t icg_conv2d_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const size_t T = (size_t)B * (H / 2) * (W / 2);
  return wino_al(16 * T * Cin * sizeof(float)) + wino_al(16 * T * Cout * sizeof(float)) +
         wino_al((size_t)16 * Cin * Cout * sizeof(float)) + wino_al(icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)T, 16));
}

// dw[r][s][ci][co] of the 3x3 / stride-1 / pad-1 convolution of act(x) given dy, through the Winograd domain:
//   dU[xi] = V[xi]^T DY[xi]   (V = B^T act(x) B per tile, DY = A dy A^T per tile),   dw = G^T dU G
extern "C" int icg_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, const float* scale, const float* shift,
                                     int64_t ss_bstride, int B, int H, int W, int Cin, int Cout, unsigned flags,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE((H % 2 == 0) && (W % 2 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && !(flags & ICG_UPSAMPLE2X));
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift && (ss_bstride % 4 == 0));
  if (workspace_bytes < icg_conv2d_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout)) return ICG_ERR_WORKSPACE;
  const long T = (long)B * (H / 2) * (W / 2);
  ICG_REQUIRE(T * 16 < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  float* V = (float*)base;                    base += wino_al(16 * T * Cin * sizeof(float));
  float* DY = (float*)base;                   base += wino_al(16 * T * Cout * sizeof(float));
  float* dU = (float*)base;                   base += wino_al((size_t)16 * Cin * Cout * sizeof(float));
  void* gws = base;
  const size_t gws_bytes = icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)T, 16);
  long nb = icg_cdiv(T * (Cin / 4), 256);
  if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, scale, shift, (long)ss_bstride, V, B, H, W,
                     Cin / 4, (flags & ICG_PRE_AFFINE) ? 1 : 0, (flags & ICG_PRE_RELU) ? 1 : 0);
  nb = icg_cdiv(T * (Cout / 4), 256);
  if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
  hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)nb), dim3(256), 0, st, dy, DY, B, H, W, Cout / 4);
  int rc;
  { PlanesScope ps(stream, 16, Cin, Cout, (double)T, 1); rc = icg_gemm_tn_batched(V, DY, dU, Cin, Cout, (int)T, T * Cin, T * Cout, (long)Cin * Cout, 16, gws, gws_bytes, stream); }
  if (rc != ICG_OK) return rc;
  const long n = (long)Cin * Cout;
  nb = icg_cdiv(n, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)dU, dw, n);
  return icg_check_launch();
}

extern "C" int icg_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int transA, int transB,
                                int64_t strideA, int64_t strideB, int64_t strideC, int batch, float alpha, void* stream);

extern "C" int icg_wino_weight_transform(const float* w, float* U, int N, int K, void* stream) {
  ICG_REQUIRE(w && U && N > 0 && K > 0);
  long blocks = icg_cdiv((long)N * K, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, U, N, K);
  return icg_check_launch();
}

extern "C" size_t icg_conv2d_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const size_t T = (size_t)B * (H / 2) * (W / 2);
  return 16 * T * ((size_t)Cin + (size_t)Cout) * sizeof(float);
}

// out = conv3x3(act(x), w) + bias + residual with w given in the Winograd domain (U from icg_wino_weight_transform)
extern "C" int icg_conv2d_wino_fprop(const float* x, const float* U, const float* bias, const float* residual, float* out,
                                     const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W,
                                     int Cin, int Cout, unsigned flags, float alpha, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  ICG_REQUIRE(x && U && out && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE((H % 2 == 0) && (W % 2 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && !(flags & ICG_UPSAMPLE2X));
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift && (ss_bstride % 4 == 0));
  if (workspace_bytes < icg_conv2d_wino_workspace_bytes(B, H, W, Cin, Cout)) return ICG_ERR_WORKSPACE;
  const long T = (long)B * (H / 2) * (W / 2);
  ICG_REQUIRE(T * 16 < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  float* V = (float*)workspace;
  float* Mb = V + 16 * T * Cin;
  long nb = icg_cdiv(T * (Cin / 4), 256);
  if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, scale, shift, (long)ss_bstride, V, B, H, W,
                     Cin / 4, (flags & ICG_PRE_AFFINE) ? 1 : 0, (flags & ICG_PRE_RELU) ? 1 : 0);
  int rc;
  { PlanesScope ps(stream, 16, (double)T, Cout, Cin); rc = icg_gemm_batched(V, U, Mb, (int)T, Cout, Cin, 0, 1, T * Cin, (long)Cout * Cin, T * Cout, 16, 1.0f, stream); }
  if (rc != ICG_OK) return rc;
  nb = icg_cdiv(T * (Cout / 4), 256);
  if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
  hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float*)Mb, bias, residual,
                     icg_res_mode(flags), alpha, out, B, H, W, Cout / 4);
  return icg_check_launch();
}

// =====================================================================================================================
// Winograd F(4x4, 3x3): 4x4 output tiles from 6x6 input windows, 36 GEMMs over 1/16 of the pixels = 1/4 of the direct
// multiply-adds, and transform passes over only 2.25x the activation volume (F(2x2,3x3): 4x).  Interpolation points
// 0, +-1, +-2, inf; fp32 transforms; results agree with the direct kernel to ~1e-5 relative (checked in the tests).
//   1-D input  t = B^T d : t0 = 4d0-5d2+d4, t1 = -4d1-4d2+d3+d4, t2 = 4d1-4d2-d3+d4, t3 = -2d1-d2+2d3+d4,
//                          t4 = 2d1-d2-2d3+d4, t5 = 4d1-5d3+d5
//   1-D output y = A^T m : y0 = m0+m1+m2+m3+m4, y1 = m1-m2+2m3-2m4, y2 = m1+m2+4m3+4m4, y3 = m1-m2+8m3-8m4+m5
//   1-D weight u = G g   : u0 = g0/4, u1 = -(g0+g1+g2)/6, u2 = -(g0-g1+g2)/6, u3 = g0/24+g1/12+g2/6,
//                          u4 = g0/24-g1/12+g2/6, u5 = g2
//
// Resample-fused layers (GBlock conv1 = nearest x2 upsample -> conv3x3, DBlock conv2 = conv3x3 -> 2x2 average pool) in the
// same domain: the 6-pixel window of an upsampled signal is d = [l0 l1 l1 l2 l2 l3] (tiles start on even pixels), for which
// t2 = 4d1-4d2-d3+d4 = 0 identically, and the pooled output p0 = y0+y1 = m0+2m1+3m3-m4, p1 = y2+y3 = 2m1+12m3-4m4+m5 does not
// read m2.  Component 2 therefore drops out in both dimensions: 25 GEMMs instead of 36 per tile, i.e. 25/64 of the
// multiply-adds of the 2x2-phase / 4x4-stride-2 forms (and 25/144 of the reference op graph's).  NP = 5 below selects that
// 25-plane layout (components 0,1,3,4,5 -> slots 0..4); the adjoint operations (data and weight gradients) have the same
// structure: the gradient of a pooled output is an upsampled signal and vice versa.
__device__ __forceinline__ float4 f4s(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4fma(float4 a, float s, float4 c) {
  return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <int NP> __device__ __forceinline__ constexpr int w4_slot(int k) { return NP == 6 ? k : (k < 2 ? k : k - 1); }
template <int NP> __device__ __forceinline__ constexpr bool w4_has(int k) { return NP == 6 || k != 2; }

__device__ __forceinline__ void w4_in6(const float4 d[6], float4 t[6]) {
  t[0] = f4add(f4fma(d[2], -5.f, f4s(d[0], 4.f)), d[4]);
  const float4 a = f4fma(d[2], -4.f, d[4]), b = f4fma(d[1], -4.f, d[3]);     // a = d4-4d2, b = d3-4d1
  t[1] = f4add(a, b);
  t[2] = f4sub(a, b);
  const float4 c = f4sub(d[4], d[2]), e = f4s(f4sub(d[3], d[1]), 2.f);       // c = d4-d2, e = 2(d3-d1)
  t[3] = f4add(c, e);
  t[4] = f4sub(c, e);
  t[5] = f4add(f4fma(d[3], -5.f, f4s(d[1], 4.f)), d[5]);
}
// the same for the upsampled window d = [l0 l1 l1 l2 l2 l3]:  t0 = 4l0-5l1+l2, t1 = -8l1+2l2, t2 = 0, t3 = 3(l2-l1),
// t4 = l1-l2, t5 = 4l1-5l2+l3
__device__ __forceinline__ void w4_in_up(const float4 l[4], float4 t[6]) {
  t[0] = f4add(f4fma(l[1], -5.f, f4s(l[0], 4.f)), l[2]);
  t[1] = f4fma(l[1], -8.f, f4s(l[2], 2.f));
  t[2] = f4zero();
  const float4 c = f4sub(l[2], l[1]);
  t[3] = f4s(c, 3.f);
  t[4] = f4s(c, -1.f);
  t[5] = f4add(f4fma(l[2], -5.f, f4s(l[1], 4.f)), l[3]);
}

// V[slot(i)*NP + slot(j)][t][c] = (B^T d B)[i][j], one thread per (tile, channel quad).  UP = 0: d = act(x) on the 6x6 window
// of tile t of x [B][H][W][C];  UP = 1: d = the window of the nearest-x2 upsampled act(x), x [B][H/2][W/2][C] (4x4 loads).
template <int UP, int NP>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long ssb, float* __restrict__ V,
                                                          int B, int H, int W, int C4, int affine, int relu) {
  static_assert(!UP || NP == 5, "the upsampled window has no component 2");
  const int th = H >> 2, tw = W >> 2;
  const long T = (long)B * th * tw;
  const long total = T * C4, plane = T * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  const int Hx = UP ? (H >> 1) : H, Wx = UP ? (W >> 1) : W;       // dims of the stored tensor
  constexpr int NL = UP ? 4 : 6;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long t2 = t / tw;
    const int ty = (int)(t2 % th);
    const int b = (int)(t2 / th);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (affine) {
      sc = *reinterpret_cast<const float4*>(scale + (long)b * ssb + 4 * c4);
      sh = *reinterpret_cast<const float4*>(shift + (long)b * ssb + 4 * c4);
    }
    const float4* xp = reinterpret_cast<const float4*>(x) + (long)b * Hx * Wx * C4 + c4;
    const int h0 = UP ? 2 * ty - 1 : 4 * ty - 1, w0 = UP ? 2 * tx - 1 : 4 * tx - 1;
    // branch-free window load: clamped addresses, out-of-image values zeroed afterwards, so that all NL*NL loads of the
    // window are in flight together (the bounds-checked form issued them one row at a time)
    float4 d[NL][NL];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
      const int hc = min(max(h0 + r, 0), Hx - 1);
#pragma unroll
      for (int s = 0; s < NL; ++s) {
        const int wc = min(max(w0 + s, 0), Wx - 1);
        d[r][s] = xp[((long)hc * Wx + wc) * C4];
      }
    }
    float4 E[NL][6];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
      const int h = h0 + r;
#pragma unroll
      for (int s = 0; s < NL; ++s) {
        const int w = w0 + s;
        float4 v = d[r][s];
        if (affine) {
          v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const bool ok = (unsigned)h < (unsigned)Hx && (unsigned)w < (unsigned)Wx;
        d[r][s] = ok ? v : f4zero();
      }
      if constexpr (UP) w4_in_up(d[r], E[r]); else w4_in6(d[r], E[r]);
    }
    float4* vp = reinterpret_cast<float4*>(V) + t * C4 + c4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (!w4_has<NP>(j)) continue;
      float4 col[NL], o[6];
#pragma unroll
      for (int r = 0; r < NL; ++r) col[r] = E[r][j];
      if constexpr (UP) w4_in_up(col, o); else w4_in6(col, o);
#pragma unroll
      for (int r = 0; r < 6; ++r)
        if (w4_has<NP>(r)) vp[(long)(w4_slot<NP>(r) * NP + w4_slot<NP>(j)) * plane] = o[r];
    }
  }
}

__device__ __forceinline__ void w4_out4(const float4 m[6], float4 y[4]) {
  const float4 p = f4add(m[1], m[2]), q = f4sub(m[1], m[2]), r = f4add(m[3], m[4]), s = f4sub(m[3], m[4]);
  y[0] = f4add(f4add(m[0], p), r);
  y[1] = f4fma(s, 2.f, q);
  y[2] = f4fma(r, 4.f, p);
  y[3] = f4add(f4fma(s, 8.f, q), m[5]);
}
// 2-pixel sums of the above (m[2] is not read):  p0 = m0+2m1+3m3-m4,  p1 = 2m1+12m3-4m4+m5
__device__ __forceinline__ void w4_out_pool(const float4 m[6], float4 y[2]) {
  const float4 a = f4s(m[1], 2.f);
  y[0] = f4sub(f4fma(m[3], 3.f, f4add(m[0], a)), m[4]);
  y[1] = f4add(f4fma(m[4], -4.f, f4fma(m[3], 12.f, a)), m[5]);
}

// one thread per (tile, channel quad).  POOL = 0: y[b, 4ty+a, 4tx+c] = alpha (A^T m A)[a][c] + bias + residual on [B][H][W][C];
// POOL = 1: the 2x2 sums of each tile, y [B][H/2][W/2][C] (alpha = 1/4 makes it the average pool), bias / residual at that size
template <int POOL, int NP>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Mb, const float* __restrict__ bias,
                                                           const float* __restrict__ res, int res_up, float alpha,
                                                           float* __restrict__ y, int B, int H, int W, int C4) {
  static_assert(!POOL || NP == 5, "the pooled output does not read component 2");
  const int th = H >> 2, tw = W >> 2;
  const long T = (long)B * th * tw;
  const long total = T * C4, plane = T * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  constexpr int NO = POOL ? 2 : 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int c4 = (int)(i % C4);
    const long t = i / C4;
    const int tx = (int)(t % tw);
    const long t2 = t / tw;
    const int ty = (int)(t2 % th);
    const long b = t2 / th;
    const float4* mp = reinterpret_cast<const float4*>(Mb) + t * C4 + c4;
    float4 s[NO][6];                     // s[a][j] = (A^T M)[a][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (!w4_has<NP>(j)) {
#pragma unroll
        for (int a = 0; a < NO; ++a) s[a][j] = f4zero();
        continue;
      }
      float4 col[6], yy[4];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = w4_has<NP>(r) ? mp[(long)(w4_slot<NP>(r) * NP + w4_slot<NP>(j)) * plane] : f4zero();
      if constexpr (POOL) w4_out_pool(col, yy); else w4_out4(col, yy);
#pragma unroll
      for (int a = 0; a < NO; ++a) s[a][j] = yy[a];
    }
    float4 bv = f4zero();
    if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * c4);
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
#pragma unroll
    for (int a = 0; a < NO; ++a) {
      float4 o[4];
      if constexpr (POOL) w4_out_pool(s[a], o); else w4_out4(s[a], o);
      const int oy = NO * ty + a;
      const long p0 = ((b * Ho + oy) * Wo + NO * tx) * C4 + c4;
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        float4 v = make_float4(alpha * o[c].x + bv.x, alpha * o[c].y + bv.y, alpha * o[c].z + bv.z, alpha * o[c].w + bv.w);
        if (res) {
          const long rp = (!POOL && res_up == 1) ? ((b * (H >> 1) + (oy >> 1)) * (W >> 1) + ((4 * tx + c) >> 1)) * C4 + c4 : p0 + (long)c * C4;
          const float4 r = reinterpret_cast<const float4*>(res)[rp];
          v = (res_up == 2) ? f4relu_mask(v, r) : f4add(v, r);           // 2: ICG_RES_RELU_MASK
        }
        reinterpret_cast<float4*>(y)[p0 + (long)c * C4] = v;
      }
    }
  }
}

__device__ __forceinline__ void w4_g6(const float g[3], float u[6]) {
  u[0] = 0.25f * g[0];
  u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
  u[2] = -(g[0] - g[1] + g[2]) * (1.f / 6.f);
  u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
  u[5] = g[2];
}

// U[slot(i)*NP + slot(j)][n][k] = (G g G^T)[i][j]
template <int NP>
__device__ __forceinline__ void wino4_weight_body(const float* __restrict__ w, float* __restrict__ U, int N, int K, unsigned blk,
                                                  unsigned nblk) {
  const long total = (long)N * K;
  const long gstride = (long)nblk * blockDim.x;
  for (long i = (long)blk * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int k = (int)(i % K);
    const long n = i / K;
    float t[6][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float g[3], u[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) g[r] = w[((n * 3 + r) * 3 + s) * K + k];
      w4_g6(g, u);
#pragma unroll
      for (int r = 0; r < 6; ++r) t[r][s] = u[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (!w4_has<NP>(r)) continue;
      float u[6];
      w4_g6(t[r], u);
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (w4_has<NP>(j)) U[(long)(w4_slot<NP>(r) * NP + w4_slot<NP>(j)) * total + i] = u[j];
    }
  }
}
template <int NP>
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int N, int K) {
  wino4_weight_body<NP>(w, U, N, K, blockIdx.x, gridDim.x);
}

// all Winograd-domain weight copies of a network in one launch (same arithmetic per element as the single-tensor kernels)
#define ICG_WW_MAX 64
struct WinoWeightPack {
  icg_wino_weight t[ICG_WW_MAX];
  int blk_start[ICG_WW_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void wino_weight_multi_kernel(WinoWeightPack p) {
  int lo = 0, hi = p.n - 1;
  while (lo < hi) {                       // the tensor whose block range holds blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (p.blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const icg_wino_weight& t = p.t[lo];
  const unsigned blk = blockIdx.x - (unsigned)p.blk_start[lo], nblk = (unsigned)(p.blk_start[lo + 1] - p.blk_start[lo]);
  if (t.planes == 36) wino4_weight_body<6>(t.w, t.U, t.N, t.K, blk, nblk);
  else if (t.planes == 25) wino4_weight_body<5>(t.w, t.U, t.N, t.K, blk, nblk);
  else wino_weight_body(t.w, t.U, t.N, t.K, blk, nblk);
}

static void launch_wino4_input(hipStream_t st, int up, int np, const float* x, const float* scale, const float* shift, long ssb,
                               float* V, int B, int H, int W, int Cin, unsigned flags) {
  const long T = (long)B * (H / 4) * (W / 4);
  long nb = icg_cdiv(T * (Cin / 4), 256);
  if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
  const int aff = (flags & ICG_PRE_AFFINE) ? 1 : 0, relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  const dim3 g((unsigned)nb), blk(256);
  if (up) hipLaunchKernelGGL((wino4_input_kernel<1, 5>), g, blk, 0, st, x, scale, shift, ssb, V, B, H, W, Cin / 4, aff, relu);
  else if (np == 5) hipLaunchKernelGGL((wino4_input_kernel<0, 5>), g, blk, 0, st, x, scale, shift, ssb, V, B, H, W, Cin / 4, aff, relu);
  else hipLaunchKernelGGL((wino4_input_kernel<0, 6>), g, blk, 0, st, x, scale, shift, ssb, V, B, H, W, Cin / 4, aff, relu);
}

static void launch_wino4_output(hipStream_t st, int pool, int np, const float* Mb, const float* bias, const float* res, int res_up,
                                float alpha, float* y, int B, int H, int W, int Cout) {
  const long T = (long)B * (H / 4) * (W / 4);
  long nb = icg_cdiv(T * (Cout / 4), 256);
  if (nb > ICG_GRID_CAP) nb = ICG_GRID_CAP;
  const dim3 g((unsigned)nb), blk(256);
  if (pool) hipLaunchKernelGGL((wino4_output_kernel<1, 5>), g, blk, 0, st, Mb, bias, res, res_up, alpha, y, B, H, W, Cout / 4);
  else if (np == 5) hipLaunchKernelGGL((wino4_output_kernel<0, 5>), g, blk, 0, st, Mb, bias, res, res_up, alpha, y, B, H, W, Cout / 4);
  else hipLaunchKernelGGL((wino4_output_kernel<0, 6>), g, blk, 0, st, Mb, bias, res, res_up, alpha, y, B, H, W, Cout / 4);
}

extern "C" int icg_wino4_weight_transform(const float* w, float* U, int N, int K, void* stream) {
  ICG_REQUIRE(w && U && N > 0 && K > 0);
  long blocks = icg_cdiv((long)N * K, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((wino4_weight_kernel<6>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, U, N, K);
  return icg_check_launch();
}

extern "C" int icg_wino4r_weight_transform(const float* w, float* U, int N, int K, void* stream) {
  ICG_REQUIRE(w && U && N > 0 && K > 0);
  long blocks = icg_cdiv((long)N * K, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((wino4_weight_kernel<5>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, U, N, K);
  return icg_check_launch();
}

extern "C" int icg_wino_weight_transform_multi(const icg_wino_weight* items, int n, void* stream) {
  ICG_REQUIRE(items && n > 0);
  for (int base = 0; base < n; base += ICG_WW_MAX) {
    WinoWeightPack p;
    p.n = (n - base < ICG_WW_MAX) ? n - base : ICG_WW_MAX;
    int blocks = 0;
    for (int i = 0; i < p.n; ++i) {
      const icg_wino_weight& t = items[base + i];
      ICG_REQUIRE(t.w && t.U && t.N > 0 && t.K > 0 && (t.planes == 16 || t.planes == 25 || t.planes == 36));
      p.t[i] = t;
      p.blk_start[i] = blocks;
      long b = icg_cdiv((long)t.N * t.K, 256);
      if (b > 4096) b = 4096;
      blocks += (int)b;
    }
    p.blk_start[p.n] = blocks;
    hipLaunchKernelGGL(wino_weight_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  }
  return icg_check_launch();
}

// room for the fragment-major weight copy of the fused narrow-layer kernel (fwino.hip), behind the V and M regions: present for
// every shape that kernel can take, whether or not a given call routes to it (the size is a pure function of the shape)
static size_t fwino_ws_extra(int planes, int H, int W, int Cin, int Cout) {
  return (Cin % 32 == 0 && Cout % 96 == 0 && H % 16 == 0 && W % 16 == 0) ? (size_t)planes * Cin * Cout * sizeof(float) : 0;
}

extern "C" size_t icg_conv2d_wino4_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const size_t T = (size_t)B * (H / 4) * (W / 4);
  return 36 * T * ((size_t)Cin + (size_t)Cout) * sizeof(float) + fwino_ws_extra(36, H, W, Cin, Cout);
}

// fwino.hip: the fused kernel for the narrow layers
int icg_fwino_run(const float* x, int in_up, const float* Uf, const float* bias, const float* residual, int res_mode, float* out,
                  int out_pool, const float* scale, const float* shift, int64_t ssb, int B, int H, int W, int Cin, int Cout,
                  unsigned flags, float alpha, int np, float* V, void* stream);

// shared driver: V = input transform, M[xi] = V[xi] U[xi]^T (np*np batched GEMMs), output transform
static int wino4_run(const float* x, int in_up, const float* U, const float* bias, const float* residual, int res_up, float* out,
                     int out_pool, const float* scale, const float* shift, int64_t ssb, int B, int H, int W, int Cin, int Cout,
                     unsigned flags, float alpha, int np, void* workspace, void* stream) {
  // (splitting the batch into passes whose V + M fit the 256 MB Infinity Cache was measured and is slower: 4.1 -> 5.3 ms on
  // the 96-channel 256x256 layer; written lines are not served back from that cache)
  const long T = (long)B * (H / 4) * (W / 4);
  ICG_REQUIRE(T * 36 < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  float* V = (float*)workspace;
  float* Mb = V + (long)np * np * T * Cin;
  const bool keep_v = (flags & ICG_WINO_KEEP_V) != 0;
  // (measured, tools/fwino_bench.py -> profiles/r04_fwino_microbench.txt: the fused kernel wins 1.1 - 1.7x everywhere except the
  // upsample-on-read 25-plane form at 192 -> 192 channels (the data gradient of DBlock 1's pooled conv2): 0.94x, left on the composite)
  const bool fused_wins = !(np == 5 && in_up && Cin == 192 && Cout == 192) || getenv("ICG_FWINO_ALL");
  if (fused_wins && icg_fwino_applies(B, H, W, Cin, Cout) && (!keep_v || (double)np * np * T * Cin * 4.0 < 4294967296.0)) {
    // narrow layer: one fused kernel (fwino.hip).  The fragment-major copy of U goes behind the (unused) M region
    // (fwino_ws_extra); V is written only when the caller keeps it for the weight gradient.
    float* Uf = Mb + (long)np * np * T * Cout;
    int rc = icg_fwino_pack_weights(U, Uf, np * np, Cin, Cout, stream);
    if (rc != ICG_OK) return rc;
    return icg_fwino_run(x, in_up, Uf, bias, residual, res_up, out, out_pool, scale, shift, ssb, B, H, W, Cin, Cout, flags, alpha,
                         np, keep_v ? V : nullptr, stream);
  }
  launch_wino4_input(st, in_up, np, x, scale, shift, (long)ssb, V, B, H, W, Cin, flags);
  int rc;
  { PlanesScope ps(stream, np * np, (double)T, Cout, Cin); rc = icg_gemm_batched(V, U, Mb, (int)T, Cout, Cin, 0, 1, T * Cin, (long)Cout * Cin, T * Cout, np * np, 1.0f, stream); }
  if (rc != ICG_OK) return rc;
  launch_wino4_output(st, out_pool, np, Mb, bias, residual, res_up, alpha, out, B, H, W, Cout);
  return icg_check_launch();
}

extern "C" int icg_conv2d_wino4_fprop(const float* x, const float* U, const float* bias, const float* residual, float* out,
                                      const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W,
                                      int Cin, int Cout, unsigned flags, float alpha, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && U && out && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE((H % 4 == 0) && (W % 4 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && !(flags & ICG_UPSAMPLE2X));
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift && (ss_bstride % 4 == 0));
  if (workspace_bytes < icg_conv2d_wino4_workspace_bytes(B, H, W, Cin, Cout)) return ICG_ERR_WORKSPACE;
  if (flags & ICG_RES_RELU_MASK) ICG_REQUIRE(residual && !(flags & ICG_RES_UPSAMPLE2X));
  return wino4_run(x, 0, U, bias, residual, icg_res_mode(flags), out, 0, scale, shift, ss_bstride, B, H, W, Cin,
                   Cout, flags, alpha, 6, workspace, stream);
}

// ---- resample-fused layers in the 25-plane domain (H, W below are always the FULL resolution of the layer) -----------------
extern "C" size_t icg_conv2d_rs_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const size_t T = (size_t)B * (H / 4) * (W / 4);
  return 25 * T * ((size_t)Cin + (size_t)Cout) * sizeof(float) + fwino_ws_extra(25, H, W, Cin, Cout);
}

#define ICG_RS_REQUIRE(B, Hl, Wl, Cin, Cout)                                                                           \
  ICG_REQUIRE(B > 0 && Hl > 0 && Wl > 0 && Cin > 0 && Cout > 0 && (Hl % 2 == 0) && (Wl % 2 == 0) && (Cin % 4 == 0) && \
              (Cout % 4 == 0))

// out [B][2Hs][2Ws][Cout] = conv3x3(upsample2(act(x))) + bias,  x [B][Hs][Ws][Cin],  U = icg_wino4r_weight_transform(w_ohwi)
extern "C" int icg_conv2d_up_wino_fprop(const float* x, const float* U, const float* bias, float* out, const float* scale,
                                        const float* shift, int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout,
                                        unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && U && out && workspace);
  ICG_RS_REQUIRE(B, Hs, Ws, Cin, Cout);
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift && (ss_bstride % 4 == 0));
  if (workspace_bytes < icg_conv2d_rs_wino_workspace_bytes(B, 2 * Hs, 2 * Ws, Cin, Cout)) return ICG_ERR_WORKSPACE;
  return wino4_run(x, 1, U, bias, nullptr, 0, out, 0, scale, shift, ss_bstride, B, 2 * Hs, 2 * Ws, Cin, Cout, flags, 1.0f, 5,
                   workspace, stream);
}

// da [B][Hs][Ws][Cin] = sum-pool2(conv3x3^T(dy)),  dy [B][2Hs][2Ws][Cout],  U = icg_wino4r_weight_transform(w_dgrad)
extern "C" int icg_conv2d_up_wino_dgrad(const float* dy, const float* U, float* da, int B, int Hs, int Ws, int Cin, int Cout,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dy && U && da && workspace);
  ICG_RS_REQUIRE(B, Hs, Ws, Cin, Cout);
  if (workspace_bytes < icg_conv2d_rs_wino_workspace_bytes(B, 2 * Hs, 2 * Ws, Cout, Cin)) return ICG_ERR_WORKSPACE;
  return wino4_run(dy, 0, U, nullptr, nullptr, 0, da, 1, nullptr, nullptr, 0, B, 2 * Hs, 2 * Ws, Cout, Cin, 0, 1.0f, 5, workspace,
                   stream);
}

// out [B][Hp][Wp][Cout] = avgpool2(conv3x3(act(x))) + bias + residual,  x [B][2Hp][2Wp][Cin]
extern "C" int icg_conv2d_down_wino_fprop(const float* x, const float* U, const float* bias, const float* residual, float* out,
                                          int B, int Hp, int Wp, int Cin, int Cout, unsigned flags, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && U && out && workspace && !(flags & ~(ICG_PRE_RELU | ICG_WINO_KEEP_V)));
  ICG_RS_REQUIRE(B, Hp, Wp, Cin, Cout);
  if (workspace_bytes < icg_conv2d_rs_wino_workspace_bytes(B, 2 * Hp, 2 * Wp, Cin, Cout)) return ICG_ERR_WORKSPACE;
  return wino4_run(x, 0, U, bias, residual, 0, out, 1, nullptr, nullptr, 0, B, 2 * Hp, 2 * Wp, Cin, Cout, flags, 0.25f, 5,
                   workspace, stream);
}

// da [B][2Hp][2Wp][Cin] = conv3x3^T(upsample2(dy) / 4),  dy [B][Hp][Wp][Cout]
extern "C" int icg_conv2d_down_wino_dgrad(const float* dy, const float* U, float* da, int B, int Hp, int Wp, int Cin, int Cout,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dy && U && da && workspace);
  ICG_RS_REQUIRE(B, Hp, Wp, Cin, Cout);
  if (workspace_bytes < icg_conv2d_rs_wino_workspace_bytes(B, 2 * Hp, 2 * Wp, Cout, Cin)) return ICG_ERR_WORKSPACE;
  return wino4_run(dy, 1, U, nullptr, nullptr, 0, da, 0, nullptr, nullptr, 0, B, 2 * Hp, 2 * Wp, Cout, Cin, 0, 0.25f, 5, workspace,
                   stream);
}

extern "C" int icg_conv2d_down_wino_dgrad_relu(const float* dy, const float* U, const float* relu_in, float* dx, int B, int Hp,
                                               int Wp, int Cin, int Cout, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(dy && U && relu_in && dx && workspace);
  ICG_RS_REQUIRE(B, Hp, Wp, Cin, Cout);
  if (workspace_bytes < icg_conv2d_rs_wino_workspace_bytes(B, 2 * Hp, 2 * Wp, Cout, Cin)) return ICG_ERR_WORKSPACE;
  return wino4_run(dy, 1, U, nullptr, relu_in, 2, dx, 0, nullptr, nullptr, 0, B, 2 * Hp, 2 * Wp, Cout, Cin, 0, 0.25f, 5, workspace,
                   stream);
}

// ---- weight gradient through the F(4x4,3x3) domain: dw = G^T [ sum_tiles (A dy A^T) .* (B^T act(x) B) ] G --------------------
// (the adjoint of the forward form in g: 36 [Cin x tiles] x [tiles x Cout] GEMMs, 1/4 of the direct multiply-adds, both
// transform passes over 2.25x the activation volume)
//   1-D  s = A d : s0 = d0, s1 = d0+d1+d2+d3, s2 = d0-d1+d2-d3, s3 = d0+2d1+4d2+8d3, s4 = d0-2d1+4d2-8d3, s5 = d3
//   1-D  g = G^T u : g0 = u0/4-(u1+u2)/6+(u3+u4)/24, g1 = (u2-u1)/6+(u3-u4)/12, g2 = (u3+u4-u1-u2)/6+u5
__device__ __forceinline__ void w4_dy6(const float4 d[4], float4 s[6]) {
  const float4 e = f4add(d[0], d[2]), o = f4add(d[1], d[3]);
  s[0] = d[0];
  s[1] = f4add(e, o);
  s[2] = f4sub(e, o);
  const float4 e4 = f4fma(d[2], 4.f, d[0]), o4 = f4fma(d[3], 8.f, f4s(d[1], 2.f));
  s[3] = f4add(e4, o4);
  s[4] = f4sub(e4, o4);
  s[5] = d[3];
}
// the same for the upsampled tile d = [l0 l0 l1 l1]:  s0 = l0, s1 = 2l0+2l1, s2 = 0, s3 = 3l0+12l1, s4 = -l0-4l1, s5 = l1
__device__ __forceinline__ void w4_dy_up(const float4 l[2], float4 s[6]) {
  s[0] = l[0];
  s[1] = f4s(f4add(l[0], l[1]), 2.f);
  s[2] = f4zero();
  const float4 q = f4fma(l[1], 4.f, l[0]);
  s[3] = f4s(q, 3.f);
  s[4] = f4s(q, -1.f);
  s[5] = l[1];
}

// DY[slot(i)*NP + slot(j)][t][c] = alpha (A d A^T)[i][j], one thread per (tile, channel quad);  UP = 0: d = the 4x4 tile of
// dy [B][H][W][C];  UP = 1: d = the tile of the nearest-x2 upsampled dy [B][H/2][W/2][C] (2x2 loads)
// db_part != NULL: the pass also produces the bias gradient's column sums of (unscaled) dy -- every pixel of dy is read exactly
// once here, so the separate icg_colsum pass over dy (6.4 ms per cfg3 step in round 1) disappears.  Deterministic: per-thread
// sums go through LDS, channel quad q is reduced by thread q over the block's threads in index order, and each block writes
// one row [4*C4] of partials that wino_db_final_kernel sums in block order.
template <int UP, int NP>
__global__ __launch_bounds__(256) void wino4_dy_kernel(const float* __restrict__ dy, float* __restrict__ DY, int B, int H,
                                                       int W, int C4, float alpha, float* __restrict__ db_part) {
  static_assert(!UP || NP == 5, "the upsampled tile has no component 2");
  __shared__ float4 red[256];
  const int th = H >> 2, tw = W >> 2;
  const long T = (long)B * th * tw;
  const long total = T * C4, plane = T * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  constexpr int NL = UP ? 2 : 4;
  const int Hx = UP ? (H >> 1) : H, Wx = UP ? (W >> 1) : W;
  float4 bsum[2] = {f4zero(), f4zero()};                      // running sums of channel quads tid and tid + 256 (C <= 2048)
  for (long base = (long)blockIdx.x * blockDim.x; base < total; base += gstride) {
    const long i = base + threadIdx.x;
    float4 tsum = f4zero();
    if (i < total) {
      const int c4 = (int)(i % C4);
      const long t = i / C4;
      const int tx = (int)(t % tw);
      const long t2 = t / tw;
      const int ty = (int)(t2 % th);
      const long b = t2 / th;
      const float4* gp = reinterpret_cast<const float4*>(dy) + ((b * Hx + NL * ty) * Wx + NL * tx) * C4 + c4;
      float4 E[NL][6];
#pragma unroll
      for (int r = 0; r < NL; ++r) {
        float4 d[NL];
#pragma unroll
        for (int c = 0; c < NL; ++c) {
          const float4 raw = gp[((long)r * Wx + c) * C4];
          tsum = f4add(tsum, raw);
          d[c] = f4s(raw, alpha);
        }
        if constexpr (UP) w4_dy_up(d, E[r]); else w4_dy6(d, E[r]);
      }
      float4* op = reinterpret_cast<float4*>(DY) + t * C4 + c4;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (!w4_has<NP>(j)) continue;
        float4 col[NL], o[6];
#pragma unroll
        for (int r = 0; r < NL; ++r) col[r] = E[r][j];
        if constexpr (UP) w4_dy_up(col, o); else w4_dy6(col, o);
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if (w4_has<NP>(r)) op[(long)(w4_slot<NP>(r) * NP + w4_slot<NP>(j)) * plane] = o[r];
      }
    }
    if (db_part) {                      // (uniform over the block: every thread takes the same number of iterations)
      red[threadIdx.x] = tsum;
      __syncthreads();
      const int off = (int)(base % C4);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q = (int)threadIdx.x + 256 * h;
        if (q < C4) {
          float4 a = bsum[h];
          for (int j = (q - off + C4) % C4; j < 256; j += C4) a = f4add(a, red[j]);
          bsum[h] = a;
        }
      }
      __syncthreads();
    }
  }
  if (db_part) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = (int)threadIdx.x + 256 * h;
      if (q < C4) reinterpret_cast<float4*>(db_part)[(long)blockIdx.x * C4 + q] = bsum[h];
    }
  }
}

// out[y][c] = sum of the partial rows [y * rows_per_y, (y + 1) * rows_per_y) of part[.][c]: block = 32 channels x 32 row slices
// (128-byte row segments, fixed order), folded through LDS.  The bias gradient runs it twice -- the wino4_dy pass leaves up to
// T * C / 1024 partial rows (24576 x 96 at 256^2: 9.4 MB) -- WINO_DB_ROWS2 row groups first, then those rows into dbias.
#define WINO_DB_ROWS2 64
__global__ __launch_bounds__(1024) void wino_db_final_kernel(const float* __restrict__ part, int nblocks, int rows_per_y, int C,
                                                             float* __restrict__ out) {
  __shared__ double red[32][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int k0 = blockIdx.y * rows_per_y, k1 = min(nblocks, k0 + rows_per_y);
  double a = 0.0;
  if (c < C)
    for (int k = k0 + sl; k < k1; k += 32) a += (double)part[(long)k * C + c];
  red[sl][cl] = a;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 32; ++k) a += red[k][cl];
    out[(long)blockIdx.y * C + c] = (float)a;
  }
}

__device__ __forceinline__ void w4_gt3(const float u[6], float g[3]) {
  const float p = u[1] + u[2], q = u[3] + u[4];
  g[0] = 0.25f * u[0] - p * (1.f / 6.f) + q * (1.f / 24.f);
  g[1] = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
  g[2] = (q - p) * (1.f / 6.f) + u[5];
}

// dw[r][s][ci][co] = (G^T dU G)[r][s]   (HWIO, like the direct weight gradient); NP = 5: component 2 of dU is zero
template <int NP>
__global__ __launch_bounds__(256) void wino4_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, long n) {
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gstride) {
    float t[3][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float u[6], g[3];
#pragma unroll
      for (int a = 0; a < 6; ++a)
        u[a] = (w4_has<NP>(a) && w4_has<NP>(b)) ? dU[(long)(w4_slot<NP>(a) * NP + w4_slot<NP>(b)) * n + i] : 0.f;
      w4_gt3(u, g);
      t[0][b] = g[0]; t[1][b] = g[1]; t[2][b] = g[2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float g[3];
      w4_gt3(t[r], g);
      dw[(long)(3 * r + 0) * n + i] = g[0];
      dw[(long)(3 * r + 1) * n + i] = g[1];
      dw[(long)(3 * r + 2) * n + i] = g[2];
    }
  }
}

static size_t wino4_wgrad_bytes(int np, int B, int H, int W, int Cin, int Cout) {
  const size_t T = (size_t)B * (H / 4) * (W / 4), P = (size_t)np * np;
  return wino_al(P * T * Cin * sizeof(float)) + wino_al(P * T * Cout * sizeof(float)) + wino_al(P * Cin * Cout * sizeof(float)) +
         wino_al(icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)T, (int)P));
}

// shared driver: V = input transform of x (x_up: of the upsampled x), DY = transform of dy (dy_up: of the upsampled dy, scaled),
// dU[xi] = V[xi]^T DY[xi], dw = G^T dU G.  H, W: full resolution.
static long wino4_dy_blocks(long T, int Cout) {
  long nb = icg_cdiv(T * (Cout / 4), 256);
  return nb > ICG_GRID_CAP ? ICG_GRID_CAP : nb;
}

static int wino4_wgrad_run(const float* x, int x_up, const float* dy, int dy_up, float dy_alpha, float* dw, const float* scale,
                           const float* shift, int64_t ssb, int B, int H, int W, int Cin, int Cout, unsigned flags, int np,
                           void* workspace, void* stream, const float* v_saved = nullptr, float* dbias = nullptr,
                           float* db_part = nullptr) {
  const long T = (long)B * (H / 4) * (W / 4), P = (long)np * np;
  ICG_REQUIRE(T * 36 < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  float* V = (float*)base;
  if (x) base += wino_al(P * T * Cin * sizeof(float));          // (no V region when the caller supplies it)
  float* DY = (float*)base;                   base += wino_al(P * T * Cout * sizeof(float));
  float* dU = (float*)base;                   base += wino_al((size_t)P * Cin * Cout * sizeof(float));
  void* gws = base;
  const size_t gws_bytes = icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)T, (int)P);
  if (x) launch_wino4_input(st, x_up, np, x, scale, shift, (long)ssb, V, B, H, W, Cin, flags);
  else V = const_cast<float*>(v_saved);        // the forward pass's V, kept by the caller (icg_conv2d_wino4_wgrad_from_v)
  long nb = wino4_dy_blocks(T, Cout);
  const dim3 g((unsigned)nb), blk(256);
  if (dy_up) hipLaunchKernelGGL((wino4_dy_kernel<1, 5>), g, blk, 0, st, dy, DY, B, H, W, Cout / 4, dy_alpha, db_part);
  else if (np == 5) hipLaunchKernelGGL((wino4_dy_kernel<0, 5>), g, blk, 0, st, dy, DY, B, H, W, Cout / 4, dy_alpha, db_part);
  else hipLaunchKernelGGL((wino4_dy_kernel<0, 6>), g, blk, 0, st, dy, DY, B, H, W, Cout / 4, dy_alpha, db_part);
  if (db_part) {
    const dim3 cg((unsigned)icg_cdiv(Cout, 32));
    if (nb <= 32 * 8) {
      hipLaunchKernelGGL(wino_db_final_kernel, cg, dim3(1024), 0, st, (const float*)db_part, (int)nb, (int)nb, Cout, dbias);
    } else {                                   // two levels: the second level's rows sit behind the nb partial rows
      float* part2 = db_part + nb * Cout;
      const int rpy = (int)icg_cdiv(nb, WINO_DB_ROWS2), ny = (int)icg_cdiv(nb, rpy);
      hipLaunchKernelGGL(wino_db_final_kernel, dim3(cg.x, (unsigned)ny), dim3(1024), 0, st, (const float*)db_part, (int)nb, rpy, Cout, part2);
      hipLaunchKernelGGL(wino_db_final_kernel, cg, dim3(1024), 0, st, (const float*)part2, ny, ny, Cout, dbias);
    }
  }
  int rc;
  { PlanesScope ps(stream, (int)P, Cin, Cout, (double)T, 1); rc = icg_gemm_tn_batched(V, DY, dU, Cin, Cout, (int)T, T * Cin, T * Cout, (long)Cin * Cout, (int)P, gws, gws_bytes, stream); }
  if (rc != ICG_OK) return rc;
  const long n = (long)Cin * Cout;
  nb = icg_cdiv(n, 256);
  if (nb > 4096) nb = 4096;
  if (np == 5) hipLaunchKernelGGL((wino4_dw_kernel<5>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)dU, dw, n);
  else hipLaunchKernelGGL((wino4_dw_kernel<6>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)dU, dw, n);
  return icg_check_launch();
}

extern "C" size_t icg_conv2d_wino4_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  return wino4_wgrad_bytes(6, B, H, W, Cin, Cout);
}

extern "C" int icg_conv2d_wino4_wgrad(const float* x, const float* dy, float* dw, const float* scale, const float* shift,
                                      int64_t ss_bstride, int B, int H, int W, int Cin, int Cout, unsigned flags,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE((H % 4 == 0) && (W % 4 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && !(flags & ICG_UPSAMPLE2X));
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift && (ss_bstride % 4 == 0));
  if (workspace_bytes < icg_conv2d_wino4_wgrad_workspace_bytes(B, H, W, Cin, Cout)) return ICG_ERR_WORKSPACE;
  return wino4_wgrad_run(x, 0, dy, 0, 1.0f, dw, scale, shift, ss_bstride, B, H, W, Cin, Cout, flags, 6, workspace, stream);
}

extern "C" size_t icg_conv2d_rs_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  return wino4_wgrad_bytes(5, B, H, W, Cin, Cout);
}

// dw [3][3][Cin][Cout] of out = conv3x3(upsample2(act(x))):  x [B][Hs][Ws][Cin], dy [B][2Hs][2Ws][Cout]
extern "C" int icg_conv2d_up_wino_wgrad(const float* x, const float* dy, float* dw, const float* scale, const float* shift,
                                        int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout, unsigned flags,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw && workspace);
  ICG_RS_REQUIRE(B, Hs, Ws, Cin, Cout);
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift && (ss_bstride % 4 == 0));
  if (workspace_bytes < icg_conv2d_rs_wino_wgrad_workspace_bytes(B, 2 * Hs, 2 * Ws, Cin, Cout)) return ICG_ERR_WORKSPACE;
  return wino4_wgrad_run(x, 1, dy, 0, 1.0f, dw, scale, shift, ss_bstride, B, 2 * Hs, 2 * Ws, Cin, Cout, flags, 5, workspace, stream);
}

// dw [3][3][Cin][Cout] of out = avgpool2(conv3x3(act(x))):  x [B][2Hp][2Wp][Cin], dy [B][Hp][Wp][Cout]
extern "C" int icg_conv2d_down_wino_wgrad(const float* x, const float* dy, float* dw, int B, int Hp, int Wp, int Cin, int Cout,
                                          unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw && workspace && !(flags & ~ICG_PRE_RELU));
  ICG_RS_REQUIRE(B, Hp, Wp, Cin, Cout);
  if (workspace_bytes < icg_conv2d_rs_wino_wgrad_workspace_bytes(B, 2 * Hp, 2 * Wp, Cin, Cout)) return ICG_ERR_WORKSPACE;
  return wino4_wgrad_run(x, 0, dy, 1, 0.25f, dw, nullptr, nullptr, 0, B, 2 * Hp, 2 * Wp, Cin, Cout, flags, 5, workspace, stream);
}

// Weight gradient from the V planes the FORWARD pass of the same layer computed (icg_conv2d_wino4_fprop and the *_wino_fprop
// entries leave V = transform(act(x)) in the first planes * T * Cin floats of their workspace, T = B * H/4 * W/4 at the full
// resolution H x W): a caller that keeps that region alive until the backward pass (2.25x / 1.56x the activation; sized for
// 288 GB) skips the input transform here.  planes = 36 (plain 3x3) or 25 (resample-fused); dy_up = 1 with dy_alpha = 0.25
// for the avgpool-fused layer (dy at the pooled resolution), else 0 / 1.
extern "C" size_t icg_conv2d_wino4_wgrad_from_v_workspace_bytes(int B, int H, int W, int Cin, int Cout, int planes) {
  const size_t T = (size_t)B * (H / 4) * (W / 4), P = (size_t)planes;
  return wino_al(P * T * Cout * sizeof(float)) + wino_al(P * Cin * Cout * sizeof(float)) +
         wino_al(icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)T, planes));
}

// the same, and dbias [Cout] = column sums of dy (the gradient of the layer's bias) from the pass that reads dy anyway
extern "C" size_t icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes(int B, int H, int W, int Cin, int Cout, int planes) {
  const long T = (long)B * (H / 4) * (W / 4);
  return icg_conv2d_wino4_wgrad_from_v_workspace_bytes(B, H, W, Cin, Cout, planes) +
         wino_al((size_t)(wino4_dy_blocks(T, Cout) + WINO_DB_ROWS2) * Cout * sizeof(float));
}

extern "C" int icg_conv2d_wino4_wgrad_from_v_db(const float* V, const float* dy, float* dw, float* dbias, int B, int H, int W,
                                                int Cin, int Cout, int planes, int dy_up, float dy_alpha, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(V && dy && dw && dbias && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE((H % 4 == 0) && (W % 4 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && (planes == 36 || planes == 25));
  ICG_REQUIRE((!dy_up || planes == 25) && Cout <= 2048);
  if (workspace_bytes < icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes(B, H, W, Cin, Cout, planes)) return ICG_ERR_WORKSPACE;
  float* part = (float*)((char*)workspace + icg_conv2d_wino4_wgrad_from_v_workspace_bytes(B, H, W, Cin, Cout, planes));
  return wino4_wgrad_run(nullptr, 0, dy, dy_up ? 1 : 0, dy_alpha, dw, nullptr, nullptr, 0, B, H, W, Cin, Cout, 0,
                         planes == 25 ? 5 : 6, workspace, stream, V, dbias, part);
}

extern "C" int icg_conv2d_wino4_wgrad_from_v(const float* V, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                             int planes, int dy_up, float dy_alpha, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  ICG_REQUIRE(V && dy && dw && workspace && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE((H % 4 == 0) && (W % 4 == 0) && (Cin % 4 == 0) && (Cout % 4 == 0) && (planes == 36 || planes == 25));
  ICG_REQUIRE(!dy_up || planes == 25);
  if (workspace_bytes < icg_conv2d_wino4_wgrad_from_v_workspace_bytes(B, H, W, Cin, Cout, planes)) return ICG_ERR_WORKSPACE;
  return wino4_wgrad_run(nullptr, 0, dy, dy_up ? 1 : 0, dy_alpha, dw, nullptr, nullptr, 0, B, H, W, Cin, Cout, 0,
                         planes == 25 ? 5 : 6, workspace, stream, V);
}
