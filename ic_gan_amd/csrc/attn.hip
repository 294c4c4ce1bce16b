// Attention scores with the softmax in the epilogue: beta[b][i][:] = softmax_j( sum_k theta[b][i][k] phi[b][j][k] )
// (layers.py:233-238: beta = F.softmax(torch.bmm(theta^T, phi), -1); theta [B][n][d] and phi [B][m][d] are the NHWC outputs of the
// two 1x1 projections, m = n / 4 after the 2x2 max-pool).
//
// The three-kernel form writes the scores s [B][n][m] (1 GiB at B = 64, n = 4096, m = 1024), reads them back for the softmax and
// writes beta: 3 passes over that tensor for 26 GFLOP of arithmetic.  Here a workgroup owns 32 query rows and ALL m keys: wave w
// holds the 32 x 128 scores of keys [128 w, 128 w + 128) in 64 accumulator registers (exact-fp32 v_mfma_f32_16x16x4_f32, operand
// fragments loaded straight from L2 -- d is 24 or 48, there is nothing to stage), the row maxima / sums are combined by two
// wave shuffles and one LDS exchange between the waves, and beta is written once.
// A lane's V consecutive k-values are the operands of V successive MFMAs (the K order is permuted identically in both operands,
// as in pgemm.hip), so a fragment is one 8- or 16-byte load; operands are swapped (D = phi-fragment x theta-fragment), which leaves
// lane (r, kk) with beta[row r][key 4 kk .. 4 kk + 3]: 16-byte stores.
#include "icg_common.h"
#include <math.h>
#include <stdlib.h>

typedef float at_f32x4 __attribute__((ext_vector_type(4)));

template <int V> struct at_vec;
template <> struct at_vec<4> { typedef float4 type; };
template <> struct at_vec<2> { typedef float2 type; };
__device__ __forceinline__ float at_get(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }
__device__ __forceinline__ float at_get(const float2& v, int s) { return s == 0 ? v.x : v.y; }

// exp(x) for x <= 0 as 2^(x log2 e) on the hardware exponential (v_exp_f32, 1 ulp): the argument's rounding adds |x| 2^-24 relative
// error -- < 2e-6 down to the e^-30 terms that still register in a float sum -- against ~25 VALU instructions for expf(); with 64
// exponentials per lane the accurate form was half of this kernel's issue time
__device__ __forceinline__ float at_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// V: k-values per lane and K-tile (a K-tile is 4 V wide); NKT: K-tiles (d = 4 V NKT); waves = m / 128 (blockDim = 64 waves)
constexpr int AT_SROW = 136;       // floats per staged row of 128 keys: 544 bytes = 32 mod 256
template <int V, int NKT, bool STAGE>
__global__ __launch_bounds__(512, 4) void icg_attn_scores_softmax_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                                      float* __restrict__ beta, int n, int m) {
  typedef typename at_vec<V>::type vec;
  constexpr int D = 4 * V * NKT;
  __shared__ float red[2][8][32];
  __shared__ __attribute__((aligned(16))) float stage[STAGE ? 8 * 16 * AT_SROW : 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int rows_per_img = n >> 5;
  const int b = blockIdx.x / rows_per_img, row0 = (blockIdx.x - b * rows_per_img) << 5;
  const float* __restrict__ Q = theta + ((long)b * n + row0) * D;
  const float* __restrict__ Kp = phi + ((long)b * m + 128 * wv) * D;

  vec qa[2][NKT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NKT; ++t) qa[i][t] = *reinterpret_cast<const vec*>(Q + (16 * i + r) * D + 4 * V * t + V * kk);

  at_f32x4 acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    vec kb[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) kb[t] = *reinterpret_cast<const vec*>(Kp + (16 * j + r) * D + 4 * V * t + V * kk);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      at_f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int s = 0; s < V; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(at_get(kb[t], s), at_get(qa[i][t], s), a, 0, 0, 0);
      acc[i][j] = a;
    }
    if (j & 1) asm volatile("" ::: "memory");       // keeps the fragment loads at most two column tiles ahead (128-register budget: 2 workgroups per CU)
  }

  // ---- row maxima: lane -> the 4 lanes of a row (xor 16, 32) -> the waves (LDS)
  float mx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float v = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) v = fmaxf(fmaxf(v, fmaxf(acc[i][j][0], acc[i][j][1])), fmaxf(acc[i][j][2], acc[i][j][3]));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    if (kk == 0) red[0][wv][16 * i + r] = v;
    mx[i] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float v = red[0][0][16 * i + r];
    for (int w = 1; w < nw; ++w) v = fmaxf(v, red[0][w][16 * i + r]);
    mx[i] = v;
  }
  // ---- exponentials and row sums (same path; fixed order over the waves)
  float inv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = at_exp(acc[i][j][e] - mx[i]);
      s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kk == 0) red[1][wv][16 * i + r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = red[1][0][16 * i + r];
    for (int w = 1; w < nw; ++w) s += red[1][w][16 * i + r];
    inv[i] = 1.0f / s;
  }
  if constexpr (STAGE) {
    // beta leaves through LDS: in the accumulator layout a store instruction covers 16 rows x 64 bytes (half a cache line per row);
    // re-laid-out per wave (16 rows x 128 keys, row stride 544 bytes = 32 mod 256: conflict-free 16-byte writes in the operand
    // pattern and conflict-free row-contiguous reads) every store instruction writes two whole 512-byte row segments.
    float* __restrict__ st = stage + wv * (16 * AT_SROW);
    float* __restrict__ out = beta + ((long)b * n + row0) * m + 128 * wv + 4 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<at_f32x4*>(st + r * AT_SROW + 16 * j + 4 * kk) = acc[i][j] * inv[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = 2 * q + (lane >> 5);
        const at_f32x4 v = *reinterpret_cast<const at_f32x4*>(st + row * AT_SROW + 4 * (lane & 31));
        *reinterpret_cast<at_f32x4*>(out + (long)(16 * i + row) * m) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else {
    float* __restrict__ out = beta + ((long)b * n + row0) * m + 128 * wv + 4 * kk;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<at_f32x4*>(out + (long)(16 * i + r) * m + 16 * j) = acc[i][j] * inv[i];
  }
}

// 1 when the fused kernel serves the shape: n a multiple of 32, m a multiple of 128 up to 1024, d in {8, 16, 24, 32, 48, 64}
extern "C" int icg_attn_scores_softmax_applies(int n, int m, int d) {
  if (n < 32 || n % 32 != 0 || m < 128 || m % 128 != 0 || m > 1024) return 0;
  return (d == 8 || d == 16 || d == 24 || d == 32 || d == 48 || d == 64) ? 1 : 0;
}

extern "C" int icg_attn_scores_softmax(const float* theta, const float* phi, float* beta, int B, int n, int m, int d, void* stream) {
  ICG_REQUIRE(theta && phi && beta && B > 0);
  ICG_REQUIRE(icg_attn_scores_softmax_applies(n, m, d));
  ICG_REQUIRE(((uintptr_t)theta % 16) == 0 && ((uintptr_t)phi % 16) == 0 && ((uintptr_t)beta % 16) == 0);
  const long blocks = (long)B * (n / 32);
  ICG_REQUIRE(blocks < 0x7fffffffL);
  const dim3 grid((unsigned)blocks), block((unsigned)(64 * (m / 128)));
  hipStream_t st = (hipStream_t)stream;
  static const bool stage = [] { const char* e = getenv("ICG_ATTN_STAGE"); return !(e && e[0] == '0'); }();
#define AT_LAUNCH(V_, NKT_)                                                                                                  \
  do {                                                                                                                       \
    if (stage) hipLaunchKernelGGL((icg_attn_scores_softmax_kernel<V_, NKT_, true>), grid, block, 0, st, theta, phi, beta, n, m);  \
    else hipLaunchKernelGGL((icg_attn_scores_softmax_kernel<V_, NKT_, false>), grid, block, 0, st, theta, phi, beta, n, m);   \
  } while (0)
  switch (d) {
    case 8: AT_LAUNCH(2, 1); break;
    case 16: AT_LAUNCH(4, 1); break;
    case 24: AT_LAUNCH(2, 3); break;
    case 32: AT_LAUNCH(4, 2); break;
    case 48: AT_LAUNCH(4, 3); break;
    default: AT_LAUNCH(4, 4); break;
  }
#undef AT_LAUNCH
  return icg_check_launch();
}


// =====================================================================================================================
// Backward of the same block: dS = beta .* (dbeta - rowsum(beta .* dbeta)),  dbeta[b][i][j] = sum_c dO[b][i][c] V[b][j][c]
// (autograd of layers.py:237-243: o = bmm(g, beta^T), beta = softmax(scores)).  The three-kernel form writes dbeta [B][n][m]
// (1 - 2 GiB), reads it back together with beta in icg_softmax_bwd and writes dS.  Here the dbeta tile never leaves the
// registers: a workgroup owns 32 query rows and all m keys (wave w: keys [128 w, 128 w + 128), 64 accumulator registers), the
// 32 x dv rows of dO are staged once in LDS (row stride = 32 bytes mod 256: the four ds_read_b128 service groups of the
// 16x16x4 operand pattern then see 16 distinct 16-byte slots), the V fragments come straight from L2 (every CU streams the
// same m x dv matrix of its image), beta is read once with 16-byte loads in the accumulator layout, the row dot products go
// through two wave shuffles and one LDS exchange in a fixed order, dS is written once.
template <int DV, bool STAGE>
__global__ __launch_bounds__(512) void icg_attn_dscores_kernel(const float* __restrict__ dO, const float* __restrict__ Vg,
                                                                const float* __restrict__ beta, float* __restrict__ dS, int n, int m) {
  constexpr int STRIDE = DV + (((DV * 4) % 256 == 128) ? 40 : 8);   // floats per staged row: 544 / 800 bytes = 32 mod 256 for DV = 96 / 192
  static_assert((DV % 16) == 0 && ((STRIDE * 4) % 256) == 32, "LDS row stride must be 32 bytes mod 256");
  __shared__ __attribute__((aligned(16))) float qs[32 * STRIDE];
  __shared__ float red[8][32];
  // STAGE (round 6): beta arrives and dS leaves through a wave-private LDS tile (16 rows x 128 keys, as in icg_attn_scores_softmax_kernel): the
  // global accesses are whole 512-byte row segments instead of 64-byte pieces of 16 rows (in the accumulator layout this kernel moved its 2 x 128 KB
  // per workgroup at 1.6 - 2.4 TB/s)
  __shared__ __attribute__((aligned(16))) float stage[STAGE ? 8 * 16 * AT_SROW : 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int rows_per_img = n >> 5;
  const int b = blockIdx.x / rows_per_img, row0 = (blockIdx.x - b * rows_per_img) << 5;
  const float* __restrict__ Q = dO + ((long)b * n + row0) * DV;
  const float* __restrict__ Kp = Vg + ((long)b * m + 128 * wv) * DV;

  // stage the 32 x DV rows of dO (float4 per thread-iteration)
  for (int i = tid; i < 32 * (DV / 4); i += blockDim.x) {
    const int row = i / (DV / 4), c4 = i - row * (DV / 4);
    *reinterpret_cast<float4*>(qs + row * STRIDE + 4 * c4) = *reinterpret_cast<const float4*>(Q + row * DV + 4 * c4);
  }
  __syncthreads();

  at_f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = at_f32x4{0.f, 0.f, 0.f, 0.f};
  // K-tiles of 16 channels (a lane's 4 consecutive channels feed 4 MFMAs); the V fragments of the next K-tile are loaded before the
  // 64 MFMAs of the current one are issued, and beta -- the epilogue's operand, independent of the GEMM -- during the last K-tile
  // (one workgroup per CU: 64 accumulator + 64 V-fragment + 64 beta registers; the first version loaded and consumed each K-tile
  // in turn and ran at 70 TF)
  constexpr int NKT = DV / 16;
  const long off = ((long)b * n + row0) * m + 128 * wv + 4 * kk;
  const long offr = ((long)b * n + row0) * m + 128 * wv + 4 * (lane & 31);      // (STAGE: the row-contiguous view of the same 32 x 128 block)
  at_f32x4 bt[2][8];
  float4 kb[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) kb[0][j] = *reinterpret_cast<const float4*>(Kp + (16 * j + r) * DV + 4 * kk);
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (t + 1 < NKT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) kb[(t + 1) & 1][j] = *reinterpret_cast<const float4*>(Kp + (16 * j + r) * DV + 16 * (t + 1) + 4 * kk);
    } else if constexpr (STAGE) {      // row-contiguous: instruction q of group i brings rows 2 q, 2 q + 1 (lane >> 5), 16 bytes at key 4 (lane & 31)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q)
          bt[i][q] = *reinterpret_cast<const at_f32x4*>(beta + offr + (long)(16 * i + 2 * q + (lane >> 5)) * m);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) bt[i][j] = *reinterpret_cast<const at_f32x4*>(beta + off + (long)(16 * i + r) * m + 16 * j);
    }
    float4 qa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) qa[i] = *reinterpret_cast<const float4*>(qs + (16 * i + r) * STRIDE + 16 * t + 4 * kk);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(at_get(kb[t & 1][j], s), at_get(qa[i], s), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: lane (r, kk) holds dbeta[row 16 i + r][key 128 wv + 16 j + 4 kk .. + 3]
  float* __restrict__ stw = stage + (STAGE ? wv * (16 * AT_SROW) : 0);
  if constexpr (STAGE) {             // beta: row-contiguous registers -> stage -> accumulator layout, one 16-row group at a time
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<at_f32x4*>(stw + (2 * q + (lane >> 5)) * AT_SROW + 4 * (lane & 31)) = bt[i][q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int j = 0; j < 8; ++j) bt[i][j] = *reinterpret_cast<const at_f32x4*>(stw + r * AT_SROW + 16 * j + 4 * kk);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  float dot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const at_f32x4 pr = bt[i][j] * acc[i][j];
      s += (pr[0] + pr[1]) + (pr[2] + pr[3]);
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (kk == 0) red[wv][16 * i + r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float s = red[0][16 * i + r];
    for (int w = 1; w < nw; ++w) s += red[w][16 * i + r];
    dot[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      at_f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = bt[i][j][e] * (acc[i][j][e] - dot[i]);
      if constexpr (STAGE) *reinterpret_cast<at_f32x4*>(stw + r * AT_SROW + 16 * j + 4 * kk) = o;
      else *reinterpret_cast<at_f32x4*>(dS + off + (long)(16 * i + r) * m + 16 * j) = o;
    }
    if constexpr (STAGE) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = 2 * q + (lane >> 5);
        *reinterpret_cast<at_f32x4*>(dS + offr + (long)(16 * i + row) * m) = *reinterpret_cast<const at_f32x4*>(stw + row * AT_SROW + 4 * (lane & 31));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// 1 when the fused backward serves the shape: n a multiple of 32, m a multiple of 128 up to 1024, dv in {96, 192}
extern "C" int icg_attn_dscores_applies(int n, int m, int dv) {
  if (n < 32 || n % 32 != 0 || m < 128 || m % 128 != 0 || m > 1024) return 0;
  return (dv == 96 || dv == 192) ? 1 : 0;
}

// dS [B][n][m] from dO [B][n][dv], V = g [B][m][dv] and beta [B][n][m]
extern "C" int icg_attn_dscores(const float* dO, const float* V, const float* beta, float* dS, int B, int n, int m, int dv,
                                void* stream) {
  ICG_REQUIRE(dO && V && beta && dS && B > 0);
  ICG_REQUIRE(icg_attn_dscores_applies(n, m, dv));
  ICG_REQUIRE(((uintptr_t)dO % 16) == 0 && ((uintptr_t)V % 16) == 0 && ((uintptr_t)beta % 16) == 0 && ((uintptr_t)dS % 16) == 0);
  const long blocks = (long)B * (n / 32);
  ICG_REQUIRE(blocks < 0x7fffffffL);
  const dim3 grid((unsigned)blocks), block((unsigned)(64 * (m / 128)));
  hipStream_t st = (hipStream_t)stream;
  static const bool stage = [] { const char* e = getenv("ICG_ATTN_STAGE"); return !(e && e[0] == '0'); }();
  if (stage) {
    if (dv == 96) hipLaunchKernelGGL((icg_attn_dscores_kernel<96, true>), grid, block, 0, st, dO, V, beta, dS, n, m);
    else hipLaunchKernelGGL((icg_attn_dscores_kernel<192, true>), grid, block, 0, st, dO, V, beta, dS, n, m);
  } else if (dv == 96) hipLaunchKernelGGL((icg_attn_dscores_kernel<96, false>), grid, block, 0, st, dO, V, beta, dS, n, m);
  else hipLaunchKernelGGL((icg_attn_dscores_kernel<192, false>), grid, block, 0, st, dO, V, beta, dS, n, m);
  return icg_check_launch();
}

// ---------------------------------------------------------------- the three input projections of the block as ONE 1x1 convolution
// theta, phi and g (layers.py:217-231) read the same x; with their weights stacked into one [2 d + dv][C] matrix the block runs one
// GEMM with 2 d + dv = 288 / 144 columns instead of three with 48 / 48 / 192 (24 / 24 / 96) -- the narrow ones fill a quarter or
// half of a 96-column MFMA tile -- and one data-gradient and one weight-gradient GEMM in the backward pass.  These two kernels sit
// between that GEMM and the attention core: they split its output y [B][H][W][2 d + dv] into theta [B][HW][d] and the 2x2
// max-pooled phi [B][HW/4][d] and g [B][HW/4][dv] (forward), and assemble dy from dtheta and the pooled gradients routed to the
// first maximum of each window, exactly as icg_maxpool2_bwd does (backward).  One thread per (pooled pixel, channel quad).
template <int BWD>
__global__ __launch_bounds__(256) void attn_split_pool_kernel(const float4* __restrict__ y, const float4* __restrict__ gt,
                                                              const float4* __restrict__ gp, const float4* __restrict__ gg,
                                                              float4* __restrict__ o0, float4* __restrict__ o1,
                                                              float4* __restrict__ o2, int B, int H, int W, int d4, int dv4) {
  const int C4 = 2 * d4 + dv4, Hp = H >> 1, Wp = W >> 1;
  const long total = (long)B * Hp * Wp * C4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int wp = (int)(t % Wp);
    t /= Wp;
    const int hp = (int)(t % Hp);
    const long b = t / Hp;
    const long pix = (b * H + 2 * hp) * W + 2 * wp;               // top-left pixel of the window
    const long base = pix * C4 + c, o01 = C4, o10 = (long)W * C4, o11 = o10 + C4;
    const long ppix = (b * Hp + hp) * Wp + wp;
    if (c < d4) {                                                 // theta: a copy
      const long tb = pix * d4 + c, t10 = (long)W * d4;
      if (!BWD) {
        o0[tb] = y[base]; o0[tb + d4] = y[base + o01]; o0[tb + t10] = y[base + o10]; o0[tb + t10 + d4] = y[base + o11];
      } else {
        o0[base] = gt[tb]; o0[base + o01] = gt[tb + d4]; o0[base + o10] = gt[tb + t10]; o0[base + o11] = gt[tb + t10 + d4];
      }
      continue;
    }
    const float4 v0 = y[base], v1 = y[base + o01], v2 = y[base + o10], v3 = y[base + o11];
    const bool is_phi = c < 2 * d4;
    const long pidx = is_phi ? ppix * d4 + (c - d4) : ppix * dv4 + (c - 2 * d4);
    if (!BWD) {
      float4 m;
      m.x = fmaxf(fmaxf(fmaxf(v0.x, v1.x), v2.x), v3.x);
      m.y = fmaxf(fmaxf(fmaxf(v0.y, v1.y), v2.y), v3.y);
      m.z = fmaxf(fmaxf(fmaxf(v0.z, v1.z), v2.z), v3.z);
      m.w = fmaxf(fmaxf(fmaxf(v0.w, v1.w), v2.w), v3.w);
      (is_phi ? o1 : o2)[pidx] = m;
    } else {
      const float4 g = (is_phi ? gp : gg)[pidx];
      float4 r0, r1, r2, r3;
#define AT_ROUTE(f)                                                        \
  {                                                                        \
    int arg = 0;                                                           \
    float m = v0.f;                                                        \
    if (v1.f > m) { m = v1.f; arg = 1; }                                   \
    if (v2.f > m) { m = v2.f; arg = 2; }                                   \
    if (v3.f > m) { m = v3.f; arg = 3; }                                   \
    r0.f = arg == 0 ? g.f : 0.f; r1.f = arg == 1 ? g.f : 0.f;              \
    r2.f = arg == 2 ? g.f : 0.f; r3.f = arg == 3 ? g.f : 0.f;              \
  }
      AT_ROUTE(x) AT_ROUTE(y) AT_ROUTE(z) AT_ROUTE(w)
#undef AT_ROUTE
      o0[base] = r0; o0[base + o01] = r1; o0[base + o10] = r2; o0[base + o11] = r3;
    }
  }
}

static bool at_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int icg_attn_split_pool(const float* y, float* theta, float* phi_p, float* g_p, int B, int H, int W, int d, int dv,
                                   void* stream) {
  ICG_REQUIRE(y && theta && phi_p && g_p && B > 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && d > 0 && dv > 0 && !(d & 3) && !(dv & 3));
  ICG_REQUIRE(at_al16(y) && at_al16(theta) && at_al16(phi_p) && at_al16(g_p));
  const long total = (long)B * (H / 2) * (W / 2) * ((2 * d + dv) / 4);
  const long blocks = icg_cdiv(total, 256);
  hipLaunchKernelGGL(attn_split_pool_kernel<0>, dim3((unsigned)(blocks > ICG_GRID_CAP ? ICG_GRID_CAP : blocks)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)y, (const float4*)nullptr, (const float4*)nullptr, (const float4*)nullptr,
                     (float4*)theta, (float4*)phi_p, (float4*)g_p, B, H, W, d / 4, dv / 4);
  return icg_check_launch();
}

extern "C" int icg_attn_split_pool_bwd(const float* y, const float* dtheta, const float* dphi_p, const float* dg_p, float* dy, int B,
                                       int H, int W, int d, int dv, void* stream) {
  ICG_REQUIRE(y && dtheta && dphi_p && dg_p && dy && B > 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && d > 0 && dv > 0 && !(d & 3) &&
              !(dv & 3));
  ICG_REQUIRE(at_al16(y) && at_al16(dtheta) && at_al16(dphi_p) && at_al16(dg_p) && at_al16(dy));
  const long total = (long)B * (H / 2) * (W / 2) * ((2 * d + dv) / 4);
  const long blocks = icg_cdiv(total, 256);
  hipLaunchKernelGGL(attn_split_pool_kernel<1>, dim3((unsigned)(blocks > ICG_GRID_CAP ? ICG_GRID_CAP : blocks)), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)y, (const float4*)dtheta, (const float4*)dphi_p, (const float4*)dg_p,
                     (float4*)dy, (float4*)nullptr, (float4*)nullptr, B, H, W, d / 4, dv / 4);
  return icg_check_launch();
}

// ---------------------------------------------------------------- gamma folded into the output projection
// out = gamma * conv1x1(a, W / sigma) + x (layers.py:242-244) as ONE convolution with the weight gamma * W / sigma and x as the
// residual operand of its epilogue: the projection's output o and the gamma * o + x pass never touch HBM (two reads + one write of
// [B][C][H][W] per forward, the same again in the backward pass).  gamma lives on the device: these two single-workgroup kernels
// scale the (tiny) weight matrices by it and, in the backward pass, turn the gradient of the scaled weight into
//   dgamma = <dWs, W / sigma>,   d(W / sigma) = gamma * dWs.
__global__ __launch_bounds__(1024) void attn_gamma_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ a,
                                                                float* __restrict__ as, const float* __restrict__ b,
                                                                float* __restrict__ bs, long n) {
  const float g = gamma[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    as[i] = g * a[i];
    if (b) bs[i] = g * b[i];
  }
}

__global__ __launch_bounds__(1024) void attn_gamma_bwd_kernel(const float* __restrict__ gamma, const float* __restrict__ dws,
                                                              const float* __restrict__ w, float* __restrict__ dw,
                                                              float* __restrict__ dgamma, long n) {
  __shared__ double red[16];
  const float g = gamma[0];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = dws[i];
    acc += (double)v * (double)w[i];
    dw[i] = g * v;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += red[k];
    dgamma[0] = (float)s;
  }
}

extern "C" int icg_attn_gamma_scale(const float* gamma, const float* w_a, float* ws_a, const float* w_b, float* ws_b, int64_t n,
                                    void* stream) {
  ICG_REQUIRE(gamma && w_a && ws_a && n > 0 && (!w_b || ws_b));
  const long blocks = icg_cdiv(n, 1024);
  hipLaunchKernelGGL(attn_gamma_scale_kernel, dim3((unsigned)(blocks > 256 ? 256 : blocks)), dim3(1024), 0, (hipStream_t)stream, gamma,
                     w_a, ws_a, w_b, ws_b, (long)n);
  return icg_check_launch();
}

extern "C" int icg_attn_gamma_bwd(const float* gamma, const float* dws, const float* w, float* dw, float* dgamma, int64_t n,
                                  void* stream) {
  ICG_REQUIRE(gamma && dws && w && dw && dgamma && n > 0);
  hipLaunchKernelGGL(attn_gamma_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, gamma, dws, w, dw, dgamma, (long)n);
  return icg_check_launch();
}
