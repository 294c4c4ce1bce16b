// Exact L2 k-nearest-neighbour build over the instance-feature table.
// Replaces faiss IndexFlatL2.search / sklearn pairwise_distances + argpartition behind
// ILSVRC_HDF5_feats._obtain_nns (data_utils/datasets_common.py:695-746, 747-769) and data_utils/make_hdf5_nns.py:97-172:
// every row of feats [N][D] is searched against the table itself for its k nearest rows (the caller passes k_nn + 1 and
// drops the query, as the reference does).
//
//   |a - b|^2 = |a|^2 + |b|^2 - 2 a.b : the N x N inner products come from the fp32 MFMA GEMM (icg_gemm_batched, A B^T form),
//   one block of query rows at a time; the selection is one wavefront per query row with the running top-k kept SORTED
//   ACROSS LANES (lane l = l-th nearest so far): 64 candidates are compared with the k-th best at a time (one ballot), and
//   the few that beat it are inserted with a ballot-popcount position search and one wave shuffle (shift right by one lane).
//   Expected insertions per row ~ k ln(N / k); everything else is a coalesced streaming read of the inner-product block.
// Ties resolve to the lower index (candidates arrive in index order and are inserted behind equal distances); the query row
// itself is forced to rank 0.  k <= 64.
#include "icg_common.h"
#include <math.h>

extern "C" int icg_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int transA, int transB,
                                int64_t strideA, int64_t strideB, int64_t strideC, int batch, float alpha, void* stream);

// sq[i] = |feats[i]|^2, one wavefront per row
__global__ __launch_bounds__(256) void knn_rownorm_kernel(const float* __restrict__ f, int N, int D, float* __restrict__ sq) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float* r = f + i * D;
  float a = 0.f;
  for (int j = lane; j < D; j += 64) a = fmaf(r[j], r[j], a);
  a = wave_sum(a);
  if (lane == 0) sq[i] = a;
}

__global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ G, const float* __restrict__ sq, int q0,
                                                         int QB, int N, int k, int64_t* __restrict__ idx_out,
                                                         float* __restrict__ d2_out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= QB) return;                       // (whole wavefronts exit together)
  const int q = q0 + r;
  const float* g = G + (long)r * N;
  const float sq_q = sq[q];
  float bd = INFINITY;                       // lane l < k: distance of the l-th nearest so far (ascending over lanes)
  int bi = -1;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int j = c0 + lane;
    float d = INFINITY;
    if (j < N) {
      d = fmaxf(sq_q + sq[j] - 2.f * g[j], 0.f);
      if (j == q) d = -1.f;                  // the query is its own nearest hit by construction, not by rounding luck
    }
    float thr = __shfl(bd, k - 1, 64);
    unsigned long long mask = __ballot(d < thr);
    while (mask) {                           // wave-uniform loop over the candidates that beat the current k-th best
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const float cd = __shfl(d, b, 64);
      if (!(cd < thr)) continue;             // the threshold may have tightened since the ballot
      const int cj = c0 + b;
      const int pos = __popcll(__ballot(lane < k && bd <= cd));   // entries not farther than the candidate stay in front
      const float up_d = __shfl_up(bd, 1, 64);
      const int up_i = __shfl_up(bi, 1, 64);
      if (lane == pos) { bd = cd; bi = cj; }
      else if (lane > pos && lane < k) { bd = up_d; bi = up_i; }
      thr = __shfl(bd, k - 1, 64);
    }
  }
  if (lane < k) {
    idx_out[(long)q * k + lane] = (int64_t)bi;
    d2_out[(long)q * k + lane] = fmaxf(bd, 0.f);
  }
}

static int knn_block_rows(int N) {
  long qb = (1L << 28) / (N > 0 ? N : 1);    // inner-product block of <= 1 GiB
  if (qb > 4096) qb = 4096;
  if (qb < 128) qb = 128;
  qb = (qb / 128) * 128;
  return (int)(qb > N ? N : qb);
}

extern "C" size_t icg_knn_l2_workspace_bytes(int N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return ((size_t)N + (size_t)knn_block_rows(N) * (size_t)N) * sizeof(float) + 256;
}

extern "C" int icg_knn_l2(const float* feats, int N, int D, int k, int64_t* idx, float* d2, void* workspace,
                          size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(feats && idx && d2 && workspace && N > 0 && D > 0 && k >= 1 && k <= 64 && k <= N);
  if (workspace_bytes < icg_knn_l2_workspace_bytes(N, D)) return ICG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* sq = (float*)workspace;
  float* G = sq + (((size_t)N + 63) / 64) * 64;
  hipLaunchKernelGGL(knn_rownorm_kernel, dim3((unsigned)icg_cdiv(N, 4)), dim3(256), 0, st, feats, N, D, sq);
  const int QB = knn_block_rows(N);
  for (int q0 = 0; q0 < N; q0 += QB) {
    const int qb = (N - q0 < QB) ? N - q0 : QB;
    const int rc = icg_gemm_batched(feats + (size_t)q0 * D, feats, G, qb, N, D, 0, 1, 0, 0, 0, 1, 1.0f, stream);
    if (rc != ICG_OK) return rc;
    hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)icg_cdiv(qb, 4)), dim3(256), 0, st, (const float*)G, (const float*)sq,
                       q0, qb, N, k, idx, d2);
  }
  return icg_check_launch();
}
