// Plane GEMM, second generation: C[z] = alpha * A[z] B[z]^T for the batched GEMMs over Winograd planes (A [M][K] tiles x
// channels, B [N][K] the Winograd-domain weights; K = Cin contiguous in both) -- the forward / data-gradient half of every
// Winograd composite of winograd.hip, i.e. the kernels that dominate the IC-GAN BigGAN step (layers.py:144-153).
//
// Why a second kernel: the first-generation plane GEMM (icg_planes_body, gemm_conv.hip) stages both operands
// global -> VGPR -> ds_write_b32 x 16 -> LDS and reads one ds_read_b32 per MFMA operand; with two-level accumulation it holds
// 144 VGPR + 64 AGPR and runs at TWO waves per SIMD, which is what caps it at ~0.69 of the fp32 MFMA peak (single-level at three
// waves: 0.76).  This kernel is built around the three things gfx950 offers for exactly that problem:
//   * LDS-DMA (global_load_lds_dwordx4): operand tiles go HBM/L2 -> LDS without touching a VGPR and without a single
//     ds_write; two DMA instructions per wave and K-tile replace 4 global loads + 16 LDS stores + their address arithmetic
//   * ds_read_b128 operand fragments: with a permuted K order (a lane's four consecutive k-values are the operands of four
//     successive k-steps of v_mfma_f32_16x16x4_f32) ONE 16-byte LDS read feeds four MFMAs: 6 LDS reads per 32 MFMAs instead of 40
//   * 8 waves per workgroup, each owning a 32 x 64 (or 32 x 48) piece of the 128 x 128 (128 x 96) output tile: 32 (24) first-level
//     + 32 (24) second-level accumulator registers per lane instead of 64 + 64, so the two-level kernel fits 4 waves per SIMD
//     (two workgroups per CU) instead of 2
// The LDS image the DMA writes is lane-linear (wave-uniform base + 16 B x lane), so the bank-conflict-free layout for the
// 16-byte fragment reads is obtained by permuting the 16-byte chunks on the SOURCE side (which chunk of its row a lane
// fetches) and applying the same involution to the read address (cdna_hip_programming.md rule 21).
//
// Pipeline: 3-slot LDS ring, ONE raw s_barrier per K-tile, DMA two K-tiles ahead, counted s_waitcnt vmcnt(N) (never 0 in the
// loop).  The DMA is issued from inline asm: hipcc's own __builtin_amdgcn_global_load_lds makes the compiler drain vmcnt(0)
// before every LDS read that follows (it cannot tell the ring slots apart), which serialises load and compute.
#include "icg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PgemmP {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long ldc;
  long sA, sB, sC;        // per-plane strides (elements)
  float alpha;
  int tiles_n, tiles_mn;  // n-tiles per m-tile row, output tiles per plane
  unsigned total;         // output tiles over all planes (= grid size)
  int swz;                // XCD-aware tile order
};

// chunk permutation of the LDS image: the 16-byte chunk c (0..3) of row r sits at chunk position c ^ g(r), g = 0,0,3,3 by
// (r >> 2) & 3.  With it the four 16-lane groups a ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... --
// MI355X_MICROARCH.md, LDS) each touch 16 distinct 16-byte bank groups for the 16x16x4 operand pattern (lane l: row l & 15,
// chunk l >> 4): conflict-free.
__device__ __forceinline__ int pg_swz(int row) { return (row & 8) ? 3 : 0; }

// one LDS-DMA instruction: every lane fetches 16 bytes from gbase + voff (bytes) and the wave's 1 KiB lands at LDS byte
// address lds_dst + 16 * lane.  M0 carries the LDS address; it is compiler-reserved, so it is saved and restored inside the
// statement (cdna_hip_programming.md 5.7).  Not visible to hipcc's s_waitcnt bookkeeping: counted by hand below.
__device__ __forceinline__ void pg_dma16(const float* gbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(gbase), "s"(lds_dst)
      : "memory");
}

// NT: 16-column MFMA tiles per wave (4: 128-column workgroup tile, 3: 96-column); LEVELS: accumulation levels (see gemm_conv.hip,
// BLK: chains restart every 32 k-values and the finished chain is added into a second accumulator set)
template <int NT, int LEVELS>
__global__ __launch_bounds__(512, 4) void icg_pgemm_nn_kernel(PgemmP p) {
  constexpr int BM = 128, BN = 32 * NT, BK = 16;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, SLOT = A_BYTES + B_BYTES, NBUF = 3;
  constexpr int B_CHUNKS = BN / 16;                 // DMA instructions (16 rows x 64 B = 1 KiB) per B tile; the A tile has 8
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7, wave-uniform (SGPR)
  const int wm = wv & 3, wn = wv >> 2;                              // wave tile: rows 32 wm .., columns 16 NT wn ..
  const int r = lane & 15, kk = lane >> 4;

  // ---- output tile (XCD-aware order, as icg_gemm_body: each XCD owns a contiguous range of the (plane, m-tile, n-tile) order)
  unsigned t = blockIdx.x;
  if (p.swz) {
    const unsigned tot = p.total, q = tot >> 3, rr = tot & 7u, xcd = t & 7u;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
  }
  const int z = (int)(t / (unsigned)p.tiles_mn);
  const int tile = (int)(t - (unsigned)z * (unsigned)p.tiles_mn);
  const int nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ Ag = p.A + (long)z * p.sA;
  const float* __restrict__ Bg = p.B + (long)z * p.sB;

  // ---- DMA role of this lane: every wave fetches 16 rows of the A tile (1 KiB) and BN/8 rows of the B tile per K-tile; the
  // lane's 16 bytes land at chunk position lane & 3 of row (lane >> 2) of the wave's piece
  constexpr int BROWS = BN / 8;                                     // 16 (all 64 lanes) or 12 (lanes 0..47; the rest are masked off)
  const int drowA = 16 * wv + (lane >> 2), drowB = BROWS * wv + (lane >> 2);
  const unsigned voffA =
      ((unsigned)min(m0 + drowA, p.M - 1) * (unsigned)p.K + 4u * (unsigned)((lane & 3) ^ pg_swz(drowA))) * 4u;
  const unsigned voffB =
      ((unsigned)min(n0 + min(drowB, BN - 1), p.N - 1) * (unsigned)p.K + 4u * (unsigned)((lane & 3) ^ pg_swz(drowB))) * 4u;
  const bool dma_b_lane = (BROWS == 16) || ((lane >> 2) < BROWS);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1024u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * (BROWS * 64u);
  const int nk = p.K / BK;                                          // even (K % 32 == 0)
  auto issue = [&](int kt, unsigned slot_off) {                     // K-tile kt -> ring slot (past the end: the last tile again)
    const int kc = min(kt, nk - 1) * BK;
    pg_dma16(Ag + kc, voffA, ldsA + slot_off);
    if (dma_b_lane) pg_dma16(Bg + kc, voffB, ldsB + slot_off);
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  // ---- fragment addresses: lane (r, kk) reads the 16 bytes of chunk kk of row (tile base + r)
  const int fo = r * 64 + ((kk ^ pg_swz(r)) * 16);
  const char* fa = lds + fo + (32 * wm) * 64;
  const char* fb = lds + fo + A_BYTES + (16 * NT * wn) * 64;

  f32x4 acc[2][NT], acc2[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  issue(0, 0u);
  issue(1, (unsigned)SLOT);

  // one K-tile from the ring slot at byte offset `cur`; FLUSH: fold the finished 32-deep chains into the second level and
  // start fresh ones
  auto tile_step = [&](int kt, unsigned cur, auto flush_c) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    // K-tile kt has landed once at most the two DMAs of K-tile kt+1 are outstanding; this wave's LDS reads of K-tile kt-1 have
    // all returned (their MFMAs were issued), so after the barrier that tile's slot may be overwritten
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + 2, next_slot(next_slot(cur)));                       // into the slot K-tile kt-1 occupied
    float4 a[2], b[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(fa + cur + i * 1024);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(fb + cur + j * 1024);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float av = s == 0 ? a[i].x : (s == 1 ? a[i].y : (s == 2 ? a[i].z : a[i].w));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = s == 0 ? b[j].x : (s == 1 ? b[j].y : (s == 2 ? b[j].z : b[j].w));
          if (LEVELS == 2 && FLUSH && s == 0) {
            acc2[i][j] += acc[i][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  unsigned cur = 0u;
  for (int kt = 0; kt < nk; kt += 2) {                              // flush period 2: the pair is the loop body
    tile_step(kt, cur, std::true_type{});
    cur = next_slot(cur);
    tile_step(kt + 1, cur, std::false_type{});
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the over-fetched tiles: nothing may land after the exit

  // ---- epilogue.  The MFMA operands are swapped (D = B-fragment x A-fragment), so lane (r, kk) holds C[16 i + r][16 j + 4 kk .. + 3]
  // of its wave tile: one 16-byte store per 16 x 16 tile when C allows it
  float* __restrict__ Cg = p.C + (long)z * p.sC;
  const bool c_vec = (((uintptr_t)p.C | (uintptr_t)(p.ldc * 4) | (uintptr_t)(p.sC * 4)) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 16 * i + r;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 16 * NT * wn + 16 * j + 4 * kk;
      f32x4 v = (LEVELS == 2) ? acc[i][j] + acc2[i][j] : acc[i][j];
      v *= p.alpha;
      if (m < p.M && n < p.N) {
        float* dst = Cg + (long)m * p.ldc + n;
        if (c_vec) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = v[e];
        }
      }
    }
  }
}

// ---- the same kernel on v_mfma_f32_32x32x2_f32: a wave owns the same 32 x 64 piece as ONE 32-row block x TWO
// 32-column blocks, i.e. 16 MFMAs of 64 cycles per K-tile instead of 32 of 32 cycles -- half the matrix instructions to issue
// next to the DMA / LDS / barrier traffic, and a dependent accumulator chain that the instruction's own latency (64 cycles issue =
// 64 dependent) never stalls.  Operand fragments: lane l holds row l & 31, k-half l >> 5 (cdna_hip_programming.md, MFMA); with the
// permuted K order a 16-byte LDS read feeds four successive MFMAs -- read q of a K-tile gives lane (row, kh) the chunk 2 q + kh, so
// k-step t of read q contracts k = 4 (2 q + kh) + t, identically in both operands.  LDS image: the same lane-linear DMA rows of 64
// bytes; chunk c of row r sits at position c ^ g(r >> 2) with g(u) = (u ^ (u >> 1)) & 3, which is a bijection on {0, 3, 5, 6} and on
// {1, 2, 4, 7} -- the row quads the two 16-lane service groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a ds_read_b128 touch
// (MI355X_MICROARCH.md, LDS) -- so every group reads 16 distinct 16-byte bank slots.  128-column tile only (N % 128 == 0).
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int pg32_swz(int row) { const int u = (row >> 2) & 7; return (u ^ (u >> 1)) & 3; }

template <int LEVELS>
__global__ __launch_bounds__(512, 4) void icg_pgemm_nn32_kernel(PgemmP p) {
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, SLOT = A_BYTES + B_BYTES, NBUF = 3;
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv & 3, wn = wv >> 2;                              // wave tile: rows 32 wm .., columns 64 wn ..
  const int r32 = lane & 31, kh = lane >> 5;

  unsigned t = blockIdx.x;
  if (p.swz) {
    const unsigned tot = p.total, q = tot >> 3, rr = tot & 7u, xcd = t & 7u;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
  }
  const int z = (int)(t / (unsigned)p.tiles_mn);
  const int tile = (int)(t - (unsigned)z * (unsigned)p.tiles_mn);
  const int nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const float* __restrict__ Ag = p.A + (long)z * p.sA;
  const float* __restrict__ Bg = p.B + (long)z * p.sB;

  const int drow = 16 * wv + (lane >> 2);                           // DMA role: row of the A / B tile, chunk position lane & 3
  const unsigned voffA = ((unsigned)min(m0 + drow, p.M - 1) * (unsigned)p.K + 4u * (unsigned)((lane & 3) ^ pg32_swz(drow))) * 4u;
  const unsigned voffB = ((unsigned)min(n0 + drow, p.N - 1) * (unsigned)p.K + 4u * (unsigned)((lane & 3) ^ pg32_swz(drow))) * 4u;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1024u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * 1024u;
  const int nk = p.K / BK;                                          // even (K % 32 == 0)
  auto issue = [&](int kt, unsigned slot_off) {
    const int kc = min(kt, nk - 1) * BK;
    pg_dma16(Ag + kc, voffA, ldsA + slot_off);
    pg_dma16(Bg + kc, voffB, ldsB + slot_off);
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  // fragment addresses: read q of a K-tile = chunk 2 q + kh of the lane's row
  const int rowA = 32 * wm + r32, sA = pg32_swz(rowA);
  const char* fa[2];
  const char* fb[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    fa[q] = lds + rowA * 64 + (((2 * q + kh) ^ sA) * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rowB = 64 * wn + 32 * j + r32;
      fb[j][q] = lds + A_BYTES + rowB * 64 + (((2 * q + kh) ^ pg32_swz(rowB)) * 16);
    }
  }

  f32x16 acc[2], acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; acc2[j][e] = 0.f; }

  issue(0, 0u);
  issue(1, (unsigned)SLOT);

  auto tile_step = [&](int kt, unsigned cur, auto flush_c) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + 2, next_slot(next_slot(cur)));
    float4 a[2], b[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      a[q] = *reinterpret_cast<const float4*>(fa[q] + cur);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j][q] = *reinterpret_cast<const float4*>(fb[j][q] + cur);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = s == 0 ? a[q].x : (s == 1 ? a[q].y : (s == 2 ? a[q].z : a[q].w));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float bv = s == 0 ? b[j][q].x : (s == 1 ? b[j][q].y : (s == 2 ? b[j][q].z : b[j][q].w));
          if (LEVELS == 2 && FLUSH && q == 0 && s == 0) {
            acc2[j] += acc[j];
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.f;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, zero, 0, 0, 0);
          } else {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[j], 0, 0, 0);
          }
        }
      }
    }
  };
  unsigned cur = 0u;
  for (int kt = 0; kt < nk; kt += 2) {
    tile_step(kt, cur, std::true_type{});
    cur = next_slot(cur);
    tile_step(kt + 1, cur, std::false_type{});
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // epilogue: D = B-fragment x A-fragment, so lane (r32, kh) register v holds C[m = r32][n = 8 (v >> 2) + 4 kh + (v & 3)] of its block
  float* __restrict__ Cg = p.C + (long)z * p.sC;
  const bool c_vec = (((uintptr_t)p.C | (uintptr_t)(p.ldc * 4) | (uintptr_t)(p.sC * 4)) & 15) == 0;
  const int m = m0 + 32 * wm + r32;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f32x16 v = (LEVELS == 2) ? acc[j] + acc2[j] : acc[j];
    v *= p.alpha;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + 64 * wn + 32 * j + 8 * g + 4 * kh;
      if (m < p.M && n < p.N) {
        float* dst = Cg + (long)m * p.ldc + n;
        if (c_vec) {
          *reinterpret_cast<f32x4*>(dst) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = v[4 * g + e];
        }
      }
    }
  }
}

// ---- streaming form of the kernel above: a workgroup owns a RUN of output tiles (first, first + vstep, ... inside its XCD's
// contiguous range of the tile order, like icg_planes_body) and treats their K-tiles as one stream -- the DMA for the next
// output tile's first two K-tiles is in flight while the last MFMAs and the epilogue of the current one run, so the pipeline
// fill (two K-tiles of HBM latency) is paid once per workgroup instead of once per output tile.  That is what the short-K
// layers need (K = 96 / 192: 6 / 12 K-tiles per output tile).  The MFMA operands are swapped (D = B-fragment x A-fragment), which
// leaves lane (r, kk) with C[m = r][n = 4 kk .. 4 kk + 3]: the epilogue is one 16-byte store per 16 x 16 tile instead of four
// 4-byte stores.  Epilogue stores count on vmcnt like the DMA loads but complete out of order with them, so the first K-tile after
// an epilogue waits for vmcnt(0) (its two prefetched K-tiles were issued before the epilogue: no bubble).
struct PgemmSP {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long ldc;
  long sA, sB, sC;
  float alpha;
  int tiles_n, tiles_mn;
  unsigned total;         // output tiles over all planes
  int swz;
};

// NBUF: LDS ring slots; the DMA runs NBUF - 1 K-tiles ahead of the MFMAs
template <int NT, int LEVELS, int NBUF = 3>
__global__ __launch_bounds__(512, 4) void icg_pgemm_nn_stream_kernel(PgemmSP p) {
  constexpr int BM = 128, BN = 32 * NT, BK = 16;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, SLOT = A_BYTES + B_BYTES;
  constexpr int BROWS = BN / 8;
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv & 3, wn = wv >> 2;
  const int r = lane & 15, kk = lane >> 4;

  // this workgroup's run of output tiles
  const unsigned tot = p.total, lin = blockIdx.x;
  unsigned first, last, vstep;
  if (p.swz) {
    const unsigned q = tot >> 3, rr = tot & 7u, xcd = lin & 7u;
    const unsigned base = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    vstep = gridDim.x >> 3;
    first = base + (lin >> 3);
    last = base + q + (xcd < rr ? 1u : 0u);
  } else {
    vstep = gridDim.x;
    first = lin;
    last = tot;
  }
  if (first >= last) return;

  const int nk = p.K / BK;                                          // even
  const bool dma_b_lane = (BROWS == 16) || ((lane >> 2) < BROWS);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1024u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * (BROWS * 64u);

  // ---- load cursor: (output tile lv, K-tile lk) of the next DMA
  unsigned lv = first;
  int lk = 0;
  const float* lAg;
  const float* lBg;
  unsigned lvoffA, lvoffB;
  auto set_load_tile = [&](unsigned v) {
    // (runs once per output tile: the lane constants of the DMA role are recomputed here, from a lane id that costs no live
    // register, instead of being kept in registers the 128-register budget does not have)
    int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // lane id, from the exec mask
    asm volatile("" : "+v"(ln));                                                            // (what follows stays in this branch)
    const int drowA = 16 * wv + (ln >> 2), drowB = BROWS * wv + (ln >> 2);
    const unsigned swzA = 4u * (unsigned)((ln & 3) ^ pg_swz(drowA)), swzB = 4u * (unsigned)((ln & 3) ^ pg_swz(drowB));
    const int z = (int)(v / (unsigned)p.tiles_mn);
    const int tile = (int)(v - (unsigned)z * (unsigned)p.tiles_mn);
    const int nt = tile % p.tiles_n, mt = tile / p.tiles_n;
    lAg = p.A + (long)z * p.sA;
    lBg = p.B + (long)z * p.sB;
    lvoffA = ((unsigned)min(mt * BM + drowA, p.M - 1) * (unsigned)p.K + swzA) * 4u;
    lvoffB = ((unsigned)min(nt * BN + min(drowB, BN - 1), p.N - 1) * (unsigned)p.K + swzB) * 4u;
  };
  auto issue_next = [&](unsigned slot_off) {
    pg_dma16(lAg + lk * BK, lvoffA, ldsA + slot_off);
    if (dma_b_lane) pg_dma16(lBg + lk * BK, lvoffB, ldsB + slot_off);
    if (++lk == nk) {
      if (lv + vstep < last) { lv += vstep; lk = 0; set_load_tile(lv); }
      else lk = nk - 1;                                             // past the end of the run: the last K-tile again (never read)
    }
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  const int fo = r * 64 + ((kk ^ pg_swz(r)) * 16);
  const char* fa = lds + fo + (32 * wm) * 64;
  const char* fb = lds + fo + A_BYTES + (16 * NT * wn) * 64;

  f32x4 acc[2][NT], acc2[2][NT];
  set_load_tile(lv);
#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d) issue_next((unsigned)(d * SLOT));

  // one K-tile.  Every MFMA accumulates in place (same registers in and out, in every step): with separate "fresh chain" forms
  // of the step the register allocator rotates the accumulators through new registers at the joins (+20 registers, spills).
  // FLUSH (every even K-tile): the finished 32-deep chains are folded into the second level and cleared by VALU moves; at the
  // first K-tile of an output tile both levels are zero already (cleared after the epilogue), so the same code serves.
  auto tile_step = [&](unsigned cur, auto flush_c, bool first_of_tile) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    if (FLUSH && first_of_tile) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next(cur == 0u ? (unsigned)((NBUF - 1) * SLOT) : cur - (unsigned)SLOT);      // the slot K-tile kt-1 occupied
    if (FLUSH && LEVELS == 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc2[i][j] += acc[i][j];
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float4 a[2], b[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(fa + cur + i * 1024);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(fb + cur + j * 1024);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float av = s == 0 ? a[i].x : (s == 1 ? a[i].y : (s == 2 ? a[i].z : a[i].w));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = s == 0 ? b[j].x : (s == 1 ? b[j].y : (s == 2 ? b[j].z : b[j].w));
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
        }
      }
    }
  };
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  unsigned cur = 0u;
  for (unsigned v = first; v < last; v += vstep) {
    for (int kt = 0; kt < nk; kt += 2) {
      tile_step(cur, std::true_type{}, kt == 0);
      cur = next_slot(cur);
      tile_step(cur, std::false_type{}, false);
      cur = next_slot(cur);
    }
    // epilogue of output tile v: swapped operands -> lane (r, kk) holds C[16 i + r][16 j + 4 kk .. + 3] of its wave tile
    const int z = (int)(v / (unsigned)p.tiles_mn);
    const int tile = (int)(v - (unsigned)z * (unsigned)p.tiles_mn);
    const int n0 = (tile % p.tiles_n) * BN, m0 = (tile / p.tiles_n) * BM;
    // address = wave-uniform base (scalar registers) + one 32-bit lane offset that is the same for every tile of the run
    float* __restrict__ Cw = p.C + (long)z * p.sC + (long)(m0 + 32 * wm) * p.ldc + (n0 + 16 * NT * wn);
    const unsigned loff = (unsigned)r * (unsigned)p.ldc + 4u * (unsigned)kk;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool row_ok = m0 + 32 * wm + 16 * i + r < p.M;          // (N is a multiple of the tile width)
      float* __restrict__ Ci = Cw + (long)(16 * i) * p.ldc;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v4 = (LEVELS == 2) ? acc[i][j] + acc2[i][j] : acc[i][j];
        v4 *= p.alpha;
        if (row_ok) *reinterpret_cast<f32x4*>(Ci + 16 * j + loff) = v4;
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};                      // both levels clear for the next output tile
        if (LEVELS == 2) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- weight-gradient plane GEMM: C[z][s] = A[z]^T B[z] over the K-slice s, A [K][M] (V planes: tiles x Cin), B [K][N] (transformed
// dy: tiles x Cout), K = tiles (long; split into slices whose partial slabs a second kernel sums in fixed order).  Same pipeline
// as above; the operands are M- / N-contiguous, so a K-tile's LDS image is [16 k][128 m] (512-byte rows, two per DMA
// instruction) and an MFMA operand is one float per lane: lane (r, kk) of k-step s reads row 4 s + kk, column r.  The two
// k-rows a 32-lane LDS group touches (kk = 0 / 1) differ by 128 floats = the same banks, so odd k-rows are stored with their
// 16-byte chunks XOR 4 (columns XOR 16): source-side permutation again, applied to the read address as well.
struct PgemmTnP {
  const float* A;
  const float* B;
  float* C;               // slabs: [plane][slice][M][N]  (slices == 1: the result itself)
  int M, N, K;
  long sA, sB;            // per-plane strides (elements)
  int kchunk, slices;     // K-rows per slice (multiple of 32), slices per plane
  int tiles_n, tiles_mn;
  unsigned total;         // tiles_mn * planes * slices
  int swz;
};

// WM: 32-row wave tiles along M (4: 128-row workgroup tile, 8 waves; 3: 96-row tile, 6 waves -- for M = Cin = 96 / 192, where a
// 128-row tile would spend a quarter of its MFMAs on rows that do not exist)
template <int NT, int LEVELS, int WM = 4, int NBUF = 3>
__global__ __launch_bounds__(128 * WM, (WM == 4 ? 4 : 3)) void icg_pgemm_tn_kernel(PgemmTnP p) {
  constexpr int BM = 32 * WM, BN = 32 * NT, BK = 16, NW = 2 * WM;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, SLOT = A_BYTES + B_BYTES;
  constexpr int ACH = BM / 4, BCH = BN / 4;                         // 16-byte chunks per A / B row: 32 or 24
  constexpr int A_LANES = A_BYTES / 16 / NW, B_LANES = B_BYTES / 16 / NW;   // chunks (lanes) per wave and K-tile: 64, or 48 for 6 KB over 8 waves
  static_assert(A_LANES * NW * 16 == A_BYTES && B_LANES * NW * 16 == B_BYTES && A_LANES <= 64 && B_LANES <= 64, "tile / wave split");
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv % WM, wn = wv / WM;
  const int r = lane & 15, kk = lane >> 4;

  unsigned t = blockIdx.x;
  if (p.swz) {
    const unsigned tot = p.total, q = tot >> 3, rr = tot & 7u, xcd = t & 7u;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
  }
  const int zs = (int)(t / (unsigned)p.tiles_mn);                   // plane * slices + slice
  const int tile = (int)(t - (unsigned)zs * (unsigned)p.tiles_mn);
  const int z = zs / p.slices, sl = zs - z * p.slices;
  const int nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = sl * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg) / BK;                                // even
  const float* __restrict__ Ag = p.A + (long)z * p.sA + (long)kbeg * p.M;
  const float* __restrict__ Bg = p.B + (long)z * p.sB + (long)kbeg * p.N;

  // DMA role: the wave's contiguous piece of each tile image (chunk index L = wave * lanes-per-wave + lane -> row L / chunks-per-
  // row, chunk L % chunks-per-row); columns beyond the matrix are clamped (they only feed masked outputs)
  const int la = min(wv * A_LANES + lane, BK * ACH - 1), lb = min(wv * B_LANES + lane, BK * BCH - 1);
  const int arow = la / ACH, achunk = (la % ACH) ^ (4 * (arow & 1));
  const unsigned voffA = ((unsigned)arow * (unsigned)p.M + (unsigned)min(m0 + 4 * achunk, p.M - 4)) * 4u;
  const int brow = lb / BCH, bchunk = (lb % BCH) ^ (4 * (brow & 1));
  const unsigned voffB = ((unsigned)brow * (unsigned)p.N + (unsigned)min(n0 + 4 * bchunk, p.N - 4)) * 4u;
  const bool dma_a_lane = (A_LANES == 64) || (lane < A_LANES), dma_b_lane = (B_LANES == 64) || (lane < B_LANES);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * (A_LANES * 16u), ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * (B_LANES * 16u);
  const long a_step = (long)BK * p.M, b_step = (long)BK * p.N;      // elements per K-tile
  auto issue = [&](int kt, unsigned slot_off) {
    const int kc = min(kt, nk - 1);
    if (dma_a_lane) pg_dma16(Ag + kc * a_step, voffA, ldsA + slot_off);
    if (dma_b_lane) pg_dma16(Bg + kc * b_step, voffB, ldsB + slot_off);
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  // fragment addresses: k-step s reads row 4 s + kk (parity kk & 1), column (tile base + r) ^ 16 (kk & 1)
  // (the XOR moves a lane between adjacent 16-column MFMA tiles, so every tile gets its own address register)
  const int px = 16 * (kk & 1);
  const char* fa[2];
  const char* fb[NT];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = lds + kk * (BM * 4) + ((32 * wm + 16 * i + r) ^ px) * 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[j] = lds + A_BYTES + kk * (BN * 4) + ((16 * NT * wn + 16 * j + r) ^ px) * 4;

  f32x4 acc[2][NT], acc2[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d) issue(d, (unsigned)(d * SLOT));

  auto tile_step = [&](int kt, unsigned cur, auto flush_c) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + NBUF - 1, cur == 0u ? (unsigned)((NBUF - 1) * SLOT) : cur - (unsigned)SLOT);
    float a[4][2], b[4][NT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[s][i] = *reinterpret_cast<const float*>(fa[i] + cur + s * (4 * BM * 4));
#pragma unroll
      for (int j = 0; j < NT; ++j) b[s][j] = *reinterpret_cast<const float*>(fb[j] + cur + s * (4 * BN * 4));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (LEVELS == 2 && FLUSH && s == 0) {
            acc2[i][j] += acc[i][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  unsigned cur = 0u;
  for (int kt = 0; kt < nk; kt += 2) {
    tile_step(kt, cur, std::true_type{});
    cur = next_slot(cur);
    tile_step(kt + 1, cur, std::false_type{});
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* __restrict__ Cg = p.C + (long)zs * p.M * p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mrow = m0 + 32 * wm + 16 * i + 4 * kk;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 16 * NT * wn + 16 * j + r;
      const f32x4 v = (LEVELS == 2) ? acc[i][j] + acc2[i][j] : acc[i][j];
      if (n < p.N) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (mrow + e < p.M) Cg[(long)(mrow + e) * p.N + n] = v[e];
      }
    }
  }
}

// ---- the same weight-gradient plane GEMM on a 96 x 96 workgroup tile owned by FOUR waves (2 x 2, 48 x 48 = 3 x 3 MFMA tiles each) for the narrow
// layers (M, N multiples of 96: 96 / 192 channels), round 6.  The 6-wave form above keeps 2 workgroups = 12 waves on a CU; these GEMMs sit on
// the ridge (24 FLOP per byte of V + M) and reached 0.53 - 0.60 of the MFMA peak and ~0.6 of the achievable HBM rate at once
// (profiles/r06_pgemm_microbench.txt).  Here a CU holds 3 workgroups of 4 waves (36 KiB of LDS each, 140 registers): three independent
// barrier domains and 1.5 x the tiles in flight; a wave issues 24 single-float LDS reads per 36 MFMAs instead of 20 per 24.  (The forward-shaped sibling of this
// tile was built and measured too: -11 % at 384 -> 192, +3 ... +9 % at K <= 192 and on every 128-multiple width, no gain in the step: not kept.)
// Same K order, same two-level
// accumulation, same slab layout.  DMA: a K-tile image is 16 rows x 24 chunks = 384 chunks per operand = 96 per wave, fetched as 64 + 32 lanes
// (4 DMA instructions per wave and K-tile: the counted vmcnt below is 4 per tile in flight).
template <int LEVELS, int NBUF>
__global__ __launch_bounds__(256, 3) void icg_pgemm_tn96_kernel(PgemmTnP p) {
  constexpr int BM = 96, BN = 96, BK = 16, IT = 3, NT = 3;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, SLOT = A_BYTES + B_BYTES;
  constexpr int CH = BM / 4;                                        // 24 16-byte chunks per row (both operands)
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..3
  const int wm = wv & 1, wn = wv >> 1;
  const int r = lane & 15, kk = lane >> 4;

  unsigned t = blockIdx.x;
  if (p.swz) {
    const unsigned tot = p.total, q = tot >> 3, rr = tot & 7u, xcd = t & 7u;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
  }
  const int zs = (int)(t / (unsigned)p.tiles_mn);
  const int tile = (int)(t - (unsigned)zs * (unsigned)p.tiles_mn);
  const int z = zs / p.slices, sl = zs - z * p.slices;
  const int nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = sl * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg) / BK;                                // even
  const float* __restrict__ Ag = p.A + (long)z * p.sA + (long)kbeg * p.M;
  const float* __restrict__ Bg = p.B + (long)z * p.sB + (long)kbeg * p.N;

  // DMA role: the wave's 96 consecutive chunks of each image, as lanes 0..63 (chunks 96 wv ..) and lanes 0..31 (chunks 96 wv + 64 ..)
  const int l0 = 96 * wv + lane, l1 = 96 * wv + 64 + (lane & 31);
  const int ar0 = l0 / CH, ac0 = (l0 % CH) ^ (4 * (ar0 & 1)), ar1 = l1 / CH, ac1 = (l1 % CH) ^ (4 * (ar1 & 1));
  const unsigned voffA0 = ((unsigned)ar0 * (unsigned)p.M + (unsigned)(m0 + 4 * ac0)) * 4u;
  const unsigned voffA1 = ((unsigned)ar1 * (unsigned)p.M + (unsigned)(m0 + 4 * ac1)) * 4u;
  const unsigned voffB0 = ((unsigned)ar0 * (unsigned)p.N + (unsigned)(n0 + 4 * ac0)) * 4u;
  const unsigned voffB1 = ((unsigned)ar1 * (unsigned)p.N + (unsigned)(n0 + 4 * ac1)) * 4u;
  const bool half = lane < 32;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1536u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * 1536u;
  const long a_step = (long)BK * p.M, b_step = (long)BK * p.N;
  auto issue = [&](int kt, unsigned slot_off) {
    const int kc = min(kt, nk - 1);
    pg_dma16(Ag + kc * a_step, voffA0, ldsA + slot_off);
    if (half) pg_dma16(Ag + kc * a_step, voffA1, ldsA + 1024u + slot_off);
    pg_dma16(Bg + kc * b_step, voffB0, ldsB + slot_off);
    if (half) pg_dma16(Bg + kc * b_step, voffB1, ldsB + 1024u + slot_off);
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  const int px = 16 * (kk & 1);
  const char* fa[IT];
  const char* fb[NT];
#pragma unroll
  for (int i = 0; i < IT; ++i) fa[i] = lds + kk * (BM * 4) + ((48 * wm + 16 * i + r) ^ px) * 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) fb[j] = lds + A_BYTES + kk * (BN * 4) + ((48 * wn + 16 * j + r) ^ px) * 4;

  f32x4 acc[IT][NT], acc2[IT][NT];
#pragma unroll
  for (int i = 0; i < IT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
  for (int d = 0; d < NBUF - 1; ++d) issue(d, (unsigned)(d * SLOT));

  auto tile_step = [&](int kt, unsigned cur, auto flush_c) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + NBUF - 1, cur == 0u ? (unsigned)((NBUF - 1) * SLOT) : cur - (unsigned)SLOT);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float a[IT], b[NT];
#pragma unroll
      for (int i = 0; i < IT; ++i) a[i] = *reinterpret_cast<const float*>(fa[i] + cur + s * (4 * BM * 4));
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float*>(fb[j] + cur + s * (4 * BN * 4));
#pragma unroll
      for (int i = 0; i < IT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (LEVELS == 2 && FLUSH && s == 0) {
            acc2[i][j] += acc[i][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  unsigned cur = 0u;
  for (int kt = 0; kt < nk; kt += 2) {
    tile_step(kt, cur, std::true_type{});
    cur = next_slot(cur);
    tile_step(kt + 1, cur, std::false_type{});
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* __restrict__ Cg = p.C + (long)zs * p.M * p.N;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int mrow = m0 + 48 * wm + 16 * i + 4 * kk;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 48 * wn + 16 * j + r;
      const f32x4 v = (LEVELS == 2) ? acc[i][j] + acc2[i][j] : acc[i][j];
#pragma unroll
      for (int e = 0; e < 4; ++e) Cg[(long)(mrow + e) * p.N + n] = v[e];
    }
  }
}

// ---- second-generation implicit-GEMM convolution (forward / data-gradient shaped: A gathered from an NHWC activation, B = weights
// [N][taps][Cin], K = taps x Cin in tap-minor order -- all taps of a 16-channel slice, then the next slice, like icg_gemm_body's
// fast path).  Same pipeline as the plane GEMM; the A tile's DMA source address is recomputed per K-tile (one tap of one slice:
// pixel * stride + tap - pad, bounds check; out-of-image taps fetch from a 64-byte zero page, so zero padding costs no extra
// instruction in the MFMA loop), the B tile advances by a scalar.  Prologue: none or ReLU (applied to the A fragments after
// the LDS read: max(0, 0) keeps the padding zero); a BN / ccbn affine prologue stays on the first-generation kernel.  Epilogue:
// alpha, bias, residual add or ReLU mask, optional phase scatter (the 4-phase 2x2 forms), 16-byte stores.  Chains are two-level
// (32 k-values), which a direct convolution's K of up to 13 824 needs even more than the plane GEMMs do.
struct PconvP {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int H, W, Cin, R, up, Hs, Ws;
  int pad_h, pad_w, gs, Hb, Wb;
  long ldb, ldc;
  const float* bias;
  const float* res;
  int res_mode;            // 0: added, 2: ReLU mask (output kept where res > 0)
  float alpha;
  long strideB;            // per phase
  int phase_mode, oH, oW;  // 1: blockIdx selects the phase (al, be): pad = (1 - al, 1 - be), output row (2h + al, 2w + be)
  int pre_relu;
  int tiles_n, tiles_mn;
  unsigned total;
  int swz;
};

__device__ float g_pg_zero_page[16];      // zero-initialised: DMA source of the padding taps

__device__ __forceinline__ void pg_dma16_ptr(const float* lane_src, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_src), "s"(lds_dst)
      : "memory");
}

// LEVELS: 2 on the 96-column tile; the 128-column tile keeps single-level chains (64 + 64 accumulator registers plus the gather
// state do not fit the 128-register budget of 4 waves per SIMD), which is what the first-generation direct kernel does everywhere
template <int NT, int RELU, int LEVELS>
__global__ __launch_bounds__(512, 4) void icg_pconv_kernel(PconvP p) {
  constexpr int BM = 128, BN = 32 * NT, BK = 16;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, SLOT = A_BYTES + B_BYTES, NBUF = 3;
  constexpr int BROWS = BN / 8;
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv & 3, wn = wv >> 2;
  const int r = lane & 15, kk = lane >> 4;

  unsigned t = blockIdx.x;
  if (p.swz) {
    const unsigned tot = p.total, q = tot >> 3, rr = tot & 7u, xcd = t & 7u;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
  }
  const int z = (int)(t / (unsigned)p.tiles_mn);                    // phase (phase_mode 1), else 0
  const int tile = (int)(t - (unsigned)z * (unsigned)p.tiles_mn);
  const int nt = tile % p.tiles_n, mt = tile / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;
  const int ph_a = p.phase_mode ? (z >> 1) : 0, ph_b = p.phase_mode ? (z & 1) : 0;
  const int pad_h = p.phase_mode ? 1 - ph_a : p.pad_h, pad_w = p.phase_mode ? 1 - ph_b : p.pad_w;
  const float* __restrict__ Bg = p.B + (long)z * p.strideB;

  // ---- DMA role, A: pixel row 16 wv + (lane >> 2) of the tile, chunk position lane & 3
  const int drowA = 16 * wv + (lane >> 2), drowB = BROWS * wv + (lane >> 2);
  const int am = m0 + drowA;
  const bool a_row_ok = am < p.M;
  const int amm = a_row_ok ? am : 0;
  const int aw = amm % p.W, at = amm / p.W, ah = at % p.H, ab = at / p.H;
  const int hs0 = ah * p.gs - pad_h, ws0 = aw * p.gs - pad_w;       // source coordinate of tap (0, 0)
  const unsigned img = (unsigned)ab * (unsigned)(p.Hs * p.Ws);
  const unsigned achunk = 4u * (unsigned)((lane & 3) ^ pg_swz(drowA));
  const unsigned voffB =
      ((unsigned)min(n0 + min(drowB, BN - 1), p.N - 1) * (unsigned)p.ldb + 4u * (unsigned)((lane & 3) ^ pg_swz(drowB))) * 4u;
  const bool dma_b_lane = (BROWS == 16) || ((lane >> 2) < BROWS);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1024u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * (BROWS * 64u);
  const int nk = p.K / BK;                                          // even (Cin % 16 == 0 and R*R*Cin/16 even: checked by the launcher)

  // load cursor (wave-uniform): K-tile -> (channel slice lc0, tap (ltr, lts)); tap-minor order
  int lc0 = 0, ltr = 0, lts = 0, ltap = 0, lkt = 0;
  auto issue_next = [&](unsigned slot_off) {
    const int hi = hs0 + ltr, wi = ws0 + lts;
    const bool ok = a_row_ok & ((unsigned)hi < (unsigned)p.Hb) & ((unsigned)wi < (unsigned)p.Wb);
    const unsigned pix = img + (unsigned)(hi >> p.up) * (unsigned)p.Ws + (unsigned)(wi >> p.up);
    const float* src = ok ? p.A + ((size_t)pix * (unsigned)p.Cin + (unsigned)lc0 + achunk) : g_pg_zero_page;
    pg_dma16_ptr(src, ldsA + slot_off);
    if (dma_b_lane) pg_dma16(Bg + (ltap * p.Cin + lc0), voffB, ldsB + slot_off);
    if (++lkt < nk) {                                               // past the end: the last K-tile again (never read)
      ++ltap;
      if (++lts == p.R) { lts = 0; ++ltr; }
      if (ltap == p.R * p.R) { ltap = 0; ltr = 0; lts = 0; lc0 += BK; }
    } else {
      lkt = nk;
    }
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  const int fo = r * 64 + ((kk ^ pg_swz(r)) * 16);
  const char* fa = lds + fo + (32 * wm) * 64;
  const char* fb = lds + fo + A_BYTES + (16 * NT * wn) * 64;

  f32x4 acc[2][NT], acc2[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  issue_next(0u);
  issue_next((unsigned)SLOT);

  auto tile_step = [&](unsigned cur, auto flush_c) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next(next_slot(next_slot(cur)));
    float4 a[2], b[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = *reinterpret_cast<const float4*>(fa + cur + i * 1024);
      if (RELU) { a[i].x = fmaxf(a[i].x, 0.f); a[i].y = fmaxf(a[i].y, 0.f); a[i].z = fmaxf(a[i].z, 0.f); a[i].w = fmaxf(a[i].w, 0.f); }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(fb + cur + j * 1024);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float av = s == 0 ? a[i].x : (s == 1 ? a[i].y : (s == 2 ? a[i].z : a[i].w));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = s == 0 ? b[j].x : (s == 1 ? b[j].y : (s == 2 ? b[j].z : b[j].w));
          if (LEVELS == 2 && FLUSH && s == 0) {
            acc2[i][j] += acc[i][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  };
  unsigned cur = 0u;
  for (int kt = 0; kt < nk; kt += 2) {
    tile_step(cur, std::true_type{});
    cur = next_slot(cur);
    tile_step(cur, std::false_type{});
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane (r, kk) holds C[16 i + r][16 j + 4 kk .. + 3] of its wave tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 16 * i + r;
    long out_row = m;
    if (p.phase_mode) {
      const int mm = m < p.M ? m : 0;
      const int w = mm % p.W, tq = mm / p.W, h = tq % p.H, b = tq / p.H;
      const int oh = p.oH ? p.oH : 2 * p.H, ow = p.oW ? p.oW : 2 * p.W;
      const int y = 2 * h + ph_a, x = 2 * w + ph_b;
      out_row = (y < oh && x < ow) ? ((long)b * oh + y) * ow + x : -1;
    }
    const bool row_ok = m < p.M && out_row >= 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 16 * NT * wn + 16 * j + 4 * kk;
      f32x4 v = (LEVELS == 2) ? acc[i][j] + acc2[i][j] : acc[i][j];
      v *= p.alpha;
      if (row_ok && n < p.N) {
        if (p.bias != nullptr) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.res != nullptr) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + out_row * p.ldc + n);
          if (p.res_mode == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rv[e] > 0.f ? v[e] : 0.f;
          } else {
            v += rv;
          }
        }
        *reinterpret_cast<f32x4*>(p.C + out_row * p.ldc + n) = v;
      }
    }
  }
}

static bool pgemm_enabled() {      // measurement switch (ICG_PGEMM=0: first-generation plane GEMMs), read once per process
  static const bool on = [] { const char* e = getenv("ICG_PGEMM"); return !(e && e[0] == '0'); }();
  return on;
}

#ifndef ICG_PGEMM_L1_MINK_DEFAULT
#define ICG_PGEMM_L1_MINK_DEFAULT 0      // 0: two-level chains at every K ("accumulation levels")
#endif
#ifndef ICG_PGEMM_MFMA32_DEFAULT
#define ICG_PGEMM_MFMA32_DEFAULT 0
#endif
static int pgemm_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}

// -> ICG_OK when launched, 1 when the shape is not one this kernel takes (caller falls back to icg_planes_body)
int icg_pgemm_nn_launch(const float* A, const float* B, float* C, int M, int N, int K, long ldc, long sA, long sB, long sC,
                        int planes, float alpha, int levels, hipStream_t st, int* tn_out) {
  if (!pgemm_enabled()) return 1;
  const int nt = (N % 128 == 0) ? 4 : ((N % 96 == 0) ? 3 : 0);
  if (nt == 0 || K % 32 != 0 || M < 1 || planes < 1) return 1;
  if ((uintptr_t)A % 16 || (uintptr_t)B % 16 || sA % 4 || sB % 4) return 1;
  if ((long)M * K >= (1L << 30) || (long)N * K >= (1L << 30)) return 1;        // 32-bit byte offsets inside a plane
  const int tiles_n = N / (32 * nt);
  const long tiles_mn = icg_cdiv(M, 128) * tiles_n, total = tiles_mn * planes;
  if (total <= 0 || total >= 0x7fffffffL) return 1;
  static const bool no_swz = [] { const char* e = getenv("ICG_NO_XCD_SWIZZLE"); return e && e[0] == '1'; }();
  const int swz = (total >= 16 && !no_swz) ? 1 : 0;
  static const int stream = pgemm_env_int("ICG_PGEMM_STREAM", 1);              // measurement switches, read once
  static const int run_ktiles = pgemm_env_int("ICG_PGEMM_RUN_KTILES", 48);
  const bool c_vec = ((uintptr_t)C % 16 == 0) && (ldc % 4 == 0) && (sC % 4 == 0);
  dim3 block(512);
  static const int l1_maxk = pgemm_env_int("ICG_PGEMM_L1_MAXK", 0);           // K up to which chains stay single-level (experiments)
  if (K <= l1_maxk) levels = 1;
  static const int l1_mink = pgemm_env_int("ICG_PGEMM_L1_MINK", ICG_PGEMM_L1_MINK_DEFAULT);      // K from which chains are single-level
  if (l1_mink > 0 && K >= l1_mink) levels = 1;
  if (tn_out) *tn_out = nt + (levels == 1 ? 100 : 0);                          // (+100: single-level chains ran)
  // the 128-column two-level kernel holds 64 + 64 accumulator registers: its streaming form does not fit the 128-register budget
  // of 4 waves per SIMD without spills (scratch traffic would sit on the hand-counted vmcnt), so it keeps one tile per workgroup
  if (stream && c_vec && !(nt == 4 && levels == 2)) {
    // streaming form: every workgroup owns a run of output tiles worth ~run_ktiles K-tiles, as long as the launch still queues
    // several workgroups per CU (2 resident per CU)
    PgemmSP p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.ldc = ldc; p.sA = sA; p.sB = sB; p.sC = sC; p.alpha = alpha;
    p.tiles_n = tiles_n; p.tiles_mn = (int)tiles_mn; p.total = (unsigned)total; p.swz = swz;
    const int nk = K / 16;
    int run = (run_ktiles + nk - 1) / nk;
    if (run > 32) run = 32;
    while (run > 1 && total / run < 2048) --run;
    const long per_xcd = icg_cdiv(icg_cdiv(total, 8), run);
    dim3 grid(swz ? (unsigned)(8 * per_xcd) : (unsigned)icg_cdiv(total, run));
    static const int nbuf = pgemm_env_int("ICG_PGEMM_NBUF", 3);
    if (nt == 4) {
      hipLaunchKernelGGL((icg_pgemm_nn_stream_kernel<4, 1>), grid, block, 0, st, p);       // (levels == 1: see above)
    } else if (nbuf == 4) {
      if (levels == 2) hipLaunchKernelGGL((icg_pgemm_nn_stream_kernel<3, 2, 4>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((icg_pgemm_nn_stream_kernel<3, 1, 4>), grid, block, 0, st, p);
    } else {
      if (levels == 2) hipLaunchKernelGGL((icg_pgemm_nn_stream_kernel<3, 2>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((icg_pgemm_nn_stream_kernel<3, 1>), grid, block, 0, st, p);
    }
    return icg_check_launch();
  }
  PgemmP p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.ldc = ldc; p.sA = sA; p.sB = sB; p.sC = sC;
  p.alpha = alpha;
  p.tiles_n = tiles_n;
  p.tiles_mn = (int)tiles_mn;
  p.total = (unsigned)total;
  p.swz = swz;
  dim3 grid((unsigned)total);
  static const int mfma32 = pgemm_env_int("ICG_PGEMM_MFMA32", ICG_PGEMM_MFMA32_DEFAULT);      // 128-column tile on v_mfma_f32_32x32x2_f32 (A/B: profiles/r05_pgemm_mfma32.txt)
  if (nt == 4 && mfma32) {
    if (levels == 2) hipLaunchKernelGGL((icg_pgemm_nn32_kernel<2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pgemm_nn32_kernel<1>), grid, block, 0, st, p);
  } else if (nt == 4) {
    if (levels == 2) hipLaunchKernelGGL((icg_pgemm_nn_kernel<4, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pgemm_nn_kernel<4, 1>), grid, block, 0, st, p);
  } else {
    if (levels == 2) hipLaunchKernelGGL((icg_pgemm_nn_kernel<3, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pgemm_nn_kernel<3, 1>), grid, block, 0, st, p);
  }
  return icg_check_launch();
}

// -> ICG_OK when launched, 1 when not applicable.  C: slabs [planes][slices][M][N]; kchunk: K-rows per slice
int icg_pgemm_tn_launch(const float* A, const float* B, float* C, int M, int N, int K, long sA, long sB, int planes, int kchunk,
                        int slices, int levels, hipStream_t st, int* tn_out) {
  if (!pgemm_enabled()) return 1;
  const int nt = (N % 128 == 0) ? 4 : ((N % 96 == 0) ? 3 : 0);
  if (nt == 0 || M % 4 != 0 || M < 4 || K % 32 != 0 || kchunk % 32 != 0 || kchunk < 32 || planes < 1 || slices < 1) return 1;
  if ((long)(slices - 1) * kchunk >= K) return 1;
  if ((uintptr_t)A % 16 || (uintptr_t)B % 16 || sA % 4 || sB % 4) return 1;
  if ((long)M * K >= (1L << 30) || (long)N * K >= (1L << 30)) return 1;
  static const int wm3 = pgemm_env_int("ICG_PGEMM_TN_WM3", 1);
  const bool m96 = wm3 && nt == 3 && (M % 96 == 0) && (M % 128 != 0);          // 96-row tiles: no padded rows at M = 96 / 192
  PgemmTnP p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.sA = sA; p.sB = sB;
  p.kchunk = kchunk; p.slices = slices;
  p.tiles_n = N / (32 * nt);
  const long tiles_mn = icg_cdiv(M, m96 ? 96 : 128) * p.tiles_n, total = tiles_mn * planes * slices;
  if (total <= 0 || total >= 0x7fffffffL) return 1;
  p.tiles_mn = (int)tiles_mn;
  p.total = (unsigned)total;
  static const bool no_swz = [] { const char* e = getenv("ICG_NO_XCD_SWIZZLE"); return e && e[0] == '1'; }();
  p.swz = (total >= 16 && !no_swz) ? 1 : 0;
  dim3 grid((unsigned)total), block(512);
  static const int nbuf = pgemm_env_int("ICG_PGEMM_NBUF", 3);
  static const int tn96 = pgemm_env_int("ICG_PGEMM_TN96", 2);       // 2: every M, N multiple of 96 (also 384 / 768 / 1536: 0 - 10 % faster than the 128-tile form);
  //                                                                   1: only where the 128-tile form pads; 0: the 6- / 8-wave forms; 4: 4-slot ring (measurement switches)
  if (tn96 && levels == 2 && M % 96 == 0 && N % 96 == 0 && (tn96 == 2 || M % 128 != 0 || N % 128 != 0)) {
    // 96 x 96 tiles on four waves, four workgroups per CU (see icg_pgemm_tn96_kernel)
    p.tiles_n = N / 96;
    const long tmn = (long)(M / 96) * p.tiles_n, tot2 = tmn * planes * slices;
    if (tot2 > 0 && tot2 < 0x7fffffffL) {
      p.tiles_mn = (int)tmn;
      p.total = (unsigned)tot2;
      p.swz = (tot2 >= 16 && !no_swz) ? 1 : 0;
      if (tn96 == 4) hipLaunchKernelGGL((icg_pgemm_tn96_kernel<2, 4>), dim3((unsigned)tot2), dim3(256), 0, st, p);
      else hipLaunchKernelGGL((icg_pgemm_tn96_kernel<2, 3>), dim3((unsigned)tot2), dim3(256), 0, st, p);
      if (tn_out) *tn_out = 3;
      return icg_check_launch();
    }
  }
  if (m96) {
    block = dim3(384);
    if (levels == 2 && nbuf == 4) hipLaunchKernelGGL((icg_pgemm_tn_kernel<3, 2, 3, 4>), grid, block, 0, st, p);
    else if (levels == 2) hipLaunchKernelGGL((icg_pgemm_tn_kernel<3, 2, 3>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pgemm_tn_kernel<3, 1, 3>), grid, block, 0, st, p);
  } else if (nt == 3 && levels == 2 && nbuf == 4) {
    hipLaunchKernelGGL((icg_pgemm_tn_kernel<3, 2, 4, 4>), grid, block, 0, st, p);
  } else if (nt == 4) {
    if (levels == 2) hipLaunchKernelGGL((icg_pgemm_tn_kernel<4, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pgemm_tn_kernel<4, 1>), grid, block, 0, st, p);
  } else {
    if (levels == 2) hipLaunchKernelGGL((icg_pgemm_tn_kernel<3, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pgemm_tn_kernel<3, 1>), grid, block, 0, st, p);
  }
  if (tn_out) *tn_out = nt;
  return icg_check_launch();
}

// -> ICG_OK when launched, 1 when this kernel does not take the problem (caller continues on icg_gemm_body)
int icg_pconv_launch(const float* A, const float* B, float* C, int M, int N, int K, int H, int W, int Cin, int R, int up, int Hs,
                     int Ws, int pad_h, int pad_w, int gs, int Hb, int Wb, long ldb, long ldc, const float* bias, const float* res,
                     int res_mode, float alpha, long strideB, int phase_mode, int oH, int oW, int pre_relu, int zdim,
                     hipStream_t st, int* tn_out) {
  if (!pgemm_enabled()) return 1;
  static const int on = pgemm_env_int("ICG_PCONV", 1);
  static const long min_tiles = pgemm_env_int("ICG_PCONV_MIN_TILES", 512);
  if (!on) return 1;
  const int nt = (N % 128 == 0) ? 4 : ((N % 96 == 0) ? 3 : 0);
  if (nt == 0 || Cin % 16 != 0 || K != R * R * Cin || (K / 16) % 2 != 0 || R < 1 || R > 4) return 1;
  if (res_mode == 1 || (phase_mode != 0 && (phase_mode != 1 || zdim != 4)) || (phase_mode == 0 && zdim != 1)) return 1;
  if ((uintptr_t)A % 16 || (uintptr_t)B % 16 || (uintptr_t)C % 16 || ldb % 4 || ldc % 4 || strideB % 4) return 1;
  if ((bias && (uintptr_t)bias % 16) || (res && (uintptr_t)res % 16)) return 1;
  if ((long)N * ldb >= (1L << 30)) return 1;
  const int tiles_n = N / (32 * nt);
  const long tiles_mn = icg_cdiv(M, 128) * tiles_n, total = tiles_mn * zdim;
  if (total < min_tiles || total >= 0x7fffffffL) return 1;         // small launches: split-K / the first-generation kernel
  PconvP p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.H = H; p.W = W; p.Cin = Cin; p.R = R; p.up = up; p.Hs = Hs; p.Ws = Ws;
  p.pad_h = pad_h; p.pad_w = pad_w; p.gs = gs; p.Hb = Hb; p.Wb = Wb;
  p.ldb = ldb; p.ldc = ldc; p.bias = bias; p.res = res; p.res_mode = res_mode; p.alpha = alpha;
  p.strideB = strideB; p.phase_mode = phase_mode; p.oH = oH; p.oW = oW; p.pre_relu = pre_relu;
  p.tiles_n = tiles_n; p.tiles_mn = (int)tiles_mn; p.total = (unsigned)total;
  static const bool no_swz = [] { const char* e = getenv("ICG_NO_XCD_SWIZZLE"); return e && e[0] == '1'; }();
  p.swz = (total >= 16 && !no_swz) ? 1 : 0;
  dim3 grid((unsigned)total), block(512);
  if (nt == 4) {
    if (pre_relu) hipLaunchKernelGGL((icg_pconv_kernel<4, 1, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pconv_kernel<4, 0, 1>), grid, block, 0, st, p);
  } else {
    if (pre_relu) hipLaunchKernelGGL((icg_pconv_kernel<3, 1, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((icg_pconv_kernel<3, 0, 2>), grid, block, 0, st, p);
  }
  if (tn_out) *tn_out = nt;
  return icg_check_launch();
}
