// fp32 MFMA implicit-GEMM for gfx950: fused convolution (fprop / dgrad / wgrad), every Linear and the
// attention contractions of the IC-GAN BigGAN G+D step.
//
// Replaces cuDNN/ATen behind F.conv2d / F.linear / torch.bmm at BigGAN_PyTorch/layers.py:144-153,
// 164-165, 237-243 together with the elementwise glue around them (BN apply, ReLU, nearest upsample,
// residual add: layers.py:542-552, 587-613).
//
// Design (MI355X-first):
//   * one workgroup = 4 wavefronts (64 lanes each) computes a 128 x (32*TN) tile of C; wave w owns rows
//     32w..32w+31 and TN accumulators of v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain)
//   * both operands are staged K-major in LDS ([k][m], [k][n]) so every MFMA operand fetch is one
//     conflict-free ds_read_b32 of 32 consecutive floats per half-wave
//   * global->register prefetch of tile t+1 is issued before the 8*TN MFMAs of tile t (2 LDS buffers,
//     one barrier per K-tile); the BN/ccbn affine, ReLU, zero padding and the nearest-upsample index
//     map are applied in registers on the way into LDS, so the normalised / upsampled tensor never
//     exists in HBM
//   * NHWC activations: the K-slice of an im2col row is contiguous (coalesced 16-byte loads)
#include "icg_common.h"
#include <type_traits>

#ifndef ICG_PLANES_BLOCKED
#define ICG_PLANES_BLOCKED 1     // two-level accumulation in the Winograd-plane GEMMs (0: ablation build of tools/)
#endif
#ifndef ICG_PLANES_PERSISTENT
#define ICG_PLANES_PERSISTENT 1  // forward plane GEMMs on icg_planes_body (0: one output tile per workgroup; ablation build)
#endif
#ifndef ICG_PLANES_FLUSH_TILES
#define ICG_PLANES_FLUSH_TILES 2 // K-tiles (of 16) per first-level chain
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { A_K = 0, A_M = 1 };  // A: K-contiguous rows (im2col gather) | M-contiguous rows (transposed gather)
enum { B_K = 0, B_N = 1 };  // B: [N][K] | [K][N]

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int H, W, Cin, R, up, Hs, Ws;  // gather geometry of the A operand (pixel grid = conv OUTPUT grid)
  const float* scale;
  const float* shift;
  long ss_bstride;
  int pre_affine, pre_relu;
  long ldb, ldc;
  const float* bias;
  const float* res;
  int res_up;
  float alpha;
  long strideA, strideB, strideC;  // per blockIdx.z
  int kchunk;                      // >0: split-K, z selects the K range
  int ntiles_n;
  int lw, lh;                      // log2(W), log2(H) (fast path: H, W powers of two)
  // generalised gather geometry: source coordinate = pixel*gs + tap - pad, valid inside [0,Hb) x [0,Wb)
  int pad_h, pad_w, gs, Hb, Wb;
  // phase decomposition of the nearest-x2-upsample-fused 3x3 convolution (4 phases of 2x2 taps at source resolution):
  //   1: fprop  - blockIdx.z = phase (al, be); pad = (1-al, 1-be); C rows scatter to (2h+al, 2w+be); B += z*strideB
  //   2: wgrad  - blockIdx.z = phase*nsplit + split; pad as above; B rows gather from (2h+al, 2w+be)
  int phase_mode, nsplit;
  int oH, oW;                      // phase_mode 1: extent of the scattered output grid (0 = 2H x 2W); rows beyond are dropped
  int swz;                         // XCD-aware tile order (see kernel head)
  int bsplit;                      // > 0: batched split-K, slices per batch (see kernel head)
  int zmask;                       // generic A_K paths only: source is zero-inserted by (zmask+1): hi, wi must be multiples
  int vec_b;                       // PATH 1 only: 16-byte loads allowed on the B operand (A is vectorised)
  // persistent plane GEMM (icg_planes_body): output tiles per plane, planes, consecutive output tiles per workgroup
  int pt_tiles, pt_z, pt_run;      // (pt_run is informational: the grid size fixes the tiles per workgroup)
  int plain;                       // A [M][K] x B [N][K]^T batched GEMM outside the Winograd composites (attention Q K^T, dO V^T, kNN Gram):
                                   // takes the persistent body too, with single-level chains as on the generic kernel
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// PATH 0: element-wise gather (any Cin); 1: 16-byte loads, generic index decode; 2: 16-byte loads, 32-bit offsets,
// branch-free loads (clamped address + select), scalar tap tracking (A_K: Cin % 16 == 0) / shift-mask pixel decode
// (A_M: H, W powers of two).  PATH 2 cuts the per-K-tile address section from ~330 to ~90 instructions.
// BLK = 1: two-level ("blocked") accumulation.  The MFMA is a k-ordered fp32 fmaf chain; in the Winograd domain the partial
// sums are ~40x larger than the result they cancel to in the output transform, so the rounding of a chain of length K = Cin
// (forward / data gradient) or K = tiles-per-slice (weight gradient) IS the Winograd error (tools/winograd_error_model.py:
// 3.5e-6 -> 1.2e-6 per layer at C = 384).  Every 2 K-tiles (32 k-values) the first MFMA of the pair starts a fresh chain
// from C = 0 and the finished chain is added into a second accumulator set by VALU adds that sit in the shadow of the
// preceding MFMA (their operands were produced TN MFMAs earlier: no dependency stall).  Cost: 16*TN more registers.
template <int AMODE, int BMODE, int TN, int PATH, int BLK = 0, int PLANES = 0>
__device__ __forceinline__ void icg_gemm_body(const GemmP& p) {
  // PATH 3 = PATH 2 with the BN/ccbn affine prologue compiled in (PATH 2 itself has none): keeps the hot loop
  // free of uniform branches so that the scheduler can interleave the staging work with the MFMAs
  constexpr bool VEC = PATH >= 1;
  constexpr bool FAST = PATH >= 2;
  constexpr bool FAST_AFFINE = PATH == 3;
  // PLAIN: the batched GEMMs over Winograd planes (A [M][K], B [N][K], H = W = R = 1, no prologue, no split-K): the loader
  // keeps one row offset per staged row instead of the convolution gather state (image base, pixel coordinates, tap tracker,
  // validity flags) -- ~12 fewer live VGPRs and no per-tile address arithmetic beyond one add.  Rows >= M / >= N are clamped
  // to the last row, not zeroed: they only feed outputs the epilogue masks.
  constexpr bool PLAIN = (PLANES != 0) && AMODE == A_K && BMODE == B_K && PATH == 2;
  // PLAIN_M: the same for the weight-gradient plane GEMMs (A [K][M], B [K][N], K = tiles of a split-K slice): row index
  // k0 + krow + 8 i clamped into the slice, A rows past its end zeroed (B rows there are then irrelevant)
  constexpr bool PLAIN_M = (PLANES != 0) && AMODE == A_M && BMODE == B_N && PATH == 2;
  constexpr int BM = 128, BN = 32 * TN, BK = 16;
  constexpr int LDA = (AMODE == A_K) ? BM + 1 : BM + 4;
  constexpr int LDB = (BMODE == B_K) ? BN + 1 : BN + 4;
  constexpr int NBUF = 3;   // 3-deep LDS ring: lets the single barrier per K-tile sit mid-tile (see main loop)
  __shared__ __attribute__((aligned(16))) float As[NBUF][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK * LDB];

  const int tid = threadIdx.x;
  // XCD-aware tile order.  The dispatcher places workgroup `lin` on XCD lin % 8, each XCD with a private 4 MiB L2.
  // Giving every XCD one CONTIGUOUS range of the (z, m-tile, n-tile) order keeps the tiles that share operand panels
  // (all n-tiles of an m-tile, neighbouring pixel rows and their 3x3 halos, all tiles of one split-K / phase plane) on one
  // L2 instead of fetching each panel into all eight.  Bijective for any grid size (cdna_hip_programming.md T1);
  // placement is a speed choice only — results do not depend on it.
  unsigned tile = blockIdx.x, zz = blockIdx.z;
  if (p.swz) {
    const unsigned nwg = gridDim.x, lin = blockIdx.x + blockIdx.z * nwg, tot = nwg * gridDim.z;
    const unsigned q = tot >> 3, r = tot & 7u, xcd = lin & 7u;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    zz = t / nwg;
    tile = t - zz * nwg;
  }
  const int nt = tile % p.ntiles_n;
  const int mt = tile / p.ntiles_n;
  const int z = (int)zz;
  const int m0 = mt * BM, n0 = nt * BN;
  // batched split-K (p.bsplit > 0): z = batch * bsplit + slice; operands advance per batch, the output per (batch, slice)
  const int zb = (p.bsplit > 0) ? z / p.bsplit : z;
  const float* __restrict__ Ag = p.A + (long)zb * p.strideA;
  const float* __restrict__ Bg = p.B + (long)zb * p.strideB;
  float* __restrict__ Cg = p.C + (long)z * p.strideC;
  int kbeg = 0, kend = p.K;
  int ksplit = (p.bsplit > 0) ? z - zb * p.bsplit : z, ph_a = 0, ph_b = 0;
  int pad_h = p.pad_h, pad_w = p.pad_w;
  if (p.phase_mode) {
    const int phase = (p.phase_mode == 2) ? z / p.nsplit : z;
    ksplit = (p.phase_mode == 2) ? z - phase * p.nsplit : 0;
    ph_a = phase >> 1;
    ph_b = phase & 1;
    pad_h = 1 - ph_a;
    pad_w = 1 - ph_b;
  }
  if (p.kchunk > 0) {
    kbeg = ksplit * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
  }
  const int gs = p.gs;
  const int Cin = p.Cin;
  // row of the (2H x 2W) grid that phase (ph_a, ph_b) of source pixel index q = (b, h, w) maps to
  auto phase_row = [&](int q) -> long {
    const int w = q % p.W;
    const int t = q / p.W;
    const int h = t % p.H;
    const int b = t / p.H;
    const int oh = p.oH ? p.oH : 2 * p.H, ow = p.oW ? p.oW : 2 * p.W;
    const int y = 2 * h + ph_a, x = 2 * w + ph_b;
    if (y >= oh || x >= ow) return -1;
    return ((long)b * oh + y) * ow + x;
  };

  // ------------------------------------------------------------------ loader state
  float4 ra[2], rsc[2], rsh[2], rb[2];
  // A_K : thread -> (k quad kq, rows arow + 64 i);  A_M : thread -> (m quad mq, k rows krow + 8 i)
  const int kq = tid & 3, arow = tid >> 2;
  const int mq = tid & 31, krow = tid >> 5;
  int ab[2], ah[2], aw[2];
  bool amv[2];
  int am_tap_r[4], am_tap_s[4], am_c[4];
  bool am_mv[4];
  if (AMODE == A_K) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = m0 + arow + 64 * i;
      amv[i] = m < p.M;
      int mm = amv[i] ? m : 0;
      aw[i] = mm % p.W;
      int t = mm / p.W;
      ah[i] = t % p.H;
      ab[i] = t / p.H;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int m = m0 + 4 * mq + j;
      am_mv[j] = m < p.M;
      int mm = am_mv[j] ? m : 0;
      int tap = mm / Cin;
      am_c[j] = mm - tap * Cin;
      am_tap_r[j] = tap / p.R;
      am_tap_s[j] = tap - am_tap_r[j] * p.R;
    }
  }

  // ---- PATH 2 state -------------------------------------------------------------------------------------
  unsigned f_img[2], f_ss[2], f_boff[2];
  int f_h[2], f_w[2];
  bool f_rowok[2], f_ok[2];
  int f_tr = 0, f_ts = 0, f_c0 = 0;          // wave-uniform tap tracking (A_K)
  int f_btap = 0, f_bc0 = 0;                 // the same position, for the B operand (advanced by load_B_fast)
  if (FAST && AMODE == A_K && kbeg > 0) {    // split-K: K-tile kbeg/BK of the tap-minor order = (slice, tap)
    const int it0 = kbeg / 16, RRt = p.R * p.R;
    const int sl = it0 / RRt, tap0 = it0 - sl * RRt;
    f_c0 = f_bc0 = sl * 16;
    f_btap = tap0;
    f_tr = tap0 / p.R;
    f_ts = tap0 - f_tr * p.R;
  }
  unsigned pl_a[2] = {0u, 0u}, pl_b[2] = {0u, 0u};   // PLAIN: row offset + 4 * kq of this thread's two A / B rows
  if (PLAIN) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = min(m0 + arow + 64 * i, p.M - 1);
      const int n = min(n0 + min(arow + 64 * i, BN - 1), p.N - 1);
      pl_a[i] = (unsigned)m * (unsigned)p.K + 4u * (unsigned)kq;
      pl_b[i] = (unsigned)n * (unsigned)p.ldb + 4u * (unsigned)kq;
    }
  }
  unsigned pm_a = 0u, pm_b = 0u;               // PLAIN_M: first of this thread's 4 columns of A / B (clamped into the matrix)
  if (PLAIN_M) {
    pm_a = (unsigned)min(m0 + 4 * mq, max(p.M - 4, 0));
    pm_b = (unsigned)min(n0 + min(4 * mq, BN - 4), max(p.N - 4, 0));
  }
  unsigned f_mtap_c = 0;                      // A_M: channel of this thread's 4 columns
  int f_mr = 0, f_ms = 0;
  bool f_mok = false;
  if (FAST && !PLAIN && !PLAIN_M) {
    if (AMODE == A_K) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f_rowok[i] = amv[i];
        f_h[i] = ah[i];
        f_w[i] = aw[i];
        f_img[i] = (unsigned)ab[i] * (unsigned)(p.Hs * p.Ws);
        f_ss[i] = (unsigned)ab[i] * (unsigned)p.ss_bstride;
      }
    } else {
      f_mok = am_mv[0];
      f_mtap_c = (unsigned)am_c[0];
      f_mr = am_tap_r[0];
      f_ms = am_tap_s[0];
    }
    if (BMODE == B_K) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int nl = arow + 64 * i;
        const int n = n0 + nl;
        f_boff[i] = (nl < BN && n < p.N) ? (unsigned)n * (unsigned)p.ldb : 0u;
      }
    }
  }

  // one row (i = 0 / 1) of this thread's A staging; `last` advances the tap tracker
  // staging row i (0/1) of this thread's A data, split into an address piece and a load piece so that each fits
  // into one MFMA shadow; `last` advances the tap tracker
  unsigned f_idx[2], f_so[2];
  auto addr_A_fast = [&](int k0, int i, bool last) {
    if (PLAIN) {
      f_idx[i] = pl_a[i] + (unsigned)min(k0, p.K - BK);      // the pipeline over-fetches one tile past the end: clamp
    } else if (PLAIN_M) {
      const int kp = k0 + krow + 8 * i;
      f_ok[i] = kp < kend;
      f_idx[i] = (unsigned)min(kp, kend - 1) * (unsigned)Cin + pm_a;
    } else if (AMODE == A_K) {
      const unsigned c = (unsigned)(f_c0 + 4 * kq);
      const int hi = f_h[i] * gs + f_tr - pad_h, wi = f_w[i] * gs + f_ts - pad_w;
      // (f_c0 < Cin fails only on the tile the pipeline over-fetches past the end of K)
      const bool ok = f_rowok[i] & (f_c0 < Cin) & ((unsigned)hi < (unsigned)p.Hb) & ((unsigned)wi < (unsigned)p.Wb);
      const unsigned idx =
          (f_img[i] + (unsigned)(hi >> p.up) * (unsigned)p.Ws + (unsigned)(wi >> p.up)) * (unsigned)Cin + c;
      f_ok[i] = ok;
      f_idx[i] = ok ? idx : 0u;
      f_so[i] = ok ? f_ss[i] + c : 0u;
      if (last) {
        // K order of the fast path: TAP-MINOR — all R*R taps of one 16-channel slice, then the next slice (a K-tile never
        // straddles a tap: Cin % BK == 0).  The R*R shifted re-reads of an activation element are then ~R*R K-tiles apart
        // (a few KB of footprint per workgroup) instead of a full Cin sweep apart (hundreds of KB), so they hit L1/L2
        // instead of going back to HBM: measured fetch traffic of the 3x3 layers drops accordingly.
        if (++f_ts == p.R) {
          f_ts = 0;
          if (++f_tr == p.R) { f_tr = 0; f_c0 += BK; }
        }
      }
    } else {
      const int kp = k0 + krow + 8 * i;
      const int w = kp & (p.W - 1);
      const int h = (kp >> p.lw) & (p.H - 1);
      const int b = kp >> (p.lw + p.lh);
      const int hi = h * gs + f_mr - pad_h, wi = w * gs + f_ms - pad_w;
      const bool ok = f_mok & (kp < kend) & ((unsigned)hi < (unsigned)p.Hb) & ((unsigned)wi < (unsigned)p.Wb);
      const unsigned idx = (((unsigned)b * (unsigned)p.Hs + (unsigned)(hi >> p.up)) * (unsigned)p.Ws +
                            (unsigned)(wi >> p.up)) * (unsigned)Cin + f_mtap_c;
      f_ok[i] = ok;
      f_idx[i] = ok ? idx : 0u;
      f_so[i] = ok ? (unsigned)b * (unsigned)p.ss_bstride + f_mtap_c : 0u;
    }
  };
  auto issue_A_fast = [&](int i) {
#if defined(ICG_DBG_LOAD_FIXED)
    ra[i] = ld4(Ag + ((tid * 4) & 1023));
#elif defined(ICG_DBG_ADDR_ONLY)
    { unsigned keep = f_idx[i]; asm volatile("" ::"v"(keep)); ra[i] = zero4(); }
#else
    ra[i] = ld4(Ag + f_idx[i]);
#endif
    if (FAST_AFFINE) {
      rsc[i] = ld4(p.scale + f_so[i]);
      rsh[i] = ld4(p.shift + f_so[i]);
    }
  };
  auto load_A_fast = [&](int k0, int i, bool last) {
    addr_A_fast(k0, i, last);
    issue_A_fast(i);
  };

  auto load_B_fast = [&](int k0) {
    if (PLAIN) {
      const unsigned kc = (unsigned)min(k0, p.K - BK);
#pragma unroll
      for (int i = 0; i < 2; ++i) rb[i] = ld4(Bg + (pl_b[i] + kc));
    } else if (PLAIN_M) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        rb[i] = ld4(Bg + ((unsigned)min(k0 + krow + 8 * i, kend - 1) * (unsigned)p.ldb + pm_b));   // A is zero past kend
    } else if (BMODE == B_K) {
      // tap-minor K order (see addr_A_fast): tile -> (16-channel slice f_bc0, tap f_btap); the weight matrix keeps its
      // [N][tap][Cin] layout, only the order in which its K-tiles are visited changes.  Clamp: the pipeline over-fetches
      // one tile past the end.
      unsigned kg;
      if (AMODE == A_K) {
        kg = (unsigned)(min(f_btap * Cin + f_bc0, p.K - BK) + 4 * kq);
        if (++f_btap == p.R * p.R) { f_btap = 0; f_bc0 += BK; }
      } else {
        kg = (unsigned)(min(k0, kend - BK) + 4 * kq);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#if defined(ICG_DBG_LOAD_FIXED)
        rb[i] = ld4(Bg + ((tid * 4) & 1023));
#elif defined(ICG_DBG_ADDR_ONLY)
        { unsigned keep = f_boff[i] + kg; asm volatile("" ::"v"(keep)); rb[i] = zero4(); }
#else
        rb[i] = ld4(Bg + (f_boff[i] + kg));   // rows >= N only feed masked outputs
#endif
      }
    } else {
      const int n = n0 + 4 * mq;
      const bool nok = (4 * mq < BN) && n < p.N;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kp = k0 + krow + 8 * i;
        const bool kok = nok & (kp < kend);
        const unsigned brow = (p.phase_mode == 2) ? (unsigned)phase_row(kok ? kp : 0) : (unsigned)kp;
        const unsigned idx = kok ? brow * (unsigned)p.ldb + (unsigned)n : 0u;
        rb[i] = ld4(Bg + idx);                                          // A is zero for kp >= kend
      }
    }
  };

  auto src_index = [&](int b, int hi, int wi, int c) -> long {
    return (((long)b * p.Hs + (hi >> p.up)) * p.Ws + (wi >> p.up)) * Cin + c;
  };

  auto load_A = [&](int k0) {
    if (AMODE == A_K) {
      const int kg = k0 + 4 * kq;
      if (VEC) {
        const bool kv = kg < kend;
        const int tap = kv ? kg / Cin : 0;
        const int c = kg - tap * Cin;
        const int r = tap / p.R, s = tap - r * p.R;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int hi = ah[i] * gs + r - pad_h, wi = aw[i] * gs + s - pad_w;
          const bool ok = kv && amv[i] && hi >= 0 && hi < p.Hb && wi >= 0 && wi < p.Wb && (((hi | wi) & p.zmask) == 0);
          ra[i] = zero4();
          rsc[i] = zero4();
          rsh[i] = zero4();
          if (ok) {
            ra[i] = ld4(Ag + src_index(ab[i], hi, wi, c));
            if (p.pre_affine) {
              rsc[i] = ld4(p.scale + (long)ab[i] * p.ss_bstride + c);
              rsh[i] = ld4(p.shift + (long)ab[i] * p.ss_bstride + c);
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[4], sc[4], sh[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = 0.f; sc[j] = 0.f; sh[j] = 0.f;
            const int kj = kg + j;
            if (kj < kend && amv[i]) {
              const int tap = kj / Cin;
              const int c = kj - tap * Cin;
              const int r = tap / p.R, s = tap - r * p.R;
              const int hi = ah[i] * gs + r - pad_h, wi = aw[i] * gs + s - pad_w;
              if (hi >= 0 && hi < p.Hb && wi >= 0 && wi < p.Wb && (((hi | wi) & p.zmask) == 0)) {
                v[j] = Ag[src_index(ab[i], hi, wi, c)];
                if (p.pre_affine) {
                  sc[j] = p.scale[(long)ab[i] * p.ss_bstride + c];
                  sh[j] = p.shift[(long)ab[i] * p.ss_bstride + c];
                }
              }
            }
          }
          ra[i] = make_float4(v[0], v[1], v[2], v[3]);
          rsc[i] = make_float4(sc[0], sc[1], sc[2], sc[3]);
          rsh[i] = make_float4(sh[0], sh[1], sh[2], sh[3]);
        }
      }
    } else {  // A_M : rows are pixels (k), columns are (tap, c) (m)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kp = k0 + krow + 8 * i;
        ra[i] = zero4();
        rsc[i] = zero4();
        rsh[i] = zero4();
        if (kp < kend) {
          const int w = kp % p.W;
          const int t = kp / p.W;
          const int h = t % p.H;
          const int b = t / p.H;
          if (VEC) {
            const int hi = h * gs + am_tap_r[0] - pad_h, wi = w * gs + am_tap_s[0] - pad_w;
            if (am_mv[0] && hi >= 0 && hi < p.Hb && wi >= 0 && wi < p.Wb) {
              ra[i] = ld4(Ag + src_index(b, hi, wi, am_c[0]));
              if (p.pre_affine) {
                rsc[i] = ld4(p.scale + (long)b * p.ss_bstride + am_c[0]);
                rsh[i] = ld4(p.shift + (long)b * p.ss_bstride + am_c[0]);
              }
            }
          } else {
            float v[4], sc[4], sh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = 0.f; sc[j] = 0.f; sh[j] = 0.f;
              const int hi = h * gs + am_tap_r[j] - pad_h, wi = w * gs + am_tap_s[j] - pad_w;
              if (am_mv[j] && hi >= 0 && hi < p.Hb && wi >= 0 && wi < p.Wb) {
                v[j] = Ag[src_index(b, hi, wi, am_c[j])];
                if (p.pre_affine) {
                  sc[j] = p.scale[(long)b * p.ss_bstride + am_c[j]];
                  sh[j] = p.shift[(long)b * p.ss_bstride + am_c[j]];
                }
              }
            }
            ra[i] = make_float4(v[0], v[1], v[2], v[3]);
            rsc[i] = make_float4(sc[0], sc[1], sc[2], sc[3]);
            rsh[i] = make_float4(sh[0], sh[1], sh[2], sh[3]);
          }
        }
      }
    }
  };

  const float relu_floor = p.pre_relu ? 0.f : -INFINITY;
  auto act4 = [&](float4 v, float4 sc, float4 sh) -> float4 {
    if (FAST) {
      if (FAST_AFFINE) {
        v.x = fmaf(v.x, sc.x, sh.x);
        v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z);
        v.w = fmaf(v.w, sc.w, sh.w);
      }
      v.x = fmaxf(v.x, relu_floor);
      v.y = fmaxf(v.y, relu_floor);
      v.z = fmaxf(v.z, relu_floor);
      v.w = fmaxf(v.w, relu_floor);
      return v;
    }
    if (p.pre_affine) {
      v.x = fmaf(v.x, sc.x, sh.x);
      v.y = fmaf(v.y, sc.y, sh.y);
      v.z = fmaf(v.z, sc.z, sh.z);
      v.w = fmaf(v.w, sc.w, sh.w);
    }
    if (p.pre_relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    return v;
  };

  float4 sv[2];
  auto prep_A_row = [&](int i) {
    if (PLAIN) { sv[i] = ra[i]; return; }
    if (PLAIN_M) { sv[i] = f_ok[i] ? ra[i] : zero4(); return; }
    float4 v = act4(ra[i], rsc[i], rsh[i]);
    if (FAST && !f_ok[i]) v = zero4();
    sv[i] = v;
  };
  auto write_A_row = [&](int buf, int i) {
    float* as = As[buf];
    const float4 v = sv[i];
    if (AMODE == A_K) {
      const int row = arow + 64 * i;
      as[(4 * kq + 0) * LDA + row] = v.x;
      as[(4 * kq + 1) * LDA + row] = v.y;
      as[(4 * kq + 2) * LDA + row] = v.z;
      as[(4 * kq + 3) * LDA + row] = v.w;
    } else {
      *reinterpret_cast<float4*>(&as[(krow + 8 * i) * LDA + 4 * mq]) = v;
    }
  };
  auto store_A_row = [&](int buf, int i) {
    float* as = As[buf];
    float4 v = (PLAIN || PLAIN_M) ? ra[i] : act4(ra[i], rsc[i], rsh[i]);
    if (FAST && !PLAIN && !f_ok[i]) v = zero4();
    if (AMODE == A_K) {
      const int row = arow + 64 * i;
      as[(4 * kq + 0) * LDA + row] = v.x;
      as[(4 * kq + 1) * LDA + row] = v.y;
      as[(4 * kq + 2) * LDA + row] = v.z;
      as[(4 * kq + 3) * LDA + row] = v.w;
    } else {
      *reinterpret_cast<float4*>(&as[(krow + 8 * i) * LDA + 4 * mq]) = v;
    }
  };
  auto store_A = [&](int buf) {
    store_A_row(buf, 0);
    store_A_row(buf, 1);
  };

  auto load_B = [&](int k0) {
    if (BMODE == B_K) {
      const int kg = k0 + 4 * kq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int nl = arow + 64 * i;
        const int n = n0 + nl;
        rb[i] = zero4();
        if (nl < BN && n < p.N) {
          const float* src = Bg + (long)n * p.ldb + kg;
          if (VEC && p.vec_b) {
            if (kg < kend) rb[i] = ld4(src);
          } else {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (kg + j < kend) ? src[j] : 0.f;
            rb[i] = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    } else {
      const int nl = 4 * mq;
      const int n = n0 + nl;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kp = k0 + krow + 8 * i;
        rb[i] = zero4();
        if (nl < BN && kp < kend) {
          const long brow = (p.phase_mode == 2) ? phase_row(kp) : (long)kp;
          const float* src = Bg + brow * p.ldb + n;
          if (VEC && p.vec_b) {
            if (n < p.N) rb[i] = ld4(src);
          } else {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (n + j < p.N) ? src[j] : 0.f;
            rb[i] = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  };

  auto store_B_row = [&](int buf, int i) {
    float* bs = Bs[buf];
    if (BMODE == B_K) {
      const int nl = arow + 64 * i;
      if (BN == 128 || nl < BN) {
        bs[(4 * kq + 0) * LDB + nl] = rb[i].x;
        bs[(4 * kq + 1) * LDB + nl] = rb[i].y;
        bs[(4 * kq + 2) * LDB + nl] = rb[i].z;
        bs[(4 * kq + 3) * LDB + nl] = rb[i].w;
      }
    } else {
      if (BN == 128 || 4 * mq < BN) *reinterpret_cast<float4*>(&bs[(krow + 8 * i) * LDB + 4 * mq]) = rb[i];
    }
  };
  auto store_B = [&](int buf) {
    store_B_row(buf, 0);
    store_B_row(buf, 1);
  };

  // ------------------------------------------------------------------ main loop
  f32x16 acc[TN];
  f32x16 acc2[BLK ? TN : 1];       // BLK: the finished 32-deep chains (second accumulation level)
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  if (BLK) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  }

  const int lane = tid & 63, wv = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nk = (kend - kbeg + BK - 1) / BK;

  auto load_A_any = [&](int k0, int half) {      // half 0 / 1: first / second staged row of this thread
    if (FAST) load_A_fast(k0, half, half == 1);
    else if (half == 0) load_A(k0);               // generic paths stage both rows at once
  };
  auto load_B_any = [&](int k0) {
    if (FAST) load_B_fast(k0); else load_B(k0);
  };

  // Software pipeline over K-tiles, ONE barrier per tile, 3-deep LDS ring (tile kt lives in buffer kt % 3):
  //   on loop entry the registers hold tile kt+1 (global loads issued one tile earlier) and the MFMA operand
  //   fragments of k-step 0 of tile kt are already in registers.
  //   k-step 0-1 : MFMAs  ||  write tile kt+1 into ring slot (kt+1)%3
  //   barrier    : tile kt+1 is complete in LDS (and every wave is done with tile kt-1, whose slot is reused next)
  //   k-step 2-4 : MFMAs  ||  issue the global loads of tile kt+2 (row 0, row 1, B)
  //   k-step 7   : MFMAs  ||  fetch the k-step-0 fragments of tile kt+1  -> the MFMA stream never drains at a
  //                tile boundary; all staging work sits in the 64-cycle shadows of the MFMAs.
  // Past the last tile the staging work is harmless (idle ring slot / clamped or masked addresses).
  float fa[2], fb[2][TN];
  if (nk > 0) {
    load_A_any(kbeg, 0);
    load_A_any(kbeg, 1);
    load_B_any(kbeg);
    store_A(0);
    store_B(0);
    load_A_any(kbeg + BK, 0);
    load_A_any(kbeg + BK, 1);
    load_B_any(kbeg + BK);
    __syncthreads();
    fa[0] = As[0][32 * wv + li + lh * LDA];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = Bs[0][li + lh * LDB + 32 * j];
  }
  int cur = 0;
  // one K-tile; FLUSH (compile time): its first k-step starts fresh chains (see BLK above)
  auto tile_body = [&](int kt, auto flush_c) {
    constexpr bool FLUSH = decltype(flush_c)::value;
    const int nxt = (cur == NBUF - 1) ? 0 : cur + 1;
    const float* as = As[cur] + 32 * wv + li + lh * LDA;
    const float* bs = Bs[cur] + li + lh * LDB;
    const float* asn = As[nxt] + 32 * wv + li + lh * LDA;
    const float* bsn = Bs[nxt] + li + lh * LDB;
    const int k2 = kbeg + (kt + 2) * BK;
    // 8*TN micro-slices: MFMA m = (k-step t, column tile j) is followed by ONE small piece of staging work, pinned
    // there with sched_barrier.  A wave issues in order and an MFMA only shadows what sits between it and the next
    // MFMA (~64 cycles), so the staging instructions must be spread evenly over the MFMA gaps, not clumped.
    //   pieces 0-3: tile kt+1 A rows (activation math | LDS write)   4,5: B rows -> LDS   6: barrier
    //   pieces 7-10: tile kt+2 A rows (address math | global loads)   11: B loads
    //   every gap also fetches one operand fragment of the next k-step (after step 7: of tile kt+1)
#pragma unroll
    for (int t = 0; t < BK / 2; ++t) {
      const int pn = (t + 1) & 1;
      const bool wrap = (t + 1 == BK / 2);
      const float* fas = wrap ? asn : as + (2 * t + 2) * LDA;
      const float* fbs = wrap ? bsn : bs + (2 * t + 2) * LDB;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (FLUSH && t == 0) {
          // fold the finished chain into the second level, restart from C = 0 (inline constant: no register zeroing)
          asm volatile("" : "+a"(acc[j]));    // ... and the accumulator -> VGPR copies must not be hoisted to the loop head
          // 8 packed adds (v_pk_add_f32) per column tile, written as asm: the vector add is otherwise legalised into 13 scalar
          // + 1.5 packed adds per tile, and -- being pure, with a result nobody reads before the next flush -- sunk to the
          // end of the loop body, which keeps all 16*TN accumulator copies live in VGPRs meanwhile
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            f32x2 a2 = {acc[j][2 * r], acc[j][2 * r + 1]};
            f32x2 b2 = {acc2[j][2 * r], acc2[j][2 * r + 1]};
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b2) : "v"(a2));
            acc2[j][2 * r] = b2.x;
            acc2[j][2 * r + 1] = b2.y;
          }
          f32x16 zero;
#pragma unroll
          for (int r = 0; r < 16; ++r) zero[r] = 0.f;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1], fb[t & 1][j], zero, 0, 0, 0);
        } else {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1], fb[t & 1][j], acc[j], 0, 0, 0);
        }
        // operand fragments of the next k-step: A with the first MFMA, B as PAIRS of column tiles (adjacent loads 32 dwords
        // apart merge into one ds_read2_b32: 8 + 8*ceil(TN/2) LDS instructions per tile instead of 8 + 8*TN)
        if (j == 0) fa[wrap ? 0 : pn] = fas[0];
        if ((j & 1) == 0) {
          fb[wrap ? 0 : pn][j] = fbs[32 * j];
          if (j + 1 < TN) fb[wrap ? 0 : pn][j + 1] = fbs[32 * (j + 1)];
        }
        // 12 staging pieces over the 8*TN MFMA gaps (PPG pieces per gap; 1 for TN >= 2)
        constexpr int PPG = (8 * TN >= 12) ? 1 : 2;
        const int m = t * TN + j;
#pragma unroll
        for (int q = m * PPG; q < (m + 1) * PPG; ++q) {
#ifndef ICG_DBG_SKIP_LDSW     // (ablation builds of tools/conv_bench.py; never defined in the product)
          if (q == 0) prep_A_row(0);
          if (q == 1) write_A_row(nxt, 0);
          if (q == 2) prep_A_row(1);
          if (q == 3) write_A_row(nxt, 1);
          if (q == 4) store_B_row(nxt, 0);
          if (q == 5) store_B_row(nxt, 1);
#endif
          if (q == 6) {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
          }
#ifndef ICG_DBG_SKIP_GLOAD
          if (FAST) {
            if (q == 7) addr_A_fast(k2, 0, false);
            if (q == 8) issue_A_fast(0);
            if (q == 9) addr_A_fast(k2, 1, true);
            if (q == 10) issue_A_fast(1);
          } else {
            if (q == 7) load_A(k2);
          }
          if (q == 11) load_B_any(k2);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = nxt;
  };
  if (BLK) {
    for (int kt = 0; kt < nk; kt += ICG_PLANES_FLUSH_TILES) {
      tile_body(kt, std::true_type{});
#pragma unroll
      for (int h = 1; h < ICG_PLANES_FLUSH_TILES; ++h)
        if (kt + h < nk) tile_body(kt + h, std::false_type{});
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] += acc2[j];
  } else {
    for (int kt = 0; kt < nk; ++kt) tile_body(kt, std::false_type{});
  }

  // ------------------------------------------------------------------ epilogue
  // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int m = m0 + 32 * wv + row;
    if (m >= p.M) continue;
    const long out_row = (p.phase_mode == 1) ? phase_row(m) : (long)m;
    if (out_row < 0) continue;
    long res_row = out_row;
    if (p.res != nullptr && p.res_up == 1) {
      const int w = m % p.W;
      const int t = m / p.W;
      const int h = t % p.H;
      const int b = t / p.H;
      res_row = ((long)b * (p.H >> 1) + (h >> 1)) * (p.W >> 1) + (w >> 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + 32 * j + li;
      if (n < p.N) {
        float v = p.alpha * acc[j][r];
        if (p.bias != nullptr) v += p.bias[n];
        if (p.res != nullptr) {
          const float rv = p.res[res_row * p.ldc + n];
          v = (p.res_up == 2) ? (rv > 0.f ? v : 0.f) : v + rv;     // 2: ICG_RES_RELU_MASK
        }
        Cg[out_row * p.ldc + n] = v;
      }
    }
  }
}


// ---- persistent plane GEMM ----------------------------------------------------------------------------------------------
// The forward / data-gradient GEMMs over Winograd planes are C[z] = A[z] B[z]^T with A [M][K], B [N][K] (K = Cin contiguous)
// and a SHORT K (96 ... 1536: 6 ... 96 K-tiles per 128 x 32*TN output tile), so a workgroup that computes one output tile spends
// a large part of its life in the pipeline fill (two K-tiles of HBM latency before the first MFMA) and in the epilogue, with
// only 2-3 workgroups per CU to cover for it.  Here a workgroup owns several output tiles of the (plane, m-tile, n-tile) order
// (grid-strided inside its XCD's range, ~pt_run of them) and treats their K-tiles as ONE stream: the global loads run two K-tiles ahead of the MFMAs
// straight across output-tile boundaries (the loads for the next tile's first K-tiles are in flight while this tile's last
// MFMAs and its epilogue run), the LDS ring and the operand-fragment prefetch never drain.  Same thread -> data mapping, LDS
// layout, MFMA order, two-level accumulation and staging schedule as icg_gemm_body's fast path, so results are bit-identical to
// it; the loader state is one row offset per staged row (rows >= M / >= N are clamped: they only feed masked outputs).
// RAGGED: K % 16 != 0 (K % 4 == 0): the quads of the last K-tile past K are staged as zeros (costs the staging a select: own instantiation)
template <int TN, int BLK, bool RAGGED = false>
__device__ __forceinline__ void icg_planes_body(const GemmP& p) {
  constexpr int BM = 128, BN = 32 * TN, BK = 16, LDA = BM + 1, LDB = BN + 1, NBUF = 3;
  __shared__ __attribute__((aligned(16))) float As[NBUF][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[NBUF][BK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int kq = tid & 3, arow = tid >> 2;
  const int nk = (p.K + BK - 1) / BK;          // K % 4 == 0; the quads of the last K-tile past K are staged as zeros

  // this workgroup's output tiles: v = first, first + vstep, ... < last inside its XCD's contiguous range of the tile order (see
  // icg_gemm_body).  The stride is the number of workgroups of the XCD, so the workgroups resident at any moment work on
  // ADJACENT tiles, exactly like a one-tile-per-workgroup launch: they share the m-tile's A panel and the plane's B panel in L2.
  const unsigned tot = (unsigned)p.pt_tiles * (unsigned)p.pt_z, lin = blockIdx.x;
  unsigned first, last, vstep;
  if (p.swz) {
    const unsigned q = tot >> 3, r = tot & 7u, xcd = lin & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    vstep = gridDim.x >> 3;
    first = base + (lin >> 3);
    last = base + q + (xcd < r ? 1u : 0u);
  } else {
    vstep = gridDim.x;
    first = lin;
    last = tot;
  }
  if (first >= last) return;

  // ---- load stream: position (tile ld_v, K offset ld_k); two K-tiles ahead of the MFMAs
  unsigned ld_v = first;
  int ld_k = 0;
  const float* Agl;
  const float* Bgl;
  unsigned oa[2], ob[2];
  auto stream_tile = [&](unsigned v) {     // operand bases and this thread's row offsets of output tile v
    const unsigned zz = v / (unsigned)p.pt_tiles, tile = v - zz * (unsigned)p.pt_tiles;
    const int nt = (int)(tile % (unsigned)p.ntiles_n), mt = (int)(tile / (unsigned)p.ntiles_n);
    Agl = p.A + (long)zz * p.strideA;
    Bgl = p.B + (long)zz * p.strideB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = min(mt * BM + arow + 64 * i, p.M - 1);
      const int n = min(nt * BN + min(arow + 64 * i, BN - 1), p.N - 1);
      oa[i] = (unsigned)m * (unsigned)p.K;
      ob[i] = (unsigned)n * (unsigned)p.ldb;
    }
  };
  auto stream_advance = [&]() {            // next K-tile of the stream; past the end of the run it stays on the last one
    ld_k += BK;
    if (ld_k >= p.K) {
      if (ld_v + vstep < last) { ld_v += vstep; ld_k = 0; stream_tile(ld_v); }
      else ld_k = (nk - 1) * BK;
    }
  };
  float4 ra[2], rb[2];
  bool kin = true;            // this thread's quad of the K-tile held in ra / rb lies inside K (applied when the quad is staged:
                              // a select right after the load would make the wave wait for the data at issue time)
  auto issue_A = [&](int i) {
    const int kk = ld_k + 4 * kq;
    ra[i] = ld4(Agl + (oa[i] + (unsigned)(RAGGED ? min(kk, p.K - 4) : kk)));
  };
  auto issue_B = [&]() {
    const int kk = ld_k + 4 * kq;
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = ld4(Bgl + (ob[i] + (unsigned)(RAGGED ? min(kk, p.K - 4) : kk)));
    if (RAGGED) kin = kk < p.K;           // (issue_B is the last load of a K-tile: the flag describes the tile now in the registers)
  };
  auto write_A_row = [&](int buf, int i) {
    float* as = As[buf];
    const int row = arow + 64 * i;
    const float4 v = (!RAGGED || kin) ? ra[i] : zero4();
    as[(4 * kq + 0) * LDA + row] = v.x;
    as[(4 * kq + 1) * LDA + row] = v.y;
    as[(4 * kq + 2) * LDA + row] = v.z;
    as[(4 * kq + 3) * LDA + row] = v.w;
  };
  auto write_B_row = [&](int buf, int i) {
    float* bs = Bs[buf];
    const int nl = arow + 64 * i;
    if (BN == 128 || nl < BN) {
      bs[(4 * kq + 0) * LDB + nl] = rb[i].x;
      bs[(4 * kq + 1) * LDB + nl] = rb[i].y;
      bs[(4 * kq + 2) * LDB + nl] = rb[i].z;
      bs[(4 * kq + 3) * LDB + nl] = rb[i].w;
    }
  };

  f32x16 acc[TN];
  f32x16 acc2[BLK ? TN : 1];
  float fa[2], fb[2][TN];

  // pipeline fill: K-tile 0 of the first output tile into ring slot 0, K-tile 1 into the registers, fragments of k-step 0
  stream_tile(ld_v);
  issue_A(0); issue_A(1); issue_B();
  write_A_row(0, 0); write_A_row(0, 1); write_B_row(0, 0); write_B_row(0, 1);
  stream_advance();
  issue_A(0); issue_A(1); issue_B();
  __syncthreads();
  fa[0] = As[0][32 * wv + li + lh * LDA];
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[0][j] = Bs[0][li + lh * LDB + 32 * j];
  int cur = 0;

  // one K-tile (see icg_gemm_body's tile_body: identical MFMA order and staging schedule)
  // MODE 0: plain K-tile; 1: its first k-step folds the finished chains into the second level and starts fresh ones (BLK);
  // 2: first K-tile of an output tile: fresh chains, (BLK) second level cleared
  auto tile_body = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
    const int nxt = (cur == NBUF - 1) ? 0 : cur + 1;
    const float* as = As[cur] + 32 * wv + li + lh * LDA;
    const float* bs = Bs[cur] + li + lh * LDB;
    const float* asn = As[nxt] + 32 * wv + li + lh * LDA;
    const float* bsn = Bs[nxt] + li + lh * LDB;
#pragma unroll
    for (int t = 0; t < BK / 2; ++t) {
      const int pn = (t + 1) & 1;
      const bool wrap = (t + 1 == BK / 2);
      const float* fas = wrap ? asn : as + (2 * t + 2) * LDA;
      const float* fbs = wrap ? bsn : bs + (2 * t + 2) * LDB;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (MODE == 2 && t == 0) {
          if (BLK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
            asm volatile("" : "+v"(acc2[j]));
          }
          f32x16 zero;
#pragma unroll
          for (int r = 0; r < 16; ++r) zero[r] = 0.f;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1], fb[t & 1][j], zero, 0, 0, 0);
        } else if (MODE == 1 && t == 0) {
          asm volatile("" : "+a"(acc[j]));
          if (TN == 3) {
            // (96-column kernel only: at 128 columns the kernel runs two waves per SIMD either way and the packed adds are faster)
            // accumulator -> VGPR -> add as ONE statement per value: left to the compiler, all 16 reads of the column tile are
            // hoisted ahead of the first add (16 live copies: the 96-column kernel then needs 180 registers, without them it
            // fits three waves per SIMD).  The MFMA that last wrote acc[j] is TN - 1 = 2 MFMAs (>= 128 cycles) back, far beyond
            // the XDL-write -> accvgpr_read wait states the compiler would insert for its own copies.
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float b1 = acc2[j][r], t1;
              asm volatile("v_accvgpr_read_b32 %1, %2\n\tv_add_f32 %0, %0, %1" : "+v"(b1), "=&v"(t1) : "a"(acc[j][r]));
              acc2[j][r] = b1;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              f32x2 a2 = {acc[j][2 * r], acc[j][2 * r + 1]};
              f32x2 b2 = {acc2[j][2 * r], acc2[j][2 * r + 1]};
              asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b2) : "v"(a2));
              acc2[j][2 * r] = b2.x;
              acc2[j][2 * r + 1] = b2.y;
            }
          }
          f32x16 zero;
#pragma unroll
          for (int r = 0; r < 16; ++r) zero[r] = 0.f;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1], fb[t & 1][j], zero, 0, 0, 0);
        } else {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1], fb[t & 1][j], acc[j], 0, 0, 0);
        }
        if (j == 0) fa[wrap ? 0 : pn] = fas[0];
        if ((j & 1) == 0) {
          fb[wrap ? 0 : pn][j] = fbs[32 * j];
          if (j + 1 < TN) fb[wrap ? 0 : pn][j + 1] = fbs[32 * (j + 1)];
        }
        constexpr int PPG = (8 * TN >= 12) ? 1 : 2;
        const int m = t * TN + j;
#pragma unroll
        for (int q = m * PPG; q < (m + 1) * PPG; ++q) {
          if (q == 1) write_A_row(nxt, 0);
          if (q == 3) write_A_row(nxt, 1);
          if (q == 4) write_B_row(nxt, 0);
          if (q == 5) write_B_row(nxt, 1);
          if (q == 6) {
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
          }
          if (q == 7) stream_advance();
          if (q == 8) issue_A(0);
          if (q == 10) issue_A(1);
          if (q == 11) issue_B();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    cur = nxt;
  };

  for (unsigned v = first; v < last; v += vstep) {
    typedef std::integral_constant<int, 0> Plain;
    typedef std::integral_constant<int, 1> Flush;
    typedef std::integral_constant<int, 2> Start;
    tile_body(Start{});
    if (BLK) {
#pragma unroll
      for (int h = 1; h < ICG_PLANES_FLUSH_TILES; ++h)
        if (h < nk) tile_body(Plain{});
      for (int kt = ICG_PLANES_FLUSH_TILES; kt < nk; kt += ICG_PLANES_FLUSH_TILES) {
        tile_body(Flush{});
#pragma unroll
        for (int h = 1; h < ICG_PLANES_FLUSH_TILES; ++h)
          if (kt + h < nk) tile_body(Plain{});
      }
    } else {
      for (int kt = 1; kt < nk; ++kt) tile_body(Plain{});
    }
    // epilogue of output tile v (the loads of the next tile's first two K-tiles are in flight meanwhile, so the staging and
    // fragment registers stay live: one column tile at a time, 16 accumulator copies in VGPRs, pinned against hoisting)
    const unsigned zz = v / (unsigned)p.pt_tiles, tile = v - zz * (unsigned)p.pt_tiles;
    const int n0 = (int)(tile % (unsigned)p.ntiles_n) * BN, m0 = (int)(tile / (unsigned)p.ntiles_n) * BM;
    float* __restrict__ Cg = p.C + (long)zz * p.strideC + (long)(m0 + 32 * wv + 4 * lh) * p.ldc + (n0 + li);
    const int mrem = p.M - (m0 + 32 * wv + 4 * lh), nrem = p.N - (n0 + li);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      asm volatile("" : "+a"(acc[j]));
#pragma unroll
      for (int r = 0; r < 8; ++r) {         // two accumulator copies in flight, (BLK) folded into the second level in place
        f32x2 c2 = {acc[j][2 * r], acc[j][2 * r + 1]};
        if (BLK) {
          f32x2 b2 = {acc2[j][2 * r], acc2[j][2 * r + 1]};
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(c2) : "v"(b2));
        } else {
          asm volatile("" : "+v"(c2));
        }
        if (32 * j < nrem) {
          const int row0 = ((2 * r) & 3) + 8 * ((2 * r) >> 2);
          float v0 = p.alpha * c2.x, v1 = p.alpha * c2.y;
          if (p.bias != nullptr) {                           // 1x1 convolutions (plain = 2): bias, residual / ReLU mask as in
            const float bv = p.bias[n0 + li + 32 * j];       // icg_gemm_body's epilogue; never set for the plane GEMMs
            v0 += bv; v1 += bv;
          }
          if (p.res != nullptr) {
            const float* rp = p.res + (Cg - p.C) + (long)row0 * p.ldc + 32 * j;
            if (row0 < mrem) { const float rv = rp[0]; v0 = (p.res_up == 2) ? (rv > 0.f ? v0 : 0.f) : v0 + rv; }
            if (row0 + 1 < mrem) { const float rv = rp[p.ldc]; v1 = (p.res_up == 2) ? (rv > 0.f ? v1 : 0.f) : v1 + rv; }
          }
          if (row0 < mrem) Cg[(long)row0 * p.ldc + 32 * j] = v0;
          if (row0 + 1 < mrem) Cg[(long)(row0 + 1) * p.ldc + 32 * j] = v1;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int AMODE, int BMODE, int TN, int PATH>
__global__ __launch_bounds__(256) void icg_gemm_kernel(GemmP p) {
  icg_gemm_body<AMODE, BMODE, TN, PATH>(p);
}

// the same code under a second name for the batched GEMMs over Winograd planes (winograd.hip), so that a profile separates
// them from the convolutions that run on the kernel directly (only the fast loader is instantiated)
#ifndef ICG_PLANES_TN4_MIN_WAVES
#define ICG_PLANES_TN4_MIN_WAVES 1     // (3: ablation build LB3 of tools/build_dbg.sh -- 168 registers with ~25 spilled dwords)
#endif
template <int AMODE, int BMODE, int TN>
__global__ __launch_bounds__(256, (TN == 4 ? ICG_PLANES_TN4_MIN_WAVES : (TN == 3 ? 3 : 1))) void icg_gemm_planes_kernel(GemmP p) {
  if constexpr (AMODE == A_K && BMODE == B_K && ICG_PLANES_PERSISTENT) icg_planes_body<TN, ICG_PLANES_BLOCKED>(p);   // 1-D grid
  else icg_gemm_body<AMODE, BMODE, TN, 2, ICG_PLANES_BLOCKED, 1>(p);
}
// ... with single-level accumulation, for plane GEMMs whose chains are short anyway (K <= planes_1level_max_k(): the
// rounding of a K-long chain grows with K, and at K <= 192 the second level buys < 2x -- 1.7e-6 / 2.3e-6 against 1.2e-6 per
// layer at 96 / 192 channels, profiles/r02_wino_microbench.txt -- while costing the 128-column kernel a wave per SIMD)
template <int AMODE, int BMODE, int TN>
__global__ __launch_bounds__(256) void icg_gemm_planes1_kernel(GemmP p) {
  if constexpr (AMODE == A_K && BMODE == B_K && ICG_PLANES_PERSISTENT) icg_planes_body<TN, 0>(p);
  else icg_gemm_body<AMODE, BMODE, TN, 2, 0, 1>(p);
}

// ... and for K % 16 != 0 (Winograd layers with Cin = 24, 40, ...; the D attention's d = 24): LEVELS as above
template <int TN, int LEVELS>
__global__ __launch_bounds__(256) void icg_gemm_planes_ragged_kernel(GemmP p) {
  icg_planes_body<TN, (LEVELS == 2 ? ICG_PLANES_BLOCKED : 0), true>(p);
}

#ifndef ICG_PLANES_1LEVEL_MAX_K
#define ICG_PLANES_1LEVEL_MAX_K 0      // K up to which plane GEMMs use the single-level kernel; 0 = never (ablation build L1_384:
#endif                                 // 4-5 ms faster, but the ill-conditioned small test networks then exceed the strict tolerances)
static constexpr int planes_1level_max_k() { return ICG_PLANES_1LEVEL_MAX_K; }
#ifndef ICG_PLANES_RUN_KTILES
#define ICG_PLANES_RUN_KTILES 64       // persistent plane GEMM: K-tiles of MFMA work per workgroup (measured flat between 16 and 256)
#endif

static thread_local int g_gemm_planes = 0;
void icg_gemm_mark_planes(int on) { g_gemm_planes = on; }

// pgemm.hip; returns 1 when the shape is not one it takes
int icg_pgemm_nn_launch(const float* A, const float* B, float* C, int M, int N, int K, long ldc, long sA, long sB, long sC,
                        int planes, float alpha, int levels, hipStream_t st, int* tn_out);
int icg_pgemm_tn_launch(const float* A, const float* B, float* C, int M, int N, int K, long sA, long sB, int planes, int kchunk,
                        int slices, int levels, hipStream_t st, int* tn_out);
int icg_pconv_launch(const float* A, const float* B, float* C, int M, int N, int K, int H, int W, int Cin, int R, int up, int Hs,
                     int Ws, int pad_h, int pad_w, int gs, int Hb, int Wb, long ldb, long ldc, const float* bias, const float* res,
                     int res_mode, float alpha, long strideB, int phase_mode, int oH, int oW, int pre_relu, int zdim,
                     hipStream_t st, int* tn_out);

// deterministic second stage of split-K: out[i] = sum_z slab[z][i].  Round 3: 16-byte loads and four independent accumulation
// chains (slabs z, z+1, z+2, z+3 in flight together; fixed combination order, so the result is still a pure function of the
// slabs) -- the scalar form issued one dependent 4-byte load per slab and ran at ~0.3 TB/s (136 us per launch in the cfg3 step)
__global__ __launch_bounds__(256) void icg_splitk_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n,
                                                                int splits) {
  const long stride = (long)gridDim.x * blockDim.x;
  const bool vec = ((n & 3) == 0) && ((((uintptr_t)slabs) | ((uintptr_t)out)) & 15) == 0;
  if (vec) {
    const long n4 = n >> 2;
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(slabs);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(out);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      int z = 0;
      for (; z + 4 <= splits; z += 4) {
        const float4 v0 = s4[(long)z * n4 + i], v1 = s4[(long)(z + 1) * n4 + i], v2 = s4[(long)(z + 2) * n4 + i],
                     v3 = s4[(long)(z + 3) * n4 + i];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; z < splits; ++z) {
        const float4 v0 = s4[(long)z * n4 + i];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      }
      o4[i] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                          (a0.w + a1.w) + (a2.w + a3.w));
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long)z * n + i];
    out[i] = s;
  }
}

// the same sum with the four accumulation chains on four thread groups of the block (64 columns x 4 chains, LDS fold in the same
// (a0 + a1) + (a2 + a3) order, the splits % 4 leftovers still at the end of chain 0: the same additions in the same order).  For the
// weight gradients' shape -- many slabs (8 ... 66), few columns: the one-thread form walks splits / 4 dependent rounds with 324
// blocks on the chip (54 us for 87 MB); here four times the threads share the walk.
__global__ __launch_bounds__(256) void icg_splitk_reduce_sliced_kernel(const float4* __restrict__ s4, float4* __restrict__ o4, long n4,
                                                                       int splits) {
  __shared__ float4 red[3][64];
  const int col = threadIdx.x & 63, ch = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const int full = splits & ~3;                       // chain ch takes z = ch, ch + 4, ... < full, in order
    int z = ch;
    for (; z + 12 < full; z += 16) {
      const float4 v0 = s4[(long)z * n4 + i], v1 = s4[(long)(z + 4) * n4 + i], v2 = s4[(long)(z + 8) * n4 + i],
                   v3 = s4[(long)(z + 12) * n4 + i];
      a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
      a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
      a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
      a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
    }
    for (; z < full; z += 4) {
      const float4 v0 = s4[(long)z * n4 + i];
      a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
    }
    if (ch == 0)
      for (z = full; z < splits; ++z) {
        const float4 v0 = s4[(long)z * n4 + i];
        a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
      }
  }
  if (ch) red[ch - 1][col] = a;
  __syncthreads();
  if (ch == 0 && i < n4) {
    const float4 a1 = red[0][col], a2 = red[1][col], a3 = red[2][col];
    o4[i] = make_float4((a.x + a1.x) + (a2.x + a3.x), (a.y + a1.y) + (a2.y + a3.y), (a.z + a1.z) + (a2.z + a3.z),
                        (a.w + a1.w) + (a2.w + a3.w));
  }
}

static void launch_splitk_reduce(const float* slabs, float* out, long n, int splits, hipStream_t st) {
  const bool vec = ((n & 3) == 0) && ((((uintptr_t)slabs) | ((uintptr_t)out)) & 15) == 0;
  if (vec && splits >= 8 && (n >> 2) <= 64L * 65535) {
    hipLaunchKernelGGL(icg_splitk_reduce_sliced_kernel, dim3((unsigned)icg_cdiv(n >> 2, 64)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(slabs), reinterpret_cast<float4*>(out), n >> 2, splits);
    return;
  }
  const int blocks = (int)(icg_cdiv(n, 256) > 2048 ? 2048 : icg_cdiv(n, 256));
  hipLaunchKernelGGL(icg_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, slabs, out, n, splits);
}

// ---------------------------------------------------------------------- host side
static int pick_tn(int N) {
  if (N <= 32) return 1;
  if (N <= 64) return 2;
  if (N % 128 == 0) return 4;
  if (N % 96 == 0) return 3;
  if (N <= 96) return 3;
  return 4;
}

static int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

static bool getenv_flag(const char* name) {   // measurement switch, read once per process
  static const bool v = [](const char* n) { const char* e = getenv(n); return e && e[0] == '1'; }(name);
  return v;
}

// template arguments of the last icg_gemm_kernel this host thread launched (measurement support: lets a profiler
// harness name its HIP-event timings exactly like rocprofv3 names the kernel)
static thread_local int g_last_variant[4] = {-1, -1, -1, -1};

// (label of a launch made outside this file: fwino.hip reports {5, in_up, out_pool, planes per dimension})
void icg_gemm_set_last_variant(int a, int b, int c, int d) {
  g_last_variant[0] = a; g_last_variant[1] = b; g_last_variant[2] = c; g_last_variant[3] = d;
}
extern "C" int icg_gemm_last_variant(int* out4) {
  ICG_REQUIRE(out4);
  for (int i = 0; i < 4; ++i) out4[i] = g_last_variant[i];
  return ICG_OK;
}

template <int AMODE, int BMODE>
static int launch_gemm(const GemmP& p0, bool vec, int zdim, hipStream_t st, bool fast_ok = false,
                       bool vec_a_only = false) {
  GemmP p = p0;
  p.lw = ilog2_exact(p.W);
  p.lh = ilog2_exact(p.H);
  p.vec_b = 1;
  if (!vec && vec_a_only) {     // A operand vectorisable, B not (e.g. weight gradient of a 3-channel output)
    vec = true;
    fast_ok = false;
    p.vec_b = 0;
  }
  int path = vec ? 1 : 0;
  const bool plain_body = AMODE == A_K && BMODE == B_K && ICG_PLANES_PERSISTENT && (g_gemm_planes || p.plain) && p.kchunk == 0 &&
                          p.bsplit == 0 && p.phase_mode == 0 && !p.pre_affine && !p.pre_relu && p.up == 0 && p.res_up != 1 &&
                          ((p.bias == nullptr && p.res == nullptr) || p.plain == 2);
  if (vec && fast_ok && p.zmask == 0 && plain_body && p.K % 4 == 0 && p.K >= 4) path = 2;     // icg_planes_body masks the K tail
  if (vec && fast_ok && p.zmask == 0) {
    if (AMODE == A_K && (p.Cin % 16 == 0) && (p.kchunk == 0 || (BMODE == B_K && p.kchunk % 16 == 0))) path = 2;
    if (AMODE == A_M && p.lw >= 0 && p.lh >= 0) path = 2;
  }
  const int tn = pick_tn(p.N);
  const int bn = 32 * tn;
  p.ntiles_n = (int)icg_cdiv(p.N, bn);
  const long tiles = icg_cdiv(p.M, 128) * p.ntiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffL || zdim <= 0 || zdim > 65535) return ICG_ERR_ARG;
  p.swz = (tiles * zdim >= 16 && tiles * zdim < 0x7fffffffL && !getenv_flag("ICG_NO_XCD_SWIZZLE")) ? 1 : 0;
  dim3 grid((unsigned)tiles, 1, (unsigned)zdim), block(256);
#define ICG_LAUNCH(TN_, PATH_) \
  hipLaunchKernelGGL((icg_gemm_kernel<AMODE, BMODE, TN_, PATH_>), grid, block, 0, st, p)
#define ICG_LAUNCH_TN(PATH_)             \
  switch (tn) {                          \
    case 1: ICG_LAUNCH(1, PATH_); break; \
    case 2: ICG_LAUNCH(2, PATH_); break; \
    case 3: ICG_LAUNCH(3, PATH_); break; \
    default: ICG_LAUNCH(4, PATH_); break; \
  }
  if (path == 2 && p.pre_affine) path = 3;
  g_last_variant[0] = AMODE; g_last_variant[1] = BMODE; g_last_variant[2] = tn; g_last_variant[3] = path;
  const bool plain = p.plain != 0 && plain_body && !g_gemm_planes;
  if constexpr (AMODE == A_K && BMODE == B_K) {
    // second-generation implicit-GEMM convolution (pgemm.hip, icg_pconv_kernel) for the forward / data-gradient shaped launches
    // of the fast path that have no affine prologue, no split-K and enough output tiles to fill the chip: D's 3x3 / 4x4-stride-2 /
    // 2x2-phase layers, the data gradients without a BN in front, the prologue-free 1x1 convolutions
    if (path == 2 && !g_gemm_planes && p.plain != 1 && p.kchunk == 0 && p.bsplit == 0 && p.zmask == 0 && !p.pre_affine &&
        p.Cin % 16 == 0 && p.K == p.R * p.R * p.Cin && (p.phase_mode == 0 || (p.phase_mode == 1 && zdim == 4)) &&
        (p.phase_mode != 0 || zdim == 1) && p.res_up != 1 && (p.strideA == 0 || zdim == 1) && (p.strideC == 0 || zdim == 1)) {
      int nt2 = 0;
      const int rc = icg_pconv_launch(p.A, p.B, p.C, p.M, p.N, p.K, p.H, p.W, p.Cin, p.R, p.up, p.Hs, p.Ws, p.pad_h, p.pad_w, p.gs,
                                      p.Hb, p.Wb, p.ldb, p.ldc, p.bias, p.res, p.res != nullptr ? p.res_up : 0, p.alpha, p.strideB,
                                      p.phase_mode, p.oH, p.oW, p.pre_relu, zdim, st, &nt2);
      if (rc != 1) {
        g_last_variant[0] = 4; g_last_variant[1] = p.pre_relu; g_last_variant[2] = nt2; g_last_variant[3] = (nt2 == 3) ? 2 : 1;
        return rc;
      }
    }
    // second-generation plane GEMM (pgemm.hip: LDS-DMA staging, 16-byte operand fragments, 8 waves per workgroup) for the
    // Winograd-plane GEMMs it has a tile for (N a multiple of 128 or 96, K a multiple of 32); everything else stays below
    if (path == 2 && g_gemm_planes && plain_body && p.bias == nullptr && p.res == nullptr && p.ldb == p.K) {
      const int levels = (p.K <= planes_1level_max_k()) ? 1 : 2;
      int nt2 = 0;
      const int rc = icg_pgemm_nn_launch(p.A, p.B, p.C, p.M, p.N, p.K, p.ldc, p.strideA, p.strideB, p.strideC, zdim, p.alpha,
                                         levels, st, &nt2);
      if (rc != 1) {
        g_last_variant[0] = 2; g_last_variant[1] = 0; g_last_variant[2] = nt2 % 100; g_last_variant[3] = (nt2 >= 100) ? 4 : 2;
        return rc;
      }
    }
  }
  if (path == 2 && (g_gemm_planes || plain) && plain_body) {
    // persistent plane GEMM (icg_planes_body): 1-D grid, every workgroup owns a run of consecutive output tiles sized for
    // ~64 K-tiles of MFMA work, as long as the launch still queues several workgroups per CU
    if (!plain_body) return ICG_ERR_ARG;
    const long tot = tiles * zdim;
    const int nk = (p.K + 15) / 16;
    int run = (ICG_PLANES_RUN_KTILES + nk - 1) / nk;
    if (run > 32) run = 32;
    while (run > 1 && tot / run < 2048) --run;
    p.pt_tiles = (int)tiles; p.pt_z = zdim; p.pt_run = run;
    const long per_xcd = icg_cdiv(icg_cdiv(tot, 8), run);
    grid = p.swz ? dim3((unsigned)(8 * per_xcd), 1, 1) : dim3((unsigned)icg_cdiv(tot, run), 1, 1);
  }
  const bool one_level = plain || (g_gemm_planes && p.K <= planes_1level_max_k());
  if (path == 2 && plain_body && (p.K % 16) != 0) {
    if (one_level) g_last_variant[3] = 4;
#define ICG_RAG(TN_) \
  if (one_level) hipLaunchKernelGGL((icg_gemm_planes_ragged_kernel<TN_, 1>), grid, block, 0, st, p); \
  else hipLaunchKernelGGL((icg_gemm_planes_ragged_kernel<TN_, 2>), grid, block, 0, st, p)
    switch (tn) {
      case 1: ICG_RAG(1); break;
      case 2: ICG_RAG(2); break;
      case 3: ICG_RAG(3); break;
      default: ICG_RAG(4); break;
    }
#undef ICG_RAG
  } else if (path == 2 && one_level && (plain || g_gemm_planes)) {
    g_last_variant[3] = 4;       // plane GEMM, single-level chains
    switch (tn) {
      case 1: hipLaunchKernelGGL((icg_gemm_planes1_kernel<AMODE, BMODE, 1>), grid, block, 0, st, p); break;
      case 2: hipLaunchKernelGGL((icg_gemm_planes1_kernel<AMODE, BMODE, 2>), grid, block, 0, st, p); break;
      case 3: hipLaunchKernelGGL((icg_gemm_planes1_kernel<AMODE, BMODE, 3>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((icg_gemm_planes1_kernel<AMODE, BMODE, 4>), grid, block, 0, st, p); break;
    }
  } else if (path == 2 && g_gemm_planes) {
    switch (tn) {
      case 1: hipLaunchKernelGGL((icg_gemm_planes_kernel<AMODE, BMODE, 1>), grid, block, 0, st, p); break;
      case 2: hipLaunchKernelGGL((icg_gemm_planes_kernel<AMODE, BMODE, 2>), grid, block, 0, st, p); break;
      case 3: hipLaunchKernelGGL((icg_gemm_planes_kernel<AMODE, BMODE, 3>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((icg_gemm_planes_kernel<AMODE, BMODE, 4>), grid, block, 0, st, p); break;
    }
  } else if (path == 3) { ICG_LAUNCH_TN(3) } else if (path == 2) { ICG_LAUNCH_TN(2) } else if (path == 1) { ICG_LAUNCH_TN(1) }
  else { ICG_LAUNCH_TN(0) }
#undef ICG_LAUNCH_TN
#undef ICG_LAUNCH
  return icg_check_launch();
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// direct kernels for 3x3 convolutions with <= 4 output channels (narrow_conv.hip)
bool icg_narrow_conv_ok(int Cin, int Cout, int R, const void* a, const void* b, const void* c, long ssb);
int icg_narrow_fprop(const float* x, const float* w, const float* bias, const float* scale, const float* shift, long ssb,
                     float* out, int B, int H, int W, int Cin, int Cout, int affine, int relu, float alpha, hipStream_t st);
size_t icg_narrow_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_narrow_wgrad(const float* x, const float* dy, const float* scale, const float* shift, long ssb, float* dw,
                     void* workspace, int B, int H, int W, int Cin, int Cout, int affine, int relu, hipStream_t st);

// 3x3 convolutions with <= 4 input channels (narrow_conv.hip)
bool icg_thin_conv_ok(int Cin, int Cout, int R);
size_t icg_thin_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_thin_fprop(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin, int Cout, int R,
                   float alpha, hipStream_t st);
int icg_thin_wgrad(const float* x, const float* dy, float* dw, void* workspace, int B, int H, int W, int Cin, int Cout, int R,
                   hipStream_t st);

// skinny linear layers (batch rows x odd K; narrow_conv.hip)
bool icg_skinny_ok(long M, int Cin, int R);
int icg_skinny_fprop(const float* x, const float* w, const float* bias, float* out, int M, int N, int K, float alpha, hipStream_t st);
int icg_skinny_wgrad(const float* x, const float* dy, float* dw, int M, int N, int K, hipStream_t st);

// ---- split-K for forward / data-gradient launches that cannot fill the chip ------------------------------------------
// A [M x N] output with fewer than ~384 tiles leaves CUs idle while every tile walks the whole K (StyleGAN2 at batch 16 below
// 32x32, small-batch sampling): the K range is cut into S slices (blockIdx.z), each writes a raw partial slab, and a second
// kernel sums the slabs in fixed order and applies the epilogue (alpha, bias, residual).  Deterministic.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ slabs, int splits, long MN, int N,
                                                              float alpha, const float* __restrict__ bias,
                                                              const float* __restrict__ res, int res_up, int H, int W,
                                                              float* __restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long)z * MN + i];
    const int n = (int)(i % N);
    float v = alpha * s;
    if (bias) v += bias[n];
    if (res) {
      long rr = i / N;
      if (res_up == 1) {
        const int w = (int)(rr % W);
        const long t = rr / W;
        const int h = (int)(t % H);
        const long b = t / H;
        rr = (b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1);
      }
      const float rv = res[rr * N + n];
      v = (res_up == 2) ? (rv > 0.f ? v : 0.f) : v + rv;
    }
    out[i] = v;
  }
}

static int fprop_splits(long M, int N, int K) {
  const long tiles = icg_cdiv(M, 128) * icg_cdiv(N, 32 * pick_tn(N));
  if (tiles >= 384) return 1;
  const int nk = K / 16;
  long s = 768 / tiles;
  if (s > nk / 16) s = nk / 16;
  if (s > 16) s = 16;
  return s < 2 ? 1 : (int)s;
}

static size_t fprop_splitk_bytes(long M, int N, int K, int Cin) {
  if (Cin % 16 != 0) return 0;
  const int s = fprop_splits(M, N, K);
  return s <= 1 ? 0 : (size_t)s * (size_t)M * (size_t)N * sizeof(float);
}

// p: a complete plain (phase_mode 0, z = 1, fast-path eligible) A_K/B_K problem.  Returns -1 if split-K does not apply.
static int launch_fprop_splitk(const GemmP& p0, bool vec, hipStream_t st, bool small, void* workspace,
                               size_t workspace_bytes) {
  if (!vec || !small || p0.Cin % 16 != 0 || p0.zmask != 0 || p0.phase_mode != 0 || workspace == nullptr) return -1;
  const int S = fprop_splits(p0.M, p0.N, p0.K);
  if (S <= 1 || workspace_bytes < (size_t)S * p0.M * p0.N * sizeof(float)) return -1;
  GemmP p = p0;
  const int nk = p.K / 16;
  p.kchunk = (int)icg_cdiv(nk, S) * 16;
  const int splits = (int)icg_cdiv(p.K, p.kchunk);
  p.C = (float*)workspace;
  p.ldc = p.N;
  p.strideA = 0; p.strideB = 0; p.strideC = (long)p.M * p.N;
  p.alpha = 1.f; p.bias = nullptr; p.res = nullptr; p.res_up = 0;
  int rc = launch_gemm<A_K, B_K>(p, vec, splits, st, small);
  if (rc != ICG_OK) return rc;
  const long MN = (long)p.M * p.N;
  long blocks = icg_cdiv(MN, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)workspace, splits, MN,
                     p.N, p0.alpha, p0.bias, p0.res, p0.res_up, p0.H, p0.W, p0.C);
  return icg_check_launch();
}

// C[m][n] = alpha sum_k A[m][k] B[n][k] (+ bias[n]) for a handful of rows: StyleGAN2's mapping / affine / epilogue dense layers at batch
// 16 (M = 16, N = 512, K = 512 .. 8192) and BigGAN's conditional-BN projections (M = 64, K = 657, N = the block's channels, layers.py:
// 367-374).  On the tiled MFMA kernel such a GEMM is N / 128 workgroups walking the whole K chain (35 us at K = 512, 550 us at
// K = 8192); the first row-per-lane kernel for the K = 657 layers (skinny_fprop_kernel) ran N / 8 blocks of 165 dependent steps: 37 us.
// Here one WAVE owns 4 output columns and 16 rows, its lanes stride over K (16-byte loads when K % 4 == 0, dwords otherwise: a B row is
// read as contiguous pieces) and the 16 x 4 partial sums are combined by a butterfly over the 64 lanes (fixed order).
template <int MR, int VEC>
__global__ __launch_bounds__(256) void smallm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                        float alpha) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n0 = ((int)blockIdx.x * 4 + wv) * 4;
  const int mbase = (int)blockIdx.y * MR;
  if (n0 >= N) return;                                   // wave-uniform
  float acc[MR][4];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
  if constexpr (VEC == 4) {
    const int K4 = K >> 2;
    const float4* Bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bp[j] = reinterpret_cast<const float4*>(B + (long)min(n0 + j, N - 1) * K);
    for (int k4 = lane; k4 < K4; k4 += 64) {
      float4 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bp[j][k4];
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const float4 a = reinterpret_cast<const float4*>(A + (long)min(mbase + m, M - 1) * K)[k4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[m][j] = fmaf(a.w, b[j].w, fmaf(a.z, b[j].z, fmaf(a.y, b[j].y, fmaf(a.x, b[j].x, acc[m][j]))));
      }
    }
  } else {
    const float* Bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bp[j] = B + (long)min(n0 + j, N - 1) * K;
    for (int k = lane; k < K; k += 64) {
      float b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bp[j][k];
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const float a = A[(long)min(mbase + m, M - 1) * K + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(a, b[j], acc[m][j]);
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[m][j];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == m * 4 + j) mine = v;                   // lane l keeps element (l / 4, l % 4)
    }
  const int m = mbase + (lane >> 2), n = n0 + (lane & 3);
  if ((lane >> 2) < MR && m < M && n < N) C[(long)m * N + n] = alpha * mine + (bias ? bias[n] : 0.f);
}

static int smallm_nt_launch(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float alpha,
                            hipStream_t st) {
  const dim3 grid((unsigned)icg_cdiv(N, 16), (unsigned)icg_cdiv(M, 16));
  if (K % 4 == 0 && aligned16(A) && aligned16(B))
    hipLaunchKernelGGL((smallm_nt_kernel<16, 4>), grid, dim3(256), 0, st, A, B, bias, C, M, N, K, alpha);
  else
    hipLaunchKernelGGL((smallm_nt_kernel<16, 1>), grid, dim3(256), 0, st, A, B, bias, C, M, N, K, alpha);
  return icg_check_launch();
}

// ---- a GROUP of dense layers that read the same few rows: the four conditional-BN projections of a GBlock (bn1.gain, bn1.bias,
// bn2.gain, bn2.bias of layers.py:367-374 applied to the same y, 64 rows x K = 657) as ONE launch per direction instead of four
// launches of ~17 us of latency each.  blockIdx.z selects the item; the per-item arithmetic is smallm_nt_kernel's / skinny_wgrad_kernel's
// (same order: bit-identical to the per-layer launches); the data gradient sums the items inside one accumulation chain.
struct LinGroupPack {
  icg_linear_item l[ICG_LINEAR_GROUP_MAX];
  int n, M, K;
};

// mode 0: out_i[m][c] = sum_k x_i[m][k] w_i[c][k]        (wave: 16 rows x 4 columns, lanes stride over k)
template <int VEC>
__global__ __launch_bounds__(256) void lin_group_fprop_kernel(LinGroupPack p) {
  const icg_linear_item& L = p.l[blockIdx.z];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n0 = ((int)blockIdx.x * 4 + wv) * 4, mbase = (int)blockIdx.y * 16;
  const int M = p.M, K = p.K, N = L.N;
  if (n0 >= N) return;
  const float* __restrict__ A = L.x;
  const float* __restrict__ B = L.w;
  float acc[16][4];
#pragma unroll
  for (int m = 0; m < 16; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
  if constexpr (VEC == 4) {
    const int K4 = K >> 2;
    const float4* Bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bp[j] = reinterpret_cast<const float4*>(B + (long)min(n0 + j, N - 1) * K);
    for (int k4 = lane; k4 < K4; k4 += 64) {
      float4 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bp[j][k4];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float4 a = reinterpret_cast<const float4*>(A + (long)min(mbase + m, M - 1) * K)[k4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[m][j] = fmaf(a.w, b[j].w, fmaf(a.z, b[j].z, fmaf(a.y, b[j].y, fmaf(a.x, b[j].x, acc[m][j]))));
      }
    }
  } else {
    const float* Bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bp[j] = B + (long)min(n0 + j, N - 1) * K;
    for (int k = lane; k < K; k += 64) {
      float b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bp[j][k];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float a = A[(long)min(mbase + m, M - 1) * K + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(a, b[j], acc[m][j]);
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int m = 0; m < 16; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[m][j];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == m * 4 + j) mine = v;
    }
  const int m = mbase + (lane >> 2), n = n0 + (lane & 3);
  if (m < M && n < N) L.out[(long)m * N + n] = mine;
}

// mode 1: dw_i[k][c .. c + 3] = sum_m x_i[m][k] dy_i[m][c .. c + 3]      (HWIO at R = 1; one thread per (k, column quad))
__global__ __launch_bounds__(256) void lin_group_wgrad_kernel(LinGroupPack p) {
  const icg_linear_item& L = p.l[blockIdx.z];
  const int N4 = L.N >> 2, M = p.M, K = p.K;
  const long total = (long)K * N4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int n4 = (int)(i % N4);
    const int k = (int)(i / N4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* g = reinterpret_cast<const float4*>(L.dy) + n4;
    for (int b = 0; b < M; ++b) {
      const float xv = L.x[(long)b * K + k];
      const float4 d = g[(long)b * N4];
      acc.x = fmaf(xv, d.x, acc.x); acc.y = fmaf(xv, d.y, acc.y); acc.z = fmaf(xv, d.z, acc.z); acc.w = fmaf(xv, d.w, acc.w);
    }
    reinterpret_cast<float4*>(L.out)[i] = acc;
  }
}

// mode 2: dx[m][k] = sum_i sum_c dy_i[m][c] wd_i[k][c]     (wd_i = W_i / sigma in the [K][N_i] layout).  Every item is reduced on
// its own (lane chains + butterfly, the arithmetic of the per-layer data-gradient launch) and the items are added in order: the
// result is what the four launches + three elementwise adds gave, up to the order of those adds.
__global__ __launch_bounds__(256) void lin_group_dgrad_kernel(LinGroupPack p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k0 = ((int)blockIdx.x * 4 + wv) * 4, mbase = (int)blockIdx.y * 16;
  const int M = p.M, K = p.K;
  if (k0 >= K) return;
  float total = 0.f;
  for (int it = 0; it < p.n; ++it) {
    const icg_linear_item& L = p.l[it];
    const int N4 = L.N >> 2;
    float acc[16][4];
#pragma unroll
    for (int m = 0; m < 16; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    const float4* Bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bp[j] = reinterpret_cast<const float4*>(L.w + (long)min(k0 + j, K - 1) * L.N);
    for (int c4 = lane; c4 < N4; c4 += 64) {
      float4 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bp[j][c4];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float4 a = reinterpret_cast<const float4*>(L.dy + (long)min(mbase + m, M - 1) * L.N)[c4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[m][j] = fmaf(a.w, b[j].w, fmaf(a.z, b[j].z, fmaf(a.y, b[j].y, fmaf(a.x, b[j].x, acc[m][j]))));
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int m = 0; m < 16; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[m][j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == m * 4 + j) mine = v;
      }
    total = it == 0 ? mine : total + mine;
  }
  const int m = mbase + (lane >> 2), k = k0 + (lane & 3);
  if (m < M && k < K) p.l[0].out[(long)m * K + k] = total;
}

extern "C" int icg_linear_group(const icg_linear_item* items, int n, int M, int K, int mode, void* stream) {
  ICG_REQUIRE(items && n >= 1 && n <= ICG_LINEAR_GROUP_MAX && M >= 1 && K >= 1 && mode >= 0 && mode <= 2);
  LinGroupPack p{};
  p.n = n; p.M = M; p.K = K;
  int nmax = 0;
  bool vec = (K % 4 == 0);
  for (int i = 0; i < n; ++i) {
    const icg_linear_item& L = items[i];
    ICG_REQUIRE(L.N >= 1 && (mode == 2 ? (L.dy && L.w) : (L.x && L.out && (mode == 0 ? L.w != nullptr : L.dy != nullptr))));
    if (mode != 0) ICG_REQUIRE(L.N % 4 == 0 && aligned16(L.dy) && (mode == 1 ? aligned16(L.out) : aligned16(L.w)));
    vec = vec && aligned16(L.x) && aligned16(L.w);
    p.l[i] = L;
    nmax = max(nmax, L.N);
  }
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    const dim3 grid((unsigned)icg_cdiv(nmax, 16), (unsigned)icg_cdiv(M, 16), (unsigned)n);
    if (vec) hipLaunchKernelGGL((lin_group_fprop_kernel<4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lin_group_fprop_kernel<1>), grid, dim3(256), 0, st, p);
  } else if (mode == 1) {
    long nb = icg_cdiv((long)K * (nmax / 4), 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(lin_group_wgrad_kernel, dim3((unsigned)nb, 1, (unsigned)n), dim3(256), 0, st, p);
  } else {
    ICG_REQUIRE(items[0].out);
    hipLaunchKernelGGL(lin_group_dgrad_kernel, dim3((unsigned)icg_cdiv(K, 16), (unsigned)icg_cdiv(M, 16)), dim3(256), 0, st, p);
  }
  return icg_check_launch();
}

static int conv2d_fprop_impl(const float* x, const float* w, const float* bias, const float* residual, float* out,
                             const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin,
                             int Cout, int R, unsigned flags, float alpha, void* workspace, size_t workspace_bytes,
                             void* stream);

extern "C" size_t icg_conv2d_fprop_workspace_bytes(int B, int H, int W, int Cin, int Cout, int R, unsigned flags) {
  if (flags & ICG_UPSAMPLE2X) return 0;
  return fprop_splitk_bytes((long)B * H * W, Cout, R * R * Cin, Cin);
}

// icg_conv2d_fprop with an optional workspace (icg_conv2d_fprop_workspace_bytes; NULL / too small = single pass)
extern "C" int icg_conv2d_fprop_ws(const float* x, const float* w, const float* bias, const float* residual, float* out,
                                   const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin,
                                   int Cout, int R, unsigned flags, float alpha, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  return conv2d_fprop_impl(x, w, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, R, flags, alpha,
                           workspace, workspace_bytes, stream);
}

extern "C" int icg_conv2d_fprop(const float* x, const float* w, const float* bias, const float* residual,
                                float* out, const float* scale, const float* shift, int64_t ss_bstride, int B,
                                int H, int W, int Cin, int Cout, int R, unsigned flags, float alpha,
                                void* stream) {
  return conv2d_fprop_impl(x, w, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, R, flags, alpha,
                           nullptr, 0, stream);
}

static int conv2d_fprop_impl(const float* x, const float* w, const float* bias, const float* residual,
                             float* out, const float* scale, const float* shift, int64_t ss_bstride, int B,
                             int H, int W, int Cin, int Cout, int R, unsigned flags, float alpha, void* workspace,
                             size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && w && out);
  ICG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (R == 1 || R == 3));
  const int up = (flags & ICG_UPSAMPLE2X) ? 1 : 0;
  if (up) ICG_REQUIRE((H % 2 == 0) && (W % 2 == 0));
  if (flags & ICG_RES_UPSAMPLE2X) ICG_REQUIRE(residual && (H % 2 == 0) && (W % 2 == 0) && !(flags & ICG_RES_RELU_MASK));
  if (flags & ICG_RES_RELU_MASK) ICG_REQUIRE(residual != nullptr);
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift);
  const long M = (long)B * H * W;
  ICG_REQUIRE(M < 0x7fffffffL);
  if (!up && !residual && aligned16(x) && M * Cin < 0x7fffffffL &&
      icg_narrow_conv_ok(Cin, Cout, R, w, (flags & ICG_PRE_AFFINE) ? scale : nullptr,
                         (flags & ICG_PRE_AFFINE) ? shift : nullptr, (flags & ICG_PRE_AFFINE) ? ss_bstride : 0))
  {
    g_last_variant[0] = -2; g_last_variant[1] = 0; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    return icg_narrow_fprop(x, w, bias, scale, shift, ss_bstride, out, B, H, W, Cin, Cout,
                            (flags & ICG_PRE_AFFINE) ? 1 : 0, (flags & ICG_PRE_RELU) ? 1 : 0, alpha, (hipStream_t)stream);
  }
  if (!up && !residual && !(flags & (ICG_PRE_AFFINE | ICG_PRE_RELU)) && icg_thin_conv_ok(Cin, Cout, R) && aligned16(out) &&
      (!bias || aligned16(bias)) && M * Cout < 0x7fffffffL) {
    g_last_variant[0] = -4; g_last_variant[1] = 0; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    return icg_thin_fprop(x, w, bias, out, B, H, W, Cin, Cout, R, alpha, (hipStream_t)stream);
  }
  // dense layers on 16 .. 256 rows (every linear layer of the networks at a training batch, and their data gradients): the
  // row-streaming GEMM; icg_skinny_ok: the K % 4 != 0 conditional-BN projections that first had a kernel of their own.  Below 16 rows
  // (the 2- / 4-image test networks) the layers keep the MFMA kernel: one of those networks (BigGAN-deep at 64 x 64, 8 base channels,
  // batch 4) turns a change of summation order in these 24-term dot products into 3.6 % of the rms of its class-embedding gradient --
  // both orders are correct to 4e-7 per layer (tools/gpu_r3_v.sh), the golden tolerance of that ill-conditioned toy is not the
  // place to absorb it
  static const bool dense_rows = [] { const char* e = getenv("ICG_SMALLM_DENSE"); return !(e && e[0] == '0'); }();       // measurement switch
  if (!up && !residual && !(flags & (ICG_PRE_AFFINE | ICG_PRE_RELU)) &&
      (icg_skinny_ok(M, Cin, R) || (dense_rows && R == 1 && H == 1 && W == 1 && M >= 16 && M <= 256 && Cin >= 16))) {
    g_last_variant[0] = -3; g_last_variant[1] = 0; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    static const bool first_gen = [] { const char* e = getenv("ICG_SKINNY_FIRST_GEN"); return e && e[0] == '1'; }();     // measurement switch
    if (first_gen) return icg_skinny_fprop(x, w, bias, out, (int)M, Cout, Cin, alpha, (hipStream_t)stream);
    return smallm_nt_launch(x, w, bias, out, (int)M, Cout, Cin, alpha, (hipStream_t)stream);
  }
  GemmP p{};
  p.A = x; p.B = w; p.C = out;
  p.M = (int)M; p.N = Cout; p.K = R * R * Cin;
  p.H = H; p.W = W; p.Cin = Cin; p.R = R; p.up = up; p.Hs = H >> up; p.Ws = W >> up;
  p.pad_h = R >> 1; p.pad_w = R >> 1; p.gs = 1; p.Hb = H; p.Wb = W;
  p.scale = scale; p.shift = shift; p.ss_bstride = ss_bstride;
  p.pre_affine = (flags & ICG_PRE_AFFINE) ? 1 : 0;
  p.pre_relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.ldb = p.K; p.ldc = Cout;
  p.bias = bias; p.res = residual; p.res_up = icg_res_mode(flags);
  p.alpha = alpha;
  p.kchunk = 0;
  bool vec = (Cin % 4 == 0) && aligned16(x) && aligned16(w);
  if (p.pre_affine) vec = vec && (ss_bstride % 4 == 0) && aligned16(scale) && aligned16(shift);
  const bool small = ((long)B * p.Hs * p.Ws * Cin < 0x7fffffffL) && ((long)Cout * p.K < 0x7fffffffL) &&
                     ((long)B * (ss_bstride > 0 ? ss_bstride : 0) + Cin < 0x7fffffffL);
  if (!up) {
    const int rc = launch_fprop_splitk(p, vec, (hipStream_t)stream, small, workspace, workspace_bytes);
    if (rc != -1) return rc;
  }
  // 1x1 convolution without a prologue = a plain [M][Cin] x [Cout][Cin]^T GEMM: the persistent body with the bias / residual /
  // ReLU-mask epilogue (the shortcut convolutions, the attention projections, every 1x1 data gradient of BigGAN-deep)
  if (R == 1 && !up && !p.pre_affine && !p.pre_relu && p.res_up != 1) p.plain = 2;
  return launch_gemm<A_K, B_K>(p, vec, 1, (hipStream_t)stream, small);
}

struct WgradPlan {
  int splits;
  int kchunk;
};

static WgradPlan wgrad_plan(long K, int M, int N) {
  const int tn = pick_tn(N);
  const long tiles = icg_cdiv(M, 128) * icg_cdiv(N, 32 * tn);
  long splits = 2048 / (tiles > 0 ? tiles : 1);
  const long ksteps = icg_cdiv(K, 16);
  if (splits > ksteps / 8) splits = ksteps / 8;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  long kchunk = icg_cdiv(icg_cdiv(K, splits), 16) * 16;
  splits = icg_cdiv(K, kchunk);
  WgradPlan pl;
  pl.splits = (int)splits;
  pl.kchunk = (int)kchunk;
  return pl;
}

extern "C" size_t icg_gemm_tn_batched_workspace_bytes(int M, int N, int K, int batch);
extern "C" int icg_gemm_tn_batched(const float* A, const float* B, float* C, int M, int N, int K, int64_t strideA,
                                   int64_t strideB, int64_t strideC, int batch, void* workspace, size_t workspace_bytes,
                                   void* stream);

// The weight gradient of a prologue-free 1x1 convolution is the plain product x^T dy over the pixels: the shape of the Winograd
// weight-gradient plane GEMMs with ONE plane, so it runs on their second-generation kernel (icg_pgemm_tn_kernel, pgemm.hip: 0.74 of the
// fp32 MFMA peak against 0.5 - 0.65 for the first-generation TN path on these K = 10^4 ... 10^5 chains) -- the 1x1 shortcuts of every
// block and the attention block's projections.  (ICG_WGRAD_1X1_TN=0: the first-generation path, a measurement switch.)
static bool wgrad_1x1_tn_ok(long K, int Cin, int Cout, int R, unsigned flags) {
  static const bool on = [] { const char* e = getenv("ICG_WGRAD_1X1_TN"); return !(e && e[0] == '0'); }();
  return on && R == 1 && !(flags & (ICG_UPSAMPLE2X | ICG_PRE_AFFINE | ICG_PRE_RELU)) && (Cout % 96 == 0 || Cout % 128 == 0) &&
         Cin % 4 == 0 && Cin >= 64 && K % 32 == 0 && K >= 4096 && K < 0x7fffffffL;
}

extern "C" size_t icg_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int R) {
  const long K = (long)B * H * W;
  const int M = R * R * Cin;
  WgradPlan pl = wgrad_plan(K, M, Cout);
  size_t need = (pl.splits <= 1) ? 16 : (size_t)pl.splits * (size_t)M * (size_t)Cout * sizeof(float);
  if (wgrad_1x1_tn_ok(K, Cin, Cout, R, 0)) {
    const size_t nn = icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)K, 1);
    if (nn > need) need = nn;
  }
  if (icg_narrow_conv_ok(Cin, Cout, R, nullptr, nullptr, nullptr, 0)) {      // the direct kernel may be chosen at run time
    const size_t nn = icg_narrow_wgrad_workspace_bytes(B, H, W, Cin, Cout);
    if (nn > need) need = nn;
  }
  if (icg_thin_conv_ok(Cin, Cout, R)) {
    const size_t nn = icg_thin_wgrad_workspace_bytes(B, H, W, Cin, Cout);
    if (nn > need) need = nn;
  }
  return need;
}

extern "C" int icg_conv2d_wgrad(const float* x, const float* dy, float* dw, const float* scale,
                                const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin, int Cout,
                                int R, unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw);
  ICG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (R == 1 || R == 3));
  const int up = (flags & ICG_UPSAMPLE2X) ? 1 : 0;
  if (up) ICG_REQUIRE((H % 2 == 0) && (W % 2 == 0));
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift);
  const long K = (long)B * H * W;
  ICG_REQUIRE(K < 0x7fffffffL);
  const int M = R * R * Cin;
  if (!up && aligned16(x) && K * Cin < 0x7fffffffL &&
      icg_narrow_conv_ok(Cin, Cout, R, (flags & ICG_PRE_AFFINE) ? scale : nullptr,
                         (flags & ICG_PRE_AFFINE) ? shift : nullptr, nullptr, (flags & ICG_PRE_AFFINE) ? ss_bstride : 0)) {
    if (workspace == nullptr || workspace_bytes < icg_narrow_wgrad_workspace_bytes(B, H, W, Cin, Cout))
      return ICG_ERR_WORKSPACE;
    g_last_variant[0] = -2; g_last_variant[1] = 1; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    return icg_narrow_wgrad(x, dy, scale, shift, ss_bstride, dw, workspace, B, H, W, Cin, Cout,
                            (flags & ICG_PRE_AFFINE) ? 1 : 0, (flags & ICG_PRE_RELU) ? 1 : 0, (hipStream_t)stream);
  }
  if (!up && !(flags & (ICG_PRE_AFFINE | ICG_PRE_RELU)) && icg_thin_conv_ok(Cin, Cout, R) && aligned16(dy) && aligned16(dw)) {
    if (workspace == nullptr || workspace_bytes < icg_thin_wgrad_workspace_bytes(B, H, W, Cin, Cout)) return ICG_ERR_WORKSPACE;
    g_last_variant[0] = -4; g_last_variant[1] = 1; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    return icg_thin_wgrad(x, dy, dw, workspace, B, H, W, Cin, Cout, R, (hipStream_t)stream);
  }
  if (!up && !(flags & (ICG_PRE_AFFINE | ICG_PRE_RELU)) && (Cout % 4 == 0) && aligned16(dy) && aligned16(dw) &&
      icg_skinny_ok(K, Cin, R)) {
    g_last_variant[0] = -3; g_last_variant[1] = 1; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    return icg_skinny_wgrad(x, dy, dw, (int)K, Cout, Cin, (hipStream_t)stream);
  }
  if (wgrad_1x1_tn_ok(K, Cin, Cout, R, flags) && aligned16(x) && aligned16(dy) && aligned16(dw) && workspace != nullptr &&
      workspace_bytes >= icg_gemm_tn_batched_workspace_bytes(Cin, Cout, (int)K, 1)) {
    icg_gemm_mark_planes(1);
    const int rc = icg_gemm_tn_batched(x, dy, dw, Cin, Cout, (int)K, 0, 0, (int64_t)Cin * Cout, 1, workspace, workspace_bytes, stream);
    icg_gemm_mark_planes(0);
    return rc;
  }
  WgradPlan pl = wgrad_plan(K, M, Cout);
  const size_t need = (pl.splits <= 1) ? 0 : (size_t)pl.splits * M * Cout * sizeof(float);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) return ICG_ERR_WORKSPACE;
  GemmP p{};
  p.A = x; p.B = dy;
  p.C = (pl.splits <= 1) ? dw : (float*)workspace;
  p.M = M; p.N = Cout; p.K = (int)K;
  p.H = H; p.W = W; p.Cin = Cin; p.R = R; p.up = up; p.Hs = H >> up; p.Ws = W >> up;
  p.pad_h = R >> 1; p.pad_w = R >> 1; p.gs = 1; p.Hb = H; p.Wb = W;
  p.scale = scale; p.shift = shift; p.ss_bstride = ss_bstride;
  p.pre_affine = (flags & ICG_PRE_AFFINE) ? 1 : 0;
  p.pre_relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.ldb = Cout; p.ldc = Cout;
  p.alpha = 1.f;
  p.kchunk = pl.kchunk;
  p.strideA = 0; p.strideB = 0; p.strideC = (long)M * Cout;
  bool vec_a = (Cin % 4 == 0) && aligned16(x);
  if (p.pre_affine) vec_a = vec_a && (ss_bstride % 4 == 0) && aligned16(scale) && aligned16(shift);
  const bool vec = vec_a && (Cout % 4 == 0) && aligned16(dy);
  const bool small = ((long)B * p.Hs * p.Ws * Cin < 0x7fffffffL) && (K * (long)Cout < 0x7fffffffL) &&
                     ((long)B * (ss_bstride > 0 ? ss_bstride : 0) + Cin < 0x7fffffffL);
  int rc = launch_gemm<A_M, B_N>(p, vec, pl.splits, (hipStream_t)stream, small, vec_a);
  if (rc != ICG_OK) return rc;
  if (pl.splits > 1) {
    const long n = (long)M * Cout;
    launch_splitk_reduce((const float*)workspace, dw, n, pl.splits, (hipStream_t)stream);
    rc = icg_check_launch();
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Nearest-x2-upsample-fused 3x3 convolution as 4 phases of 2x2 taps at SOURCE resolution (2.25x fewer MACs than
// convolving the upsampled tensor, which is what the reference does: layers.py:545-548, BigGAN.py:260).
//   out[b, 2h+al, 2w+be, co] = sum_{u,v,ci} act(x)[b, h+al-1+u, w+be-1+v, ci] * wp[al][be][co][u][v][ci] + bias[co]
// wp = phase weights emitted by icg_sn_forward (sums of the 3x3 taps that land on the same source pixel).
extern "C" int icg_conv2d_up_fprop(const float* x, const float* wp, const float* bias, float* out, const float* scale,
                                   const float* shift, int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout,
                                   unsigned flags, void* stream) {
  ICG_REQUIRE(x && wp && out && B > 0 && Hs > 0 && Ws > 0 && Cin > 0 && Cout > 0);
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift);
  const long M = (long)B * Hs * Ws;
  ICG_REQUIRE(M * 4 < 0x7fffffffL);
  GemmP p{};
  p.A = x; p.B = wp; p.C = out;
  p.M = (int)M; p.N = Cout; p.K = 4 * Cin;
  p.H = Hs; p.W = Ws; p.Cin = Cin; p.R = 2; p.up = 0; p.Hs = Hs; p.Ws = Ws;
  p.pad_h = 1; p.pad_w = 1; p.gs = 1; p.Hb = Hs; p.Wb = Ws;
  p.scale = scale; p.shift = shift; p.ss_bstride = ss_bstride;
  p.pre_affine = (flags & ICG_PRE_AFFINE) ? 1 : 0;
  p.pre_relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.ldb = p.K; p.ldc = Cout;
  p.bias = bias; p.alpha = 1.f;
  p.kchunk = 0; p.phase_mode = 1; p.nsplit = 1;
  p.strideA = 0; p.strideB = (long)Cout * p.K; p.strideC = 0;
  bool vec = (Cin % 4 == 0) && aligned16(x) && aligned16(wp);
  if (p.pre_affine) vec = vec && (ss_bstride % 4 == 0) && aligned16(scale) && aligned16(shift);
  const bool small = (M * Cin < 0x7fffffffL) && ((long)Cout * p.K < 0x7fffffffL) &&
                     ((long)B * (ss_bstride > 0 ? ss_bstride : 0) + Cin < 0x7fffffffL);
  return launch_gemm<A_K, B_K>(p, vec, 4, (hipStream_t)stream, small);
}

// Data gradient of the same layer, produced directly at SOURCE resolution (the 2x2 adjoint of the upsample is part of
// the contraction): a 4x4 / stride-2 / pad-1 gather over dy with K = 16*Cout.
//   da[b,h,w,ci] = sum_{P,Q,co} dy[b, 2h-1+P, 2w-1+Q, co] * vd[ci][P][Q][co]
extern "C" int icg_conv2d_up_dgrad(const float* dy, const float* vd, float* da, int B, int Hs, int Ws, int Cin,
                                   int Cout, void* stream) {
  ICG_REQUIRE(dy && vd && da && B > 0 && Hs > 0 && Ws > 0 && Cin > 0 && Cout > 0);
  const long M = (long)B * Hs * Ws;
  ICG_REQUIRE(M * 4 < 0x7fffffffL);
  GemmP p{};
  p.A = dy; p.B = vd; p.C = da;
  p.M = (int)M; p.N = Cin; p.K = 16 * Cout;
  p.H = Hs; p.W = Ws; p.Cin = Cout; p.R = 4; p.up = 0; p.Hs = 2 * Hs; p.Ws = 2 * Ws;
  p.pad_h = 1; p.pad_w = 1; p.gs = 2; p.Hb = 2 * Hs; p.Wb = 2 * Ws;
  p.ldb = p.K; p.ldc = Cin;
  p.alpha = 1.f;
  p.kchunk = 0; p.phase_mode = 0; p.nsplit = 1;
  const bool vec = (Cout % 4 == 0) && aligned16(dy) && aligned16(vd);
  const bool small = (M * 4 * Cout < 0x7fffffffL) && ((long)Cin * p.K < 0x7fffffffL);
  return launch_gemm<A_K, B_K>(p, vec, 1, (hipStream_t)stream, small);
}

// Weight gradient in phase form: dwp[al][be][u][v][ci][co] = sum_{b,h,w} act(x)[b,h+al-1+u,w+be-1+v,ci] *
// dy[b,2h+al,2w+be,co]; icg_sn_backward folds the 16 phase taps back onto the 3x3 parameter.
static WgradPlan up_wgrad_plan(long K, int M, int N) {
  WgradPlan pl = wgrad_plan(K, M, N);
  int s = pl.splits / 4;            // four phases share the grid
  if (s < 1) s = 1;
  long kchunk = icg_cdiv(icg_cdiv(K, s), 16) * 16;
  pl.splits = (int)icg_cdiv(K, kchunk);
  pl.kchunk = (int)kchunk;
  return pl;
}

extern "C" size_t icg_conv2d_up_wgrad_workspace_bytes(int B, int Hs, int Ws, int Cin, int Cout) {
  WgradPlan pl = up_wgrad_plan((long)B * Hs * Ws, 4 * Cin, Cout);
  if (pl.splits <= 1) return 16;
  return (size_t)4 * pl.splits * (size_t)(4 * Cin) * (size_t)Cout * sizeof(float);
}

extern "C" int icg_conv2d_up_wgrad(const float* x, const float* dy, float* dwp, const float* scale, const float* shift,
                                   int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout, unsigned flags,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dwp && B > 0 && Hs > 0 && Ws > 0 && Cin > 0 && Cout > 0);
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift);
  const long K = (long)B * Hs * Ws;
  ICG_REQUIRE(K * 4 < 0x7fffffffL);
  const int M = 4 * Cin;
  WgradPlan pl = up_wgrad_plan(K, M, Cout);
  const size_t need = (pl.splits <= 1) ? 0 : (size_t)4 * pl.splits * M * Cout * sizeof(float);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) return ICG_ERR_WORKSPACE;
  GemmP p{};
  p.A = x; p.B = dy;
  p.C = (pl.splits <= 1) ? dwp : (float*)workspace;
  p.M = M; p.N = Cout; p.K = (int)K;
  p.H = Hs; p.W = Ws; p.Cin = Cin; p.R = 2; p.up = 0; p.Hs = Hs; p.Ws = Ws;
  p.pad_h = 1; p.pad_w = 1; p.gs = 1; p.Hb = Hs; p.Wb = Ws;
  p.scale = scale; p.shift = shift; p.ss_bstride = ss_bstride;
  p.pre_affine = (flags & ICG_PRE_AFFINE) ? 1 : 0;
  p.pre_relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.ldb = Cout; p.ldc = Cout;
  p.alpha = 1.f;
  p.kchunk = pl.kchunk; p.phase_mode = 2; p.nsplit = pl.splits;
  p.strideA = 0; p.strideB = 0; p.strideC = (long)M * Cout;
  bool vec = (Cin % 4 == 0) && (Cout % 4 == 0) && aligned16(x) && aligned16(dy);
  if (p.pre_affine) vec = vec && (ss_bstride % 4 == 0) && aligned16(scale) && aligned16(shift);
  const bool small = (K * Cin < 0x7fffffffL) && (4 * K * (long)Cout < 0x7fffffffL) &&
                     ((long)B * (ss_bstride > 0 ? ss_bstride : 0) + Cin < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_gemm<A_M, B_N>(p, vec, 4 * pl.splits, st, small);
  if (rc != ICG_OK) return rc;
  if (pl.splits > 1) {
    const long n = (long)M * Cout;
    for (int ph = 0; ph < 4; ++ph)
      launch_splitk_reduce((const float*)workspace + (long)ph * pl.splits * n, dwp + (long)ph * n, n, pl.splits, st);
    rc = icg_check_launch();
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 convolution followed by 2x2 average pooling (DBlock conv2 -> nn.AvgPool2d(2): layers.py:603-606, BigGAN.py:528)
// as ONE 4x4 / stride-2 / pad-1 convolution at the pooled resolution (2.25x fewer MACs, no full-resolution output):
//   out[b,hp,wp,co] = sum_{P,Q,ci} act(x)[b, 2hp-1+P, 2wp-1+Q, ci] * vdn[co][P][Q][ci] + bias[co] + residual[b,hp,wp,co]
// x is [B][2Hp][2Wp][Cin]; vdn = [Cout][4][4][Cin] from icg_sn_forward (0.25 * sums of the 3x3 taps).
extern "C" int icg_conv2d_down_fprop(const float* x, const float* vdn, const float* bias, const float* residual,
                                     float* out, int B, int Hp, int Wp, int Cin, int Cout, unsigned flags,
                                     void* stream) {
  ICG_REQUIRE(x && vdn && out && B > 0 && Hp > 0 && Wp > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE(!(flags & (ICG_PRE_AFFINE | ICG_UPSAMPLE2X | ICG_RES_UPSAMPLE2X)));
  const long M = (long)B * Hp * Wp;
  ICG_REQUIRE(M * 4 < 0x7fffffffL);
  GemmP p{};
  p.A = x; p.B = vdn; p.C = out;
  p.M = (int)M; p.N = Cout; p.K = 16 * Cin;
  p.H = Hp; p.W = Wp; p.Cin = Cin; p.R = 4; p.up = 0; p.Hs = 2 * Hp; p.Ws = 2 * Wp;
  p.pad_h = 1; p.pad_w = 1; p.gs = 2; p.Hb = 2 * Hp; p.Wb = 2 * Wp;
  p.pre_relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.ldb = p.K; p.ldc = Cout;
  p.bias = bias; p.res = residual; p.alpha = 1.f;
  p.kchunk = 0; p.phase_mode = 0; p.nsplit = 1;
  const bool vec = (Cin % 4 == 0) && aligned16(x) && aligned16(vdn);
  const bool small = (M * 4 * Cin < 0x7fffffffL) && ((long)Cout * p.K < 0x7fffffffL);
  return launch_gemm<A_K, B_K>(p, vec, 1, (hipStream_t)stream, small);
}

// data gradient of the same layer: 4 phases of 2x2 taps scattering from the pooled gradient to full resolution
//   da[b, 2hp+al, 2wp+be, ci] = sum_{u,v,co} dy[b, hp+al-1+u, wp+be-1+v, co] * wq[al][be][ci][u][v][co]
static int down_dgrad_impl(const float* dy, const float* wq, const float* relu_in, float* da, int B, int Hp, int Wp, int Cin,
                           int Cout, void* stream) {
  ICG_REQUIRE(dy && wq && da && B > 0 && Hp > 0 && Wp > 0 && Cin > 0 && Cout > 0);
  const long M = (long)B * Hp * Wp;
  ICG_REQUIRE(M * 4 < 0x7fffffffL);
  GemmP p{};
  p.A = dy; p.B = wq; p.C = da;
  p.M = (int)M; p.N = Cin; p.K = 4 * Cout;
  p.H = Hp; p.W = Wp; p.Cin = Cout; p.R = 2; p.up = 0; p.Hs = Hp; p.Ws = Wp;
  p.pad_h = 1; p.pad_w = 1; p.gs = 1; p.Hb = Hp; p.Wb = Wp;
  p.ldb = p.K; p.ldc = Cin;
  p.alpha = 1.f;
  p.res = relu_in; p.res_up = relu_in ? 2 : 0;      // the output rows of the phase scatter index the mask as well
  p.kchunk = 0; p.phase_mode = 1; p.nsplit = 1;
  p.strideA = 0; p.strideB = (long)Cin * p.K; p.strideC = 0;
  const bool vec = (Cout % 4 == 0) && aligned16(dy) && aligned16(wq);
  const bool small = (M * Cout < 0x7fffffffL) && ((long)Cin * p.K < 0x7fffffffL);
  return launch_gemm<A_K, B_K>(p, vec, 4, (hipStream_t)stream, small);
}

extern "C" int icg_conv2d_down_dgrad(const float* dy, const float* wq, float* da, int B, int Hp, int Wp, int Cin,
                                     int Cout, void* stream) {
  return down_dgrad_impl(dy, wq, nullptr, da, B, Hp, Wp, Cin, Cout, stream);
}

extern "C" int icg_conv2d_down_dgrad_relu(const float* dy, const float* wq, const float* relu_in, float* dx, int B, int Hp,
                                          int Wp, int Cin, int Cout, void* stream) {
  ICG_REQUIRE(relu_in != nullptr);
  return down_dgrad_impl(dy, wq, relu_in, dx, B, Hp, Wp, Cin, Cout, stream);
}

// weight gradient w.r.t. the 4x4 kernel, HWIO: dvdn[P][Q][ci][co] = sum act(x)[b,2hp-1+P,2wp-1+Q,ci] * dy[b,hp,wp,co]
extern "C" size_t icg_conv2d_down_wgrad_workspace_bytes(int B, int Hp, int Wp, int Cin, int Cout) {
  WgradPlan pl = wgrad_plan((long)B * Hp * Wp, 16 * Cin, Cout);
  if (pl.splits <= 1) return 16;
  return (size_t)pl.splits * (size_t)(16 * Cin) * (size_t)Cout * sizeof(float);
}

extern "C" int icg_conv2d_down_wgrad(const float* x, const float* dy, float* dvdn, int B, int Hp, int Wp, int Cin,
                                     int Cout, unsigned flags, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  ICG_REQUIRE(x && dy && dvdn && B > 0 && Hp > 0 && Wp > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE(!(flags & (ICG_PRE_AFFINE | ICG_UPSAMPLE2X)));
  const long K = (long)B * Hp * Wp;
  ICG_REQUIRE(K * 4 < 0x7fffffffL);
  const int M = 16 * Cin;
  WgradPlan pl = wgrad_plan(K, M, Cout);
  const size_t need = (pl.splits <= 1) ? 0 : (size_t)pl.splits * M * Cout * sizeof(float);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) return ICG_ERR_WORKSPACE;
  GemmP p{};
  p.A = x; p.B = dy;
  p.C = (pl.splits <= 1) ? dvdn : (float*)workspace;
  p.M = M; p.N = Cout; p.K = (int)K;
  p.H = Hp; p.W = Wp; p.Cin = Cin; p.R = 4; p.up = 0; p.Hs = 2 * Hp; p.Ws = 2 * Wp;
  p.pad_h = 1; p.pad_w = 1; p.gs = 2; p.Hb = 2 * Hp; p.Wb = 2 * Wp;
  p.pre_relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.ldb = Cout; p.ldc = Cout;
  p.alpha = 1.f;
  p.kchunk = pl.kchunk; p.phase_mode = 0; p.nsplit = 1;
  p.strideA = 0; p.strideB = 0; p.strideC = (long)M * Cout;
  const bool vec = (Cin % 4 == 0) && (Cout % 4 == 0) && aligned16(x) && aligned16(dy);
  const bool small = (K * 4 * Cin < 0x7fffffffL) && (K * (long)Cout < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_gemm<A_M, B_N>(p, vec, pl.splits, st, small);
  if (rc != ICG_OK) return rc;
  if (pl.splits > 1) {
    const long n = (long)M * Cout;
    launch_splitk_reduce((const float*)workspace, dvdn, n, pl.splits, st);
    rc = icg_check_launch();
  }
  return rc;
}

// Transposed 3x3 convolution with stride 2 (zero insertion by 2, gather padding 2) in phase form: output parity (al, be)
// only ever meets the taps of matching parity, so each of the 4 phases is a 2x2-tap stride-1 convolution over x
// (taps {0,2} for an even, {1} for an odd coordinate; the unused slot carries a zero weight) scattered to (2m+al, 2n+be).
// 16 tap slots per 4 outputs instead of the 36 of the zero-inserted gather, and the fast loader applies.
//   wp[al][be][co][u][v][ci]: phase weights (built by the caller from the 3x3 kernel);  m in [0, Hin], n in [0, Win]
//   out[b, 2m+al, 2n+be, co] = bias[co] + sum_{u,v,ci} x[b, m-1+al+u, n-1+be+v, ci] * wp[al][be][co][u][v][ci]
extern "C" int icg_conv2d_tr2_fprop(const float* x, const float* wp, const float* bias, float* out, int B, int Hin,
                                    int Win, int Cin, int Hout, int Wout, int Cout, void* stream) {
  ICG_REQUIRE(x && wp && out && B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE(Hout > 0 && Wout > 0 && Hout <= 2 * Hin + 2 && Wout <= 2 * Win + 2);
  const long M = (long)B * (Hin + 1) * (Win + 1);
  ICG_REQUIRE(M * 4 < 0x7fffffffL);
  GemmP p{};
  p.A = x; p.B = wp; p.C = out;
  p.M = (int)M; p.N = Cout; p.K = 4 * Cin;
  p.H = Hin + 1; p.W = Win + 1; p.Cin = Cin; p.R = 2; p.up = 0; p.Hs = Hin; p.Ws = Win;
  p.pad_h = 1; p.pad_w = 1; p.gs = 1; p.Hb = Hin; p.Wb = Win;
  p.ldb = p.K; p.ldc = Cout;
  p.bias = bias; p.alpha = 1.f;
  p.kchunk = 0; p.phase_mode = 1; p.nsplit = 1;
  p.oH = Hout; p.oW = Wout;
  p.strideA = 0; p.strideB = (long)Cout * p.K; p.strideC = 0;
  const bool vec = (Cin % 4 == 0) && aligned16(x) && aligned16(wp);
  const bool small = ((long)B * Hin * Win * Cin < 0x7fffffffL) && ((long)Cout * p.K < 0x7fffffffL);
  return launch_gemm<A_K, B_K>(p, vec, 4, (hipStream_t)stream, small);
}

// ---- general strided / zero-inserted convolution (StyleGAN2 conv2d_gradfix: conv2d, conv_transpose2d and all their
// first- and second-order gradients are parameterisations of these two gathers) --------------------------------------
//   out[b,oy,ox,co] = bias[co] + sum_{r,s,ci} src(b, oy*stride + r - pad, ox*stride + s - pad, ci) * w[co][r][s][ci]
//   src = x inside [0,Hin)x[0,Win), zero outside                                         (zero_insert = 0)
//   src = x[(h/z, w/z)] where h, w are multiples of z = zero_insert's factor inside the zero-inserted extent
//         [(Hin-1)*z+1] x [(Win-1)*z+1], zero elsewhere; stride must be 1                (zero_insert = z in {2, 4})
static int conv2d_g_fprop_impl(const float* x, const float* w, const float* bias, float* out, int B, int Hin, int Win,
                               int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zero_insert,
                               void* workspace, size_t workspace_bytes, void* stream);

extern "C" size_t icg_conv2d_g_fprop_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int R,
                                                     int zero_insert) {
  if (zero_insert) return 0;
  return fprop_splitk_bytes((long)B * Hout * Wout, Cout, R * R * Cin, Cin);
}

extern "C" int icg_conv2d_g_fprop_ws(const float* x, const float* w, const float* bias, float* out, int B, int Hin,
                                     int Win, int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad,
                                     int zero_insert, void* workspace, size_t workspace_bytes, void* stream) {
  return conv2d_g_fprop_impl(x, w, bias, out, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert, workspace,
                             workspace_bytes, stream);
}

extern "C" int icg_conv2d_g_fprop(const float* x, const float* w, const float* bias, float* out, int B, int Hin, int Win,
                                  int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zero_insert,
                                  void* stream) {
  return conv2d_g_fprop_impl(x, w, bias, out, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert, nullptr, 0,
                             stream);
}

// out[m][n] = bias[n] + sum_{k < K} x[m][k] w[n][k] for K <= 3: the data gradient of StyleGAN2's toRGB layers (dy [M][3] against
// w^T [Cin][3]) and its fromRGB layers -- three multiply-adds per output, bound by the write of out; one float4 of n per thread
__global__ __launch_bounds__(256) void conv1x1_k3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, long M, int N,
                                                         int K) {
  const int N4 = N >> 2;
  const long total = M * N4, gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int n4 = (int)(i % N4);
    const long m = i / N4;
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + 4 * n4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const float xv = x[m * K + k];
      const float* wr = w + (long)(4 * n4) * K + k;
      acc.x = fmaf(xv, wr[0], acc.x); acc.y = fmaf(xv, wr[K], acc.y); acc.z = fmaf(xv, wr[2 * K], acc.z); acc.w = fmaf(xv, wr[3 * K], acc.w);
    }
    *reinterpret_cast<float4*>(out + m * N + 4 * n4) = acc;
  }
}

static int conv2d_g_fprop_impl(const float* x, const float* w, const float* bias, float* out, int B, int Hin, int Win,
                               int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zero_insert,
                               void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && w && out && B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Hout > 0 && Wout > 0 && Cout > 0);
  ICG_REQUIRE(R >= 1 && R <= 7 && stride >= 1 && stride <= 4 && pad >= 0 && pad <= 8);
  ICG_REQUIRE(zero_insert == 0 || ((zero_insert == 2 || zero_insert == 4) && stride == 1));
  const long M = (long)B * Hout * Wout;
  ICG_REQUIRE(M < 0x7fffffffL);
  if (R == 1 && stride == 1 && pad == 0 && zero_insert == 0 && Hout == Hin && Wout == Win && Cin <= 3 && Cout % 4 == 0 && Cout >= 16 &&
      aligned16(out) && (!bias || aligned16(bias))) {
    g_last_variant[0] = -5; g_last_variant[1] = 0; g_last_variant[2] = Cout; g_last_variant[3] = Cin;
    long nb = icg_cdiv(M * (Cout / 4), 256);
    if (nb > 256 * 32) nb = 256 * 32;
    hipLaunchKernelGGL(conv1x1_k3_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, M, Cout, Cin);
    return icg_check_launch();
  }
  GemmP p{};
  p.A = x; p.B = w; p.C = out;
  p.M = (int)M; p.N = Cout; p.K = R * R * Cin;
  p.H = Hout; p.W = Wout; p.Cin = Cin; p.R = R; p.Hs = Hin; p.Ws = Win;
  p.pad_h = pad; p.pad_w = pad; p.gs = stride;
  if (zero_insert) {
    p.up = (zero_insert == 2) ? 1 : 2;
    p.zmask = zero_insert - 1;
    p.Hb = (Hin - 1) * zero_insert + 1; p.Wb = (Win - 1) * zero_insert + 1;
  } else {
    p.up = 0; p.zmask = 0; p.Hb = Hin; p.Wb = Win;
  }
  p.ldb = p.K; p.ldc = Cout;
  p.bias = bias; p.alpha = 1.f;
  p.kchunk = 0; p.phase_mode = 0; p.nsplit = 1;
  const bool vec = (Cin % 4 == 0) && aligned16(x) && aligned16(w);
  // the fast path decodes pixels with shifts: needs power-of-two OUTPUT grid only in A_M mode; A_K needs Cin % 16 == 0
  const bool small = ((long)B * Hin * Win * Cin < 0x7fffffffL) && ((long)Cout * p.K < 0x7fffffffL);
  {
    const int rc = launch_fprop_splitk(p, vec, (hipStream_t)stream, small, workspace, workspace_bytes);
    if (rc != -1) return rc;
  }
  return launch_gemm<A_K, B_K>(p, vec, 1, (hipStream_t)stream, small);
}

//   dw[r][s][ci][co] = sum_{b,oy,ox} x[b, oy*stride + r - pad, ox*stride + s - pad, ci] * dy[b,oy,ox,co]      (HWIO)
extern "C" size_t icg_conv2d_g_wgrad_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int R) {
  WgradPlan pl = wgrad_plan((long)B * Hout * Wout, R * R * Cin, Cout);
  if (pl.splits <= 1) return 16;
  return (size_t)pl.splits * (size_t)(R * R * Cin) * (size_t)Cout * sizeof(float);
}

extern "C" int icg_conv2d_g_wgrad(const float* x, const float* dy, float* dw, int B, int Hin, int Win, int Cin, int Hout,
                                  int Wout, int Cout, int R, int stride, int pad, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw && B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Hout > 0 && Wout > 0 && Cout > 0);
  ICG_REQUIRE(R >= 1 && R <= 7 && stride >= 1 && stride <= 4 && pad >= 0 && pad <= 8);
  const long K = (long)B * Hout * Wout;
  ICG_REQUIRE(K < 0x7fffffffL);
  const int M = R * R * Cin;
  WgradPlan pl = wgrad_plan(K, M, Cout);
  const size_t need = (pl.splits <= 1) ? 0 : (size_t)pl.splits * M * Cout * sizeof(float);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) return ICG_ERR_WORKSPACE;
  GemmP p{};
  p.A = x; p.B = dy;
  p.C = (pl.splits <= 1) ? dw : (float*)workspace;
  p.M = M; p.N = Cout; p.K = (int)K;
  p.H = Hout; p.W = Wout; p.Cin = Cin; p.R = R; p.up = 0; p.Hs = Hin; p.Ws = Win;
  p.pad_h = pad; p.pad_w = pad; p.gs = stride; p.Hb = Hin; p.Wb = Win;
  p.ldb = Cout; p.ldc = Cout;
  p.alpha = 1.f;
  p.kchunk = pl.kchunk; p.phase_mode = 0; p.nsplit = 1;
  p.strideA = 0; p.strideB = 0; p.strideC = (long)M * Cout;
  const bool vec = (Cin % 4 == 0) && (Cout % 4 == 0) && aligned16(x) && aligned16(dy);
  const bool small = ((long)B * Hin * Win * Cin < 0x7fffffffL) && (K * (long)Cout < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_gemm<A_M, B_N>(p, vec, pl.splits, st, small);
  if (rc != ICG_OK) return rc;
  if (pl.splits > 1) {
    const long n = (long)M * Cout;
    launch_splitk_reduce((const float*)workspace, dw, n, pl.splits, st);
    rc = icg_check_launch();
  }
  return rc;
}

// out[b][i] = sum_s slab[b * splits + s][i]   (deterministic order; 16-byte loads, four chains in flight: see icg_splitk_reduce_kernel)
__global__ __launch_bounds__(256) void splitk_reduce_batched_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n,
                                                                    int splits) {
  const long b = blockIdx.y;
  const float* sl = slabs + b * splits * n;
  float* o = out + b * n;
  const long stride = (long)gridDim.x * blockDim.x;
  if (((n & 3) == 0) && ((((uintptr_t)slabs) | ((uintptr_t)out)) & 15) == 0) {
    const long n4 = n >> 2;
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(sl);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      int z = 0;
      for (; z + 4 <= splits; z += 4) {
        const float4 v0 = s4[(long)z * n4 + i], v1 = s4[(long)(z + 1) * n4 + i], v2 = s4[(long)(z + 2) * n4 + i],
                     v3 = s4[(long)(z + 3) * n4 + i];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
        a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      }
      for (; z < splits; ++z) {
        const float4 v0 = s4[(long)z * n4 + i];
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      }
      o4[i] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                          (a0.w + a1.w) + (a2.w + a3.w));
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += sl[(long)z * n + i];
    o[i] = s;
  }
}

static int tn_batched_splits(long K, int M, int N, int batch) {
  const long tiles = icg_cdiv(M, 128) * icg_cdiv(N, 32 * pick_tn(N)) * batch;
  long s = 2048 / (tiles > 0 ? tiles : 1);
  const long ksteps = icg_cdiv(K, 16);
  if (s > ksteps / 8) s = ksteps / 8;
  if (s > 256) s = 256;
  return s < 1 ? 1 : (int)s;
}

extern "C" size_t icg_gemm_tn_batched_workspace_bytes(int M, int N, int K, int batch) {
  const int s = tn_batched_splits(K, M, N, batch);
  return s <= 1 ? 16 : (size_t)batch * s * (size_t)M * N * sizeof(float);
}

// C[b] = A[b]^T B[b] for A [K][M], B [K][N] with a long K (weight-gradient shaped: K = pixels or tiles): K is split into
// slices so that batch x slices x tiles fills the chip; slabs are summed in fixed order.
extern "C" int icg_gemm_tn_batched(const float* A, const float* B, float* C, int M, int N, int K, int64_t strideA,
                                   int64_t strideB, int64_t strideC, int batch, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  ICG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0);
  int S = tn_batched_splits(K, M, N, batch);
  if (S > 1 && (workspace == nullptr || workspace_bytes < (size_t)batch * S * (size_t)M * N * sizeof(float))) S = 1;
  GemmP p{};
  p.A = A; p.B = B;
  p.M = M; p.N = N; p.K = K;
  p.H = 1; p.W = 1; p.R = 1; p.up = 0; p.Hs = 1; p.Ws = 1;
  p.pad_h = 0; p.pad_w = 0; p.gs = 1; p.Hb = 1; p.Wb = 1;
  p.Cin = M; p.ldb = N; p.ldc = N;
  p.alpha = 1.f;
  p.strideA = strideA; p.strideB = strideB;
  const bool al = aligned16(A) && aligned16(B) && (strideA % 4 == 0) && (strideB % 4 == 0);
  const bool vec = al && (M % 4 == 0) && (N % 4 == 0);
  const bool small = ((long)M * K < 0x7fffffffL) && ((long)N * K < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  if (g_gemm_planes && strideC == (int64_t)M * N) {
    // second-generation weight-gradient plane GEMM (pgemm.hip): K-slices in multiples of 32 rows, same slab layout and reduction
    const int kchunk2 = (int)(icg_cdiv(icg_cdiv(K, S > 1 ? S : 1), 32) * 32);
    const int splits2 = (int)icg_cdiv(K, kchunk2);
    int nt2 = 0;
    const int rc2 = icg_pgemm_tn_launch(A, B, splits2 > 1 ? (float*)workspace : C, M, N, K, strideA, strideB, batch, kchunk2, splits2,
                                        2, st, &nt2);
    if (rc2 != 1) {
      if (rc2 != ICG_OK) return rc2;
      g_last_variant[0] = 3; g_last_variant[1] = 1; g_last_variant[2] = nt2; g_last_variant[3] = 2;
      if (splits2 == 1) return ICG_OK;
      const long n2 = (long)M * N;
      long blocks2 = icg_cdiv(n2, 256);
      if (blocks2 > 1024) blocks2 = 1024;
      hipLaunchKernelGGL(splitk_reduce_batched_kernel, dim3((unsigned)blocks2, (unsigned)batch), dim3(256), 0, st,
                         (const float*)workspace, C, n2, splits2);
      return icg_check_launch();
    }
  }
  if (S <= 1) {
    p.C = C; p.strideC = strideC; p.kchunk = 0; p.bsplit = 0;
    return launch_gemm<A_M, B_N>(p, vec, batch, st, small);
  }
  p.kchunk = (int)(icg_cdiv(icg_cdiv(K, S), 16) * 16);
  const int splits = (int)icg_cdiv(K, p.kchunk);
  p.bsplit = splits;
  p.C = (float*)workspace; p.strideC = (long)M * N;
  int rc = launch_gemm<A_M, B_N>(p, vec, batch * splits, st, small);
  if (rc != ICG_OK) return rc;
  ICG_REQUIRE(strideC == (int64_t)M * N);
  const long n = (long)M * N;
  long blocks = icg_cdiv(n, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(splitk_reduce_batched_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, st,
                     (const float*)workspace, C, n, splits);
  return icg_check_launch();
}

extern "C" int icg_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int transA,
                                int transB, int64_t strideA, int64_t strideB, int64_t strideC, int batch,
                                float alpha, void* stream) {
  ICG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0);
  if (transA == 0 && transB == 1 && batch == 1 && M <= 32 && K % 4 == 0 && K >= 128 && N >= 16 && aligned16(A) && aligned16(B)) {
    g_last_variant[0] = -6; g_last_variant[1] = 0; g_last_variant[2] = N; g_last_variant[3] = K;
    return smallm_nt_launch(A, B, nullptr, C, M, N, K, alpha, (hipStream_t)stream);
  }
  GemmP p{};
  p.A = A; p.B = B; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.H = 1; p.W = 1; p.R = 1; p.up = 0; p.Hs = 1; p.Ws = 1;
  p.pad_h = 0; p.pad_w = 0; p.gs = 1; p.Hb = 1; p.Wb = 1;
  p.ldc = N;
  p.alpha = alpha;
  p.kchunk = 0;
  p.strideA = strideA; p.strideB = strideB; p.strideC = strideC;
  const bool al = aligned16(A) && aligned16(B) && (strideA % 4 == 0) && (strideB % 4 == 0);
  hipStream_t st = (hipStream_t)stream;
  const bool small = ((long)M * K < 0x7fffffffL) && ((long)N * K < 0x7fffffffL);
  if (transA == 0 && transB == 1) {        // A [M][K], B [N][K]
    p.Cin = K; p.ldb = K;
    p.plain = 1;
    return launch_gemm<A_K, B_K>(p, al && (K % 4 == 0), batch, st, small);
  } else if (transA == 0 && transB == 0) { // A [M][K], B [K][N]
    p.Cin = K; p.ldb = N;
    return launch_gemm<A_K, B_N>(p, al && (K % 4 == 0) && (N % 4 == 0), batch, st, small);
  } else if (transA == 1 && transB == 0) { // A [K][M], B [K][N]
    p.Cin = M; p.ldb = N;
    return launch_gemm<A_M, B_N>(p, al && (M % 4 == 0) && (N % 4 == 0), batch, st, small);
  }
  return ICG_ERR_ARG;
}
