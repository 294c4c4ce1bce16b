// fp16-input weight gradient of the gather convolution for StyleGAN2's fp16 blocks (the reference: cudnn_convolution_backward_weight
// on fp16 tensors, conv2d_gradfix.py:139-272 with training/networks.py:77-91, 581-601):
//
//     dw[r][s][ci][co] = sum_{b, oy, ox}  x[b, oy * stride + r - pad, ox * stride + s - pad, ci] * dy[b, oy, ox, co]          (HWIO, fp32)
//
// As ONE GEMM  C[m][n] = sum_k A[k][m] B[k][n]  with m = (tap, ci) -- so C is the HWIO tensor itself --, n = co and k = the output
// pixels: A[k][m] is x at the tap-shifted pixel (zero outside the image), B[k][n] = dy.  Both operands are K-STRIDED in memory
// (channels are contiguous, pixels are rows), which is the wrong way round for an MFMA fragment (8 consecutive k per lane).  The
// LDS-DMA image therefore keeps the memory orientation ([pixel][channel], a lane's 16 bytes = 8 channels of one pixel) and the
// fragments are read with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane group the element (i & 3) of
// the 8-byte pieces addressed by lanes (i >> 2) + 4 j, j = 0..3 (tools/probes/tr_probe.py prints exactly that), i.e. with lane s
// pointing at [pixel k0 + (s >> 2)][channel m0 + 4 (s & 3) ..] lane i receives channel m0 + i at pixels k0 .. k0 + 3: two such reads
// are one 8-deep operand of v_mfma_f32_16x16x32_f16.  The four pixel rows a group touches are 256 bytes apart (one bank cycle), so
// the 16-byte chunks of pixel row q are stored rotated by 2 (q & 7) positions -- chosen on the SOURCE side of the DMA, as in pgemm.hip.
//
// K-tile = 32 consecutive output pixels of one output row (tail blocks are zero-filled), 3-slot ring, one s_barrier per K-tile, two
// DMAs per wave and K-tile, hand-counted vmcnt.  Split-K over pixel slices into fp32 slabs + a deterministic reduction.
#include "icg_common.h"
#include <stdlib.h>

typedef float hw_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hw_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 hw_h4 __attribute__((ext_vector_type(4)));

struct HwgradP {
  const _Float16* X;      // [B][Hin][Win][Cin]
  const _Float16* DY;     // [B][Ho][Wo][N]
  float* C;               // [S][Mtot][N]
  int Mtot, N;
  int Hin, Win, Cin, Ho, Wo, R, stride, pad;
  int nxb, nkt, kps;      // 32-pixel blocks per output row, K-tiles in all, K-tiles per slice
  int tiles_m, tiles_n;
};

__device__ __attribute__((aligned(64))) _Float16 g_hw_zero_page[32];

__device__ __forceinline__ void hw_dma16_ptr(const void* lane_src, unsigned lds_dst) {
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

// transposing 8-byte LDS read (see the file comment); the caller waits on lgkmcnt before using the result
__device__ __forceinline__ hw_h4 hw_tr_read(unsigned lds_addr) {
  hw_h4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

// workgroup tile 128 (m) x 128 (n), 8 waves as 4 x 2, wave tile 32 x 64
__global__ __launch_bounds__(512, 4) void icg_hwgrad_kernel(HwgradP p) {
  constexpr int NT = 4, ROWB = 256, A_BYTES = 32 * ROWB, B_BYTES = 32 * ROWB, SLOT = A_BYTES + B_BYTES, NBUF = 3;
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv & 3, wn = wv >> 2;
  const int r = lane & 15, kk = lane >> 4;

  // tile order: (slice, m-tile, n-tile), tiles fastest -- the workgroups that share a pixel slice run together (x / dy from L2)
  const int tiles = p.tiles_m * p.tiles_n;
  const int slice = (int)(blockIdx.x / (unsigned)tiles), tile = (int)(blockIdx.x % (unsigned)tiles);
  const int mt = tile / p.tiles_n, nt = tile % p.tiles_n;
  const int m0 = mt * 128, n0 = nt * 128;
  const int kt0 = slice * p.kps, kt1 = min(kt0 + p.kps, p.nkt), nk = kt1 - kt0;

  // ---- DMA role: pixel row 4 wv + (lane >> 4) of the K-tile, chunk POSITION lane & 15; the chunk stored there is rotated by 2 (row & 7)
  const int prow = 4 * wv + (lane >> 4);
  const int chunk = ((lane & 15) - 2 * (prow & 7)) & 15;             // source chunk (8 channels) of this lane
  const int mg = m0 + 8 * chunk, ng = n0 + 8 * chunk;
  const bool a_ok = mg < p.Mtot, b_ok = ng < p.N;
  const int tap = a_ok ? mg / p.Cin : 0;
  const int a_ci = a_ok ? mg - tap * p.Cin : 0;
  const int a_dr = tap / p.R - p.pad, a_dt = tap % p.R - p.pad;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1024u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * 1024u;

  // load cursor (wave-uniform): K-tile -> (image lb, output row loy, 32-pixel block lxb)
  int lxb = kt0 % p.nxb, lrow = kt0 / p.nxb;
  int loy = lrow % p.Ho, lb = lrow / p.Ho, lkt = 0;
  auto issue_next = [&](unsigned slot_off) {
    const int ox = 32 * lxb + prow;
    const bool pix_ok = ox < p.Wo;
    const int ih = loy * p.stride + a_dr, iw = ox * p.stride + a_dt;
    const bool oka = a_ok & pix_ok & ((unsigned)ih < (unsigned)p.Hin) & ((unsigned)iw < (unsigned)p.Win);
    const size_t apix = ((size_t)lb * (unsigned)p.Hin + (unsigned)ih) * (unsigned)p.Win + (unsigned)iw;
    const _Float16* asrc = oka ? p.X + (apix * (unsigned)p.Cin + (unsigned)a_ci) : g_hw_zero_page;
    hw_dma16_ptr(asrc, ldsA + slot_off);
    const size_t bpix = ((size_t)lb * (unsigned)p.Ho + (unsigned)loy) * (unsigned)p.Wo + (unsigned)ox;
    const _Float16* bsrc = (b_ok & pix_ok) ? p.DY + (bpix * (unsigned)p.N + (unsigned)ng) : g_hw_zero_page;
    hw_dma16_ptr(bsrc, ldsB + slot_off);
    if (++lkt < nk) {                                               // past the end: the last K-tile again (never read)
      if (++lxb == p.nxb) {
        lxb = 0;
        if (++loy == p.Ho) { loy = 0; ++lb; }
      }
    } else {
      lkt = nk;
    }
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  // ---- fragment addresses: lane (r, kk), half h reads the 8 bytes at [pixel 8 kk + 4 h + (r >> 2)][channel base + 4 (r & 3) ..]
  unsigned fa[2][2], fb[NT][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * kk + 4 * h + (r >> 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (32 * wm + 16 * i + 4 * (r & 3)) * 2;
      fa[i][h] = lds_base + (unsigned)(row * ROWB + ((((o >> 4) + 2 * (row & 7)) & 15) << 4) + (o & 15));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int o = (16 * NT * wn + 16 * j + 4 * (r & 3)) * 2;
      fb[j][h] = lds_base + (unsigned)(A_BYTES + row * ROWB + ((((o >> 4) + 2 * (row & 7)) & 15) << 4) + (o & 15));
    }
  }

  hw_f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = hw_f32x4{0.f, 0.f, 0.f, 0.f};

  issue_next(0u);
  issue_next((unsigned)SLOT);

  unsigned cur = 0u;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next(next_slot(next_slot(cur)));
    hw_h4 al[2][2], bl[NT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) al[i][h] = hw_tr_read(fa[i][h] + cur);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) bl[j][h] = hw_tr_read(fb[j][h] + cur);
    // the reads above are invisible to the compiler's own s_waitcnt bookkeeping: wait here, and tie every result to the statement
    // so that nothing consuming them can be scheduled above it
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(al[0][0]), "+v"(al[0][1]), "+v"(al[1][0]), "+v"(al[1][1]), "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bl[1][0]),
                   "+v"(bl[1][1]), "+v"(bl[2][0]), "+v"(bl[2][1]), "+v"(bl[3][0]), "+v"(bl[3][1])
                 :
                 : "memory");
    hw_h8 a[2], b[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = __builtin_shufflevector(al[i][0], al[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = __builtin_shufflevector(bl[j][0], bl[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane (r, kk) holds C[32 wm + 16 i + r][64 wn + 16 j + 4 kk .. + 3] of this slice's slab
  float* __restrict__ Cs = p.C + (size_t)slice * (size_t)p.Mtot * (unsigned)p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 16 * i + r;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 16 * NT * wn + 16 * j + 4 * kk;
      if (m < p.Mtot && n < p.N) *reinterpret_cast<hw_f32x4*>(Cs + (size_t)m * (unsigned)p.N + n) = acc[i][j];
    }
  }
}

// out[i] = sum_s slabs[s][i], float4 per thread, fixed summation order
__global__ __launch_bounds__(256) void icg_hwgrad_reduce_kernel(const float4* __restrict__ slabs, float4* __restrict__ out, long n4,
                                                               int slices) {
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gstride) {
    float4 s0 = slabs[i], s1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 1;
    for (; s + 1 < slices; s += 2) {
      const float4 u = slabs[(long)s * n4 + i], v = slabs[(long)(s + 1) * n4 + i];
      s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w;
      s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
    }
    if (s < slices) { const float4 u = slabs[(long)s * n4 + i]; s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w; }
    out[i] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  }
}

struct HwgradPlan { int nxb, nkt, kps, slices, tiles_m, tiles_n; };

static HwgradPlan hwgrad_plan(int B, int Hout, int Wout, int Cin, int Cout, int R) {
  HwgradPlan q;
  q.nxb = (Wout + 31) / 32;
  q.nkt = B * Hout * q.nxb;
  q.tiles_m = (R * R * Cin + 127) / 128;
  q.tiles_n = (Cout + 127) / 128;
  const int tiles = q.tiles_m * q.tiles_n;
  int s = (1024 + tiles - 1) / tiles;                 // ~4 workgroups per CU in all ...
  const int smax = q.nkt / 8 > 0 ? q.nkt / 8 : 1;     // ... of at least 8 K-tiles each
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  q.kps = (q.nkt + s - 1) / s;
  q.slices = (q.nkt + q.kps - 1) / q.kps;
  return q;
}

extern "C" int icg_conv2d_g_wgrad_f16_applies(int Cin, int Cout, int R, int stride) {
  return (Cin >= 32 && Cin % 32 == 0 && Cout >= 32 && Cout % 32 == 0 && R >= 1 && R <= 7 && stride >= 1 && stride <= 4) ? 1 : 0;
}

extern "C" size_t icg_conv2d_g_wgrad_f16_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int R) {
  if (B <= 0 || Hout <= 0 || Wout <= 0 || Cin <= 0 || Cout <= 0 || R <= 0) return 0;
  const HwgradPlan q = hwgrad_plan(B, Hout, Wout, Cin, Cout, R);
  return q.slices > 1 ? (size_t)q.slices * R * R * Cin * Cout * sizeof(float) : 16;
}

extern "C" int icg_conv2d_g_wgrad_f16(const void* x, const void* dy, float* dw, int B, int Hin, int Win, int Cin, int Hout, int Wout,
                                      int Cout, int R, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
  ICG_REQUIRE(x && dy && dw && B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && pad >= 0);
  ICG_REQUIRE(icg_conv2d_g_wgrad_f16_applies(Cin, Cout, R, stride));
  ICG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0);
  ICG_REQUIRE((long)B * Hin * Win < 0x7fffffffL && (long)B * Hout * Wout < 0x7fffffffL);
  const HwgradPlan q = hwgrad_plan(B, Hout, Wout, Cin, Cout, R);
  const size_t need = icg_conv2d_g_wgrad_f16_workspace_bytes(B, Hout, Wout, Cin, Cout, R);
  ICG_REQUIRE(q.slices == 1 || (workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16) == 0));
  HwgradP p{};
  p.X = (const _Float16*)x; p.DY = (const _Float16*)dy;
  p.C = q.slices > 1 ? (float*)workspace : dw;
  p.Mtot = R * R * Cin; p.N = Cout;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Ho = Hout; p.Wo = Wout; p.R = R; p.stride = stride; p.pad = pad;
  p.nxb = q.nxb; p.nkt = q.nkt; p.kps = q.kps; p.tiles_m = q.tiles_m; p.tiles_n = q.tiles_n;
  const long total = (long)q.slices * q.tiles_m * q.tiles_n;
  ICG_REQUIRE(total > 0 && total < 0x7fffffffL);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(icg_hwgrad_kernel, dim3((unsigned)total), dim3(512), 0, st, p);
  int rc = icg_check_launch();
  if (rc != ICG_OK || q.slices == 1) return rc;
  const long n4 = (long)p.Mtot * p.N / 4;
  long nb = icg_cdiv(n4, 256);
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(icg_hwgrad_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const float4*)workspace, (float4*)dw, n4,
                     q.slices);
  return icg_check_launch();
}
