// fp16-input gather convolution for StyleGAN2's fp16 blocks (stylegan2_ada_pytorch/training/networks.py:77-91, 581-601: with
// `num_fp16_res` the reference casts activations AND weights of the highest resolutions to fp16 and lets cuDNN convolve in fp16
// with fp32 accumulation).  One kernel serves every forward / data-gradient contraction of conv2d_gradfix there:
//
//     out[b, oy, ox, n] = sum_{r, s, c}  src[b, (oy * stride + r - pad) / zins, (ox * stride + s - pad) / zins, c] * w[n][r][s][c]
//
// (terms whose coordinate is negative, past the source, or -- zins = 2, the stride-2 transposed convolution as a gather over the
// zero-inserted source -- odd, are zero).  fp16 products are exact in fp32, accumulation is fp32 (v_mfma_f32_16x16x32_f16), the
// result is rounded to fp16 once: the arithmetic of the reference's fp16 path.
//
// Structure = icg_pconv_kernel (pgemm.hip) with 2-byte elements: a K-tile is one tap of one 32-channel slice, i.e. the same 64-byte
// rows, the same lane-linear LDS-DMA image with the source-side chunk permutation, the same 3-slot ring with one raw s_barrier per
// K-tile and hand-counted vmcnt; out-of-image / odd taps fetch from the zero page.  A lane's 16-byte fragment (8 halfs, k = 8 kk ..
// 8 kk + 7 of row r) is exactly one operand of v_mfma_f32_16x16x32_f16, so a K-tile is 2 x NT MFMAs of 16 cycles per wave instead of
// 8 x NT of 32: 16x the fp32 rate per K-tile, which moves the bound of these batch-16 layers from the MFMA pipe to the L2 -> LDS
// stream.  The stride-2 transposed form (zins = 2) runs as four PHASES in one launch: output pixels of parity (al, be) meet only the
// taps r = (pad + al) mod 2, + 2, ... (x likewise), so a workgroup of phase (al, be) walks that tap sub-grid of the same weight
// tensor -- 9 tap visits per 4 outputs instead of 36, no zero operand ever reaches the MFMA.
#include "icg_common.h"
#include <stdlib.h>

typedef float hc_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 hc_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 hc_h4 __attribute__((ext_vector_type(4)));

struct HconvP {
  const _Float16* A;      // [B][Hs][Ws][Cin]
  const _Float16* Bw;     // [N][R][R][Cin]
  _Float16* C;            // [B][Ho][Wo][N]
  int M, N, K;            // M = B Ho Wo, K = R R Cin
  int Ho, Wo, Hs, Ws, Cin, R, stride, pad, zs;      // zs = log2(zero-insertion factor)
  int tiles_n;
  unsigned total;
  int swz;
  // zs = 1: four phases; phase ph = 2 al + be owns the output pixels (2 my + al, 2 mx + be), my < (Ho - al + 1) / 2, and the
  // workgroup tiles [ph_tile0[ph], ph_tile0[ph + 1]) of the launch (pixel-tile major, column-tile minor inside a phase)
  unsigned ph_tile0[5];
  // EP = 1 (zs = 0 only): the layer's epilogue on the accumulators -- y = clamp(gain * act(c * d[b][n] + noise[b][p] * strength + bias[n]))
  // with the fp16 rounding points of the reference's op graph (conv output, fma, bias_act: networks.py:86-94, 432-442) -- stored to Y;
  // the convolution output itself still goes to C when C != null (the demodulation gradient needs it)
  const float* ep_d;
  const float* ep_noise;
  const float* ep_strength;
  const float* ep_bias;
  _Float16* Y;
  long ep_nbs;
  int ep_act;
  float ep_alpha, ep_gain, ep_clamp;
  // MOD = 1: style modulation of the A operand (networks.py:78 `x * styles`): fragment (pixel of sample b, channels c .. c + 7) is
  // multiplied by fp16(sty[b][c .. c + 7]) after its LDS read, one rounding per element -- the tensor x * s is never materialised
  const float* sty;       // [B][Cin] fp32
};

__device__ __attribute__((aligned(64))) _Float16 g_hc_zero_page[32];      // zero-initialised: DMA source of the padding taps

__device__ __forceinline__ int hc_swz(int row) { return (row & 8) ? 3 : 0; }      // = pg_swz (pgemm.hip)

// LDS-DMA, 16 bytes per lane: wave-uniform base + per-lane byte offset / per-lane pointer (see pg_dma16, pgemm.hip)
__device__ __forceinline__ void hc_dma16(const void* gbase, unsigned voff, unsigned lds_dst) {
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
__device__ __forceinline__ void hc_dma16_ptr(const void* lane_src, unsigned lds_dst) {
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

// NT: 16-column MFMA tiles per wave (workgroup tile 128 x 32 NT; 8 waves as 4 x 2, wave tile 32 x 16 NT)
constexpr int HC_MOD_MAXC = 1024;                                   // MOD: the styles of the (at most two) samples a tile meets sit in LDS
template <int NT, int EP = 0, int MOD = 0>
__global__ __launch_bounds__(512, 4) void icg_hconv_kernel(HconvP p) {
  constexpr int BM = 128, BN = 32 * NT, BKC = 32;                   // BKC: channels (halfs) per K-tile = 64 bytes per row
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, SLOT = A_BYTES + B_BYTES, NBUF = 3;
  constexpr int BROWS = BN / 8;                                     // B rows per wave and K-tile: 16 / 12 / 8
  __shared__ __attribute__((aligned(1024))) char lds[NBUF * SLOT + (MOD ? 2 * HC_MOD_MAXC * 2 : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv & 3, wn = wv >> 2;
  const int r = lane & 15, kk = lane >> 4;

  unsigned t = blockIdx.x;
  if (p.swz) {
    const unsigned tot = p.total, q = tot >> 3, rr = tot & 7u, xcd = t & 7u;
    t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (t >> 3);
  }
  // phase of this workgroup (wave-uniform): al / be = parity of its output rows / columns; one phase (0, 0) when zs = 0
  int al = 0, be = 0;
  if (p.zs) {
    const int ph = (t >= p.ph_tile0[2] ? 2 : 0) + ((t >= p.ph_tile0[2] ? t >= p.ph_tile0[3] : t >= p.ph_tile0[1]) ? 1 : 0);
    al = ph >> 1; be = ph & 1;
    t -= p.ph_tile0[ph];
  }
  const int Hph = p.zs ? (p.Ho - al + 1) >> 1 : p.Ho, Wph = p.zs ? (p.Wo - be + 1) >> 1 : p.Wo;      // pixel grid of the phase
  const int Mph = p.zs ? (p.M / (p.Ho * p.Wo)) * Hph * Wph : p.M;
  const int nt = (int)(t % (unsigned)p.tiles_n), mt = (int)(t / (unsigned)p.tiles_n);
  const int m0 = mt * BM, n0 = nt * BN;
  // taps this phase meets: r = r0, r0 + rstep, ... < R (all of them when zs = 0)
  const int rstep = 1 << p.zs, r0 = p.zs ? ((p.pad + al) & 1) : 0, s0 = p.zs ? ((p.pad + be) & 1) : 0;
  const int ntr = (p.R - r0 + rstep - 1) >> p.zs, nts = (p.R - s0 + rstep - 1) >> p.zs;

  // ---- DMA role, A: output pixel 16 wv + (lane >> 2) of the tile, chunk position lane & 3; B: weight row BROWS wv + (lane >> 2)
  const int drowA = 16 * wv + (lane >> 2), drowB = BROWS * wv + (lane >> 2);
  const int am = m0 + drowA;
  const bool a_row_ok = am < Mph;
  const int amm = a_row_ok ? am : 0;
  const int aw = amm % Wph, at = amm / Wph, ah = at % Hph, ab = at / Hph;
  // (zero-inserted) source coordinate of tap (0, 0) of this lane's output pixel (2 ah + al, 2 aw + be when phased)
  const int hs0 = (p.zs ? 2 * ah + al : ah * p.stride) - p.pad, ws0 = (p.zs ? 2 * aw + be : aw * p.stride) - p.pad;
  const unsigned img = (unsigned)ab * (unsigned)(p.Hs * p.Ws);
  const unsigned achunk = 8u * (unsigned)((lane & 3) ^ hc_swz(drowA));       // halfs
  const unsigned voffB =
      ((unsigned)min(n0 + min(drowB, BN - 1), p.N - 1) * (unsigned)p.K + 8u * (unsigned)((lane & 3) ^ hc_swz(drowB))) * 2u;
  const bool dma_b_lane = (BROWS == 16) || ((lane >> 2) < BROWS);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned ldsA = lds_base + (unsigned)wv * 1024u, ldsB = lds_base + (unsigned)A_BYTES + (unsigned)wv * (BROWS * 64u);
  const int nk = ntr * nts * (p.Cin / BKC);                          // K-tiles: the phase's taps x 32-channel slices
  const unsigned hb = (unsigned)(p.Hs << p.zs), wb = (unsigned)(p.Ws << p.zs);

  // load cursor (wave-uniform): K-tile -> (channel slice lc0, tap (ltr, lts) of the phase's tap sub-grid); tap-minor order
  int lc0 = 0, ltr = r0, lts = s0, lkt = 0;
  auto issue_next = [&](unsigned slot_off) {
    const int hi = hs0 + ltr, wi = ws0 + lts;                       // even by construction when zs = 1
    const bool ok = a_row_ok & ((unsigned)hi < hb) & ((unsigned)wi < wb);
    const unsigned pix = img + (unsigned)(hi >> p.zs) * (unsigned)p.Ws + (unsigned)(wi >> p.zs);
    const _Float16* src = ok ? p.A + ((size_t)pix * (unsigned)p.Cin + (unsigned)lc0 + achunk) : g_hc_zero_page;
    hc_dma16_ptr(src, ldsA + slot_off);
    if (dma_b_lane) hc_dma16(p.Bw + ((ltr * p.R + lts) * p.Cin + lc0), voffB, ldsB + slot_off);
    if (++lkt < nk) {                                               // past the end: the last K-tile again (never read)
      lts += rstep;
      if (lts >= p.R) { lts = s0; ltr += rstep; }
      if (ltr >= p.R) { ltr = r0; lc0 += BKC; }
    } else {
      lkt = nk;
    }
  };
  auto next_slot = [](unsigned off) -> unsigned { return off == (unsigned)((NBUF - 1) * SLOT) ? 0u : off + (unsigned)SLOT; };

  const int fo = r * 64 + ((kk ^ hc_swz(r)) * 16);
  const char* fa = lds + fo + (32 * wm) * 64;
  const char* fb = lds + fo + A_BYTES + (16 * NT * wn) * 64;

  hc_f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = hc_f32x4{0.f, 0.f, 0.f, 0.f};

  // MOD: styles of the tile's first sample and the next one as fp16 [2][Cin] behind the ring (a 128-row tile spans at most two
  // samples: the host admits pixel grids of >= 127 per sample only); filled with ordinary loads BEFORE the first DMA is issued, so
  // the hand-counted vmcnt below sees DMAs only
  const char* sfp[2] = {nullptr, nullptr};
  if (MOD) {
    _Float16* stab = reinterpret_cast<_Float16*>(lds + NBUF * SLOT);
    const int hwph = Hph * Wph, nb = p.M / (p.Ho * p.Wo), bt0 = m0 / hwph;
    for (int e = tid; e < 2 * p.Cin; e += 512) {
      const int sel = e >= p.Cin ? 1 : 0, c = e - sel * p.Cin;
      stab[sel * HC_MOD_MAXC + c] = (_Float16)p.sty[(size_t)min(bt0 + sel, nb - 1) * (unsigned)p.Cin + c];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mrow = min(m0 + 32 * wm + 16 * i + r, Mph - 1);
      sfp[i] = reinterpret_cast<const char*>(stab) + ((mrow / hwph - bt0) * HC_MOD_MAXC + 8 * kk) * 2;
    }
    __syncthreads();
  }
  int cc0 = 0, ctap = 0;                                            // consume cursor: channel slice of K-tile kt, tap count inside it
  const int ntaps = ntr * nts;
  hc_h8 sf[2];

  if (nk > 0) {                                                     // (a phase without taps -- 1x1, odd parity -- stores zeros)
    issue_next(0u);
    issue_next((unsigned)SLOT);
  }

  unsigned cur = 0u;
  for (int kt = 0; kt < nk; ++kt) {
    // K-tile kt has landed once at most the two DMAs of K-tile kt + 1 are outstanding; after the barrier the slot of K-tile kt - 1
    // (all of whose LDS reads have returned: lgkmcnt(0)) may be overwritten
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_next(next_slot(next_slot(cur)));
    hc_h8 a[2], b[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const hc_h8*>(fa + cur + i * 1024);
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const hc_h8*>(fb + cur + j * 1024);
    if (MOD) {
      if (ctap == 0) {                                              // (wave-uniform) a new channel slice: its style fragments
        sf[0] = *reinterpret_cast<const hc_h8*>(sfp[0] + cc0 * 2);
        sf[1] = *reinterpret_cast<const hc_h8*>(sfp[1] + cc0 * 2);
      }
      a[0] *= sf[0];                                                // v_pk_mul_f16: x * s rounded to fp16, as the reference's tensor is
      a[1] *= sf[1];
      if (++ctap == ntaps) { ctap = 0; cc0 += BKC; }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    cur = next_slot(cur);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: operands swapped (D = B-fragment x A-fragment), so lane (r, kk) holds C[16 i + r][16 j + 4 kk .. + 3]: one
  // 8-byte store of four halfs per 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 16 * i + r;
    size_t orow = (size_t)m;
    if (p.zs) {                                                     // phase pixel -> output pixel (2 my + al, 2 mx + be)
      const int mm = m < Mph ? m : 0;
      const int mx = mm % Wph, tq = mm / Wph, my = tq % Hph, b = tq / Hph;
      orow = ((size_t)b * (unsigned)p.Ho + (unsigned)(2 * my + al)) * (unsigned)p.Wo + (unsigned)(2 * mx + be);
    }
    int eb = 0;
    float nz = 0.f;
    if (EP) {                                                        // sample and pixel of this row; its noise term, rounded as the fp16 tensor is
      const int hw = p.Ho * p.Wo, mm = m < Mph ? m : 0;
      eb = mm / hw;
      if (p.ep_noise) nz = (float)(_Float16)(p.ep_noise[(long)eb * p.ep_nbs + (mm - eb * hw)] * *p.ep_strength);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + 16 * NT * wn + 16 * j + 4 * kk;
      if (m < Mph && n < p.N) {
        const hc_f32x4 v = acc[i][j];
        hc_h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        if (!EP || p.C) *reinterpret_cast<hc_h4*>(p.C + orow * (unsigned)p.N + n) = o;
        if (EP) {
          hc_f32x4 dv = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
          if (p.ep_d) dv = *reinterpret_cast<const hc_f32x4*>(p.ep_d + (size_t)eb * (unsigned)p.N + n);
          if (p.ep_bias) bv = *reinterpret_cast<const hc_f32x4*>(p.ep_bias + n);
          hc_h4 y;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float z = (float)o[k];
            if (p.ep_d) z = (float)(_Float16)__builtin_fmaf(z, (float)(_Float16)dv[k], nz);
            else if (p.ep_noise) z = (float)(_Float16)(z + nz);
            z += (float)(_Float16)bv[k];
            z = (p.ep_act == 3 && z < 0.f) ? z * p.ep_alpha : z;
            z *= p.ep_gain;
            if (p.ep_clamp >= 0.f) z = fminf(fmaxf(z, -p.ep_clamp), p.ep_clamp);
            y[k] = (_Float16)z;
          }
          *reinterpret_cast<hc_h4*>(p.Y + orow * (unsigned)p.N + n) = y;
        }
      }
    }
  }
}

// 1 when the shape is one the kernel takes (the Python side keeps the fp32 kernel + two casts for everything else: 3-channel
// toRGB / fromRGB layers, channel counts that are not multiples of 32 / 64)
extern "C" int icg_conv2d_g_fprop_f16_applies(int Cin, int Cout, int R, int stride, int zins) {
  if (Cin < 32 || Cin % 32 != 0 || Cout < 64 || Cout % 32 != 0 || (Cout % 128 != 0 && Cout % 96 != 0 && Cout % 64 != 0)) return 0;
  if (R < 1 || R > 7 || stride < 1 || (zins != 0 && zins != 1 && zins != 2) || (zins == 2 && stride != 1)) return 0;
  return 1;
}

static long hconv_min_phase_pixels(int Hout, int Wout, int zins) {
  return zins == 2 ? (long)(Hout / 2) * (Wout / 2) : (long)Hout * Wout;
}

static int hconv_launch(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Hout, int Wout, int Cout, int R, int stride,
                        int pad, int zins, const HconvP* ep, const float* style, void* stream) {
  ICG_REQUIRE(x && w && B > 0 && H > 0 && W > 0 && Hout > 0 && Wout > 0 && pad >= 0 && (y || ep));
  ICG_REQUIRE(icg_conv2d_g_fprop_f16_applies(Cin, Cout, R, stride, zins));
  ICG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 8) == 0);
  const long M = (long)B * Hout * Wout;
  ICG_REQUIRE(M < 0x7fffffffL && (long)B * H * W < 0x7fffffffL && (long)Cout * R * R * Cin < 0x3fffffffL);
  const int nt = (Cout % 128 == 0) ? 4 : ((Cout % 96 == 0) ? 3 : 2);
  HconvP p{};
  if (ep) p = *ep;
  if (style) ICG_REQUIRE(Cin <= HC_MOD_MAXC && hconv_min_phase_pixels(Hout, Wout, zins) >= 127);
  p.sty = style;
  p.A = (const _Float16*)x; p.Bw = (const _Float16*)w; p.C = (_Float16*)y;
  p.M = (int)M; p.N = Cout; p.K = R * R * Cin;
  p.Ho = Hout; p.Wo = Wout; p.Hs = H; p.Ws = W; p.Cin = Cin; p.R = R; p.stride = stride; p.pad = pad; p.zs = (zins == 2) ? 1 : 0;
  p.tiles_n = Cout / (32 * nt);
  long total = icg_cdiv(M, 128) * p.tiles_n;
  if (p.zs) {                                                       // four phases, each with its own pixel grid
    ICG_REQUIRE(!ep);
    total = 0;
    for (int ph = 0; ph < 4; ++ph) {
      const long hp = (Hout - (ph >> 1) + 1) / 2, wp = (Wout - (ph & 1) + 1) / 2;
      p.ph_tile0[ph] = (unsigned)total;
      total += icg_cdiv((long)B * hp * wp, 128) * p.tiles_n;
    }
    p.ph_tile0[4] = (unsigned)total;
  }
  ICG_REQUIRE(total > 0 && total < 0x7fffffffL);
  p.total = (unsigned)total;
  static const bool no_swz = [] { const char* e = getenv("ICG_NO_XCD_SWIZZLE"); return e && e[0] == '1'; }();
  p.swz = (total >= 16 && !no_swz) ? 1 : 0;
  const dim3 grid((unsigned)total), block(512);
  hipStream_t st = (hipStream_t)stream;
#define ICG_HC_LAUNCH(EPV, MODV)                                                                     \
  do {                                                                                               \
    if (nt == 4) hipLaunchKernelGGL((icg_hconv_kernel<4, EPV, MODV>), grid, block, 0, st, p);        \
    else if (nt == 3) hipLaunchKernelGGL((icg_hconv_kernel<3, EPV, MODV>), grid, block, 0, st, p);   \
    else hipLaunchKernelGGL((icg_hconv_kernel<2, EPV, MODV>), grid, block, 0, st, p);                \
  } while (0)
  if (ep && style) ICG_HC_LAUNCH(1, 1);
  else if (ep) ICG_HC_LAUNCH(1, 0);
  else if (style) ICG_HC_LAUNCH(0, 1);
  else ICG_HC_LAUNCH(0, 0);
#undef ICG_HC_LAUNCH
  return icg_check_launch();
}

extern "C" int icg_conv2d_g_fprop_f16(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Hout, int Wout,
                                      int Cout, int R, int stride, int pad, int zins, void* stream) {
  ICG_REQUIRE(y);
  return hconv_launch(x, w, y, B, H, W, Cin, Hout, Wout, Cout, R, stride, pad, zins, nullptr, nullptr, stream);
}

// the same convolution (zero_insert = 0) with the StyleGAN2 layer epilogue on the accumulators (see HconvP): c (may be null) and y
extern "C" int icg_conv2d_g_fprop_f16_act(const void* x, const void* w, void* c, void* y, const float* d, const float* noise, int64_t noise_bstride,
                                          const float* strength, const float* bias, int act, float alpha, float gain, float clamp, int B, int H,
                                          int W, int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, void* stream) {
  return icg_modconv2d_f16(x, nullptr, w, c, y, d, noise, noise_bstride, strength, bias, act, alpha, gain, clamp, B, H, W, Cin, Hout, Wout, Cout, R,
                           stride, pad, 0, stream);
}

extern "C" int icg_modconv2d_f16_applies(int Cin, int Cout, int R, int stride, int zins, int Hout, int Wout) {
  return icg_conv2d_g_fprop_f16_applies(Cin, Cout, R, stride, zins) && Cin <= HC_MOD_MAXC && hconv_min_phase_pixels(Hout, Wout, zins) >= 127;
}

// modulated convolution of a StyleGAN2 layer in ONE launch: style scale on the A fragments (style != null), the contraction, and
// (y != null; zero_insert = 0) demodulation + noise + bias + activation + clamp on the accumulators
extern "C" int icg_modconv2d_f16(const void* x, const float* style, const void* w, void* c, void* y, const float* d, const float* noise,
                                 int64_t noise_bstride, const float* strength, const float* bias, int act, float alpha, float gain, float clamp,
                                 int B, int H, int W, int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zins, void* stream) {
  if (!y) {
    ICG_REQUIRE(c);
    return hconv_launch(x, w, c, B, H, W, Cin, Hout, Wout, Cout, R, stride, pad, zins, nullptr, style, stream);
  }
  ICG_REQUIRE(zins == 0 && (act == 1 || act == 3) && (!noise || strength) && ((uintptr_t)y % 8) == 0);
  ICG_REQUIRE(((uintptr_t)d % 16) == 0 && ((uintptr_t)bias % 16) == 0);
  HconvP ep{};
  ep.ep_d = d; ep.ep_noise = noise; ep.ep_strength = strength; ep.ep_bias = bias; ep.Y = (_Float16*)y; ep.ep_nbs = (long)noise_bstride;
  ep.ep_act = act; ep.ep_alpha = alpha; ep.ep_gain = gain; ep.ep_clamp = clamp;
  return hconv_launch(x, w, c, B, H, W, Cin, Hout, Wout, Cout, R, stride, pad, 0, &ep, style, stream);
}
