// 3x3 convolutions with 1..4 output channels (the generator's to-RGB layer, BigGAN.py:251-262, and the data gradient of
// the discriminator's from-RGB layer) and their weight gradient.  A GEMM tile would waste >= 29/32 of every MFMA on
// them; they are HBM-bound (one pass over the wide tensor), so they get direct kernels:
//   * a group of LP lanes (LP = power of two >= Cin/4, <= 64) owns one image column of a 32-row segment and walks down it, lane q
//     the channel quad q: the 16-byte loads of a group are contiguous, the 9 x NOUT weight quads of a lane stay in registers
//     for the whole kernel, and the 3x3 window slides: 3 new pixels per step instead of 9 (the neighbouring columns belong to
//     the other groups of the same block, so their re-reads hit the L1);
//   * fprop: per-lane partial dot products over the 9 taps, then a log2(LP)-step shuffle reduction per output channel;
//   * wgrad: per-lane accumulators dw[tap][quad][co] over the block's pixels, combined across pixel groups through LDS
//     into one slab per block; the slabs are summed by the deterministic split-K reduction kernel.
// Prologue (per-sample affine + ReLU, zero padding applied AFTER the activation) as in the GEMM loader.
#include "icg_common.h"

#define NC_ROWS 32      // rows of a column segment (2 halo rows per segment are re-read)

typedef float th_f32x16 __attribute__((ext_vector_type(16)));   // MFMA accumulator tile
typedef float nc_v2 __attribute__((ext_vector_type(2)));      // v_pk_fma_f32: two fp32 FMAs per lane and issue slot
__device__ __forceinline__ nc_v2 nc_lo(const float4& v) { return nc_v2{v.x, v.y}; }
__device__ __forceinline__ nc_v2 nc_hi(const float4& v) { return nc_v2{v.z, v.w}; }

__device__ __forceinline__ float4 nc_act(float4 v, const float4& sc, const float4& sh, bool affine, bool relu) {
  if (affine) {
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
  }
  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  return v;
}

// out[b,h,w,o] = alpha * sum_{tap,c} act(x)[b,h+r-1,w+s-1,c] * wgt[o][tap][c] + bias[o]
template <int NOUT, int LP>
__global__ __launch_bounds__(256) void narrow_fprop_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                           const float* __restrict__ bias, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, long ssb,
                                                           float* __restrict__ out, int B, int H, int W, int Cin,
                                                           int affine, int relu, float alpha) {
  constexpr int GPW = 256 / LP;                     // pixel groups per block
  const int q = threadIdx.x % LP, grp = threadIdx.x / LP;
  const int Q = Cin >> 2;
  const bool lane_on = q < Q;
  const unsigned qa = (unsigned)(lane_on ? q : Q - 1);      // address lane (idle lanes re-read the last quad)
  float4 wr[NOUT][9];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      wr[o][t] = lane_on ? *reinterpret_cast<const float4*>(wgt + ((long)o * 9 + t) * Cin + 4 * q)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  // 32-bit index arithmetic throughout (the host checks that every offset fits)
  const unsigned cbs = (unsigned)(W + GPW - 1) / GPW, rss = (unsigned)(H + NC_ROWS - 1) / NC_ROWS;
  const unsigned units = (unsigned)B * cbs * rss;
  for (unsigned u = blockIdx.x; u < units; u += gridDim.x) {
    const unsigned b = u / (cbs * rss), r2 = u - b * cbs * rss;
    const int hb = (int)(r2 / cbs) * NC_ROWS, w0 = (int)(r2 % cbs) * GPW + grp;
    const int he = min(H, hb + NC_ROWS);
    const bool col_on = w0 < W;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (affine) {
      sc = *reinterpret_cast<const float4*>(scale + b * (unsigned)ssb + 4 * qa);
      sh = *reinterpret_cast<const float4*>(shift + b * (unsigned)ssb + 4 * qa);
    }
    const float* xb = x + (b * (unsigned)H * W) * (unsigned)Cin + 4 * qa;
    // raw (unactivated) loads of the three pixels (w0-1, w0, w0+1) of row hi, clamped addresses
    auto load_row = [&](int hi, float4 (&v)[3]) {
      const unsigned hc = (unsigned)min(max(hi, 0), H - 1);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned wc = (unsigned)min(max(w0 + s - 1, 0), W - 1);
        v[s] = *reinterpret_cast<const float4*>(xb + (hc * (unsigned)W + wc) * (unsigned)Cin);
      }
    };
    auto act_row = [&](int hi, float4 (&v)[3]) {      // activation, then the zero padding of the ACTIVATED tensor
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)(w0 + s - 1) < (unsigned)W;
        v[s] = ok ? nc_act(v[s], sc, sh, affine, relu) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    float4 win[3][3], nxt[3];
    load_row(hb - 1, win[0]);
    load_row(hb, win[1]);
    load_row(hb + 1, win[2]);
    act_row(hb - 1, win[0]);
    act_row(hb, win[1]);
    act_row(hb + 1, win[2]);
    for (int h = hb; h < he; ++h) {
      load_row(h + 2, nxt);                            // in flight while this row is computed
      nc_v2 acc2[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) acc2[o] = nc_v2{0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 a = win[t / 3][t % 3];
        const nc_v2 alo = nc_lo(a), ahi = nc_hi(a);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {                                   // wr = 0 on idle lanes
          acc2[o] = __builtin_elementwise_fma(alo, nc_lo(wr[o][t]), acc2[o]);
          acc2[o] = __builtin_elementwise_fma(ahi, nc_hi(wr[o][t]), acc2[o]);
        }
      }
      float acc[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) acc[o] = acc2[o].x + acc2[o].y;
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
#pragma unroll
        for (int d = LP / 2; d >= 1; d >>= 1) acc[o] += __shfl_xor(acc[o], d, 64);
      }
      if (q == 0 && col_on) {
        const long p = ((long)b * H + h) * W + w0;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[p * NOUT + o] = alpha * acc[o] + (bias ? bias[o] : 0.f);
      }
      act_row(h + 2, nxt);
#pragma unroll
      for (int s = 0; s < 3; ++s) { win[0][s] = win[1][s]; win[1][s] = win[2][s]; win[2][s] = nxt[s]; }
    }
  }
}

// slab[block][tap][c][o] = sum over the block's pixels of act(x)[pix + tap][c] * dy[pix][o]      (HWIO, like the GEMM wgrad)
template <int NOUT, int LP>
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           long ssb, float* __restrict__ slabs, int B, int H, int W, int Cin,
                                                           int affine, int relu) {
  constexpr int GPW = 256 / LP;
  __shared__ float red[256 * 4];                     // one float4 per thread at a time
  const int q = threadIdx.x % LP, grp = threadIdx.x / LP;
  const int Q = Cin >> 2;
  const bool lane_on = q < Q;
  const unsigned qa = (unsigned)(lane_on ? q : Q - 1);
  float4 acc[9][NOUT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[t][o] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned cbs = (unsigned)(W + GPW - 1) / GPW, rss = (unsigned)(H + NC_ROWS - 1) / NC_ROWS;
  const unsigned units = (unsigned)B * cbs * rss;
  for (unsigned u = blockIdx.x; u < units; u += gridDim.x) {
    const unsigned b = u / (cbs * rss), r2 = u - b * cbs * rss;
    const int hb = (int)(r2 / cbs) * NC_ROWS, w0 = (int)(r2 % cbs) * GPW + grp;
    const int he = min(H, hb + NC_ROWS);
    const bool col_on = w0 < W;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (affine) {
      sc = *reinterpret_cast<const float4*>(scale + b * (unsigned)ssb + 4 * qa);
      sh = *reinterpret_cast<const float4*>(shift + b * (unsigned)ssb + 4 * qa);
    }
    const float* xb = x + (b * (unsigned)H * W) * (unsigned)Cin + 4 * qa;
    auto load_row = [&](int hi, float4 (&v)[3]) {
      const unsigned hc = (unsigned)min(max(hi, 0), H - 1);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const unsigned wc = (unsigned)min(max(w0 + s - 1, 0), W - 1);
        v[s] = *reinterpret_cast<const float4*>(xb + (hc * (unsigned)W + wc) * (unsigned)Cin);
      }
    };
    auto act_row = [&](int hi, float4 (&v)[3]) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)(w0 + s - 1) < (unsigned)W;
        v[s] = ok ? nc_act(v[s], sc, sh, affine, relu) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    float4 win[3][3], nxt[3];
    load_row(hb - 1, win[0]);
    load_row(hb, win[1]);
    load_row(hb + 1, win[2]);
    act_row(hb - 1, win[0]);
    act_row(hb, win[1]);
    act_row(hb + 1, win[2]);
    for (int h = hb; h < he; ++h) {
      load_row(h + 2, nxt);
      const long p = ((long)b * H + h) * W + min(w0, W - 1);
      float g[NOUT];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) g[o] = col_on ? dy[p * NOUT + o] : 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 a = win[t / 3][t % 3];
        const nc_v2 alo = nc_lo(a), ahi = nc_hi(a);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {          // idle lanes accumulate garbage that is never written out
          const nc_v2 gg = nc_v2{g[o], g[o]};
          const nc_v2 lo = __builtin_elementwise_fma(alo, gg, nc_lo(acc[t][o])), hi = __builtin_elementwise_fma(ahi, gg, nc_hi(acc[t][o]));
          acc[t][o] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      act_row(h + 2, nxt);
#pragma unroll
      for (int s = 0; s < 3; ++s) { win[0][s] = win[1][s]; win[1][s] = win[2][s]; win[2][s] = nxt[s]; }
    }
  }
  // combine the GPW pixel groups of the block (fixed order), lane q of group 0 writes its channel quad
  float* slab = slabs + (long)blockIdx.x * 9 * Cin * NOUT;
  float4* red4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      __syncthreads();
      red4[threadIdx.x] = acc[t][o];
      __syncthreads();
      if (grp == 0 && lane_on) {
        float4 s = red4[q];
        for (int gI = 1; gI < GPW; ++gI) {
          const float4 v = red4[gI * LP + q];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float* dst = slab + ((long)t * Cin + 4 * q) * NOUT + o;
        dst[0] = s.x; dst[NOUT] = s.y; dst[2 * NOUT] = s.z; dst[3 * NOUT] = s.w;
      }
    }
}

// out[i] = sum_z slab[z][i]; block = 32 outputs x 8 interleaved slices of z, combined in fixed order (deterministic)
__global__ __launch_bounds__(256) void narrow_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                            long n, int splits) {
  __shared__ float red[8][32];
  const int il = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + il;
  float s = 0.f;
  if (i < n)
    for (int z = sl; z < splits; z += 8) s += slabs[(long)z * n + i];
  red[sl][il] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    for (int k = 1; k < 8; ++k) s += red[k][il];
    out[i] = s;
  }
}

static int narrow_lp(int Cin) {
  const int Q = Cin / 4;
  int lp = 1;
  while (lp < Q) lp <<= 1;
  return lp;
}

static int narrow_blocks(int B, int H, int W, int lp, int cap = 4096) {
  const long groups = 256 / lp;
  long b = (long)B * icg_cdiv(W, groups) * icg_cdiv(H, NC_ROWS);          // (image, 8-column block, 32-row segment) units
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

bool icg_narrow_conv_ok(int Cin, int Cout, int R, const void* a, const void* b, const void* c, long ssb) {
  return R == 3 && ssb >= 0 && ssb < (1L << 20) && Cout >= 1 && Cout <= 4 && Cin % 4 == 0 && Cin >= 16 && Cin <= 256 && (ssb % 4 == 0) &&
         ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

template <int NOUT>
static void narrow_fprop_lp(int lp, dim3 grid, hipStream_t st, const float* x, const float* w, const float* bias,
                            const float* scale, const float* shift, long ssb, float* out, int B, int H, int W, int Cin,
                            int affine, int relu, float alpha) {
#define ICG_NF(LP_) \
  hipLaunchKernelGGL((narrow_fprop_kernel<NOUT, LP_>), grid, dim3(256), 0, st, x, w, bias, scale, shift, ssb, out, B, H, W, \
                     Cin, affine, relu, alpha)
  switch (lp) {
    case 4: ICG_NF(4); break;
    case 8: ICG_NF(8); break;
    case 16: ICG_NF(16); break;
    case 32: ICG_NF(32); break;
    default: ICG_NF(64); break;
  }
#undef ICG_NF
}


// ---- MFMA form of the narrow-OUTPUT forward (to-RGB, and the data gradient of from-RGB): NOUT <= 3, Cin = 32 NCT <= 128 --------
// Per image row and 32-pixel tile one GEMM  T^T[(tap, o)][pixel] = sum_c W[(tap, o)][c] * act(x)[pixel][c]  (M = 9 NOUT <= 27
// rows, K = Cin: Cin/2 MFMAs, the weight fragments live in registers), i.e. every pixel's contribution to the 9 output pixels
// around it; the row of T goes into a three-row LDS window and the output row above is assembled from its 3x3 neighbourhood:
// out[p][o] = alpha * sum_tap T[p + tap][(tap, o)] + bias[o].  A wavefront owns a 30-column chunk (tile = chunk + one halo
// column each side) of a 32-row segment and walks down it, so x is read once (+ 2 halo rows per segment, 2 of 32 columns).
// K order: lane half kk of k-step t = 4g + e holds channel 8g + 4kk + e, so a lane's operands of four k-steps are ONE 16-byte
// load; the weight fragments are permuted the same way.  Prologue (per-sample affine + ReLU) on the loaded quads, zero padding
// after it (halo pixels outside the image contribute zero fragments, rows outside leave a zero T row).
template <int NOUT, int NCT>
__global__ __launch_bounds__(256, (NCT <= 3 ? 3 : 2)) void narrow_fprop_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                                const float* __restrict__ bias, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, long ssb,
                                                                float* __restrict__ out, int B, int H, int W, int affine,
                                                                int relu, float alpha) {
  constexpr int Cin = 32 * NCT, NG = Cin / 8, MM = 9 * NOUT, TS = 29, CW = 30;
  __shared__ float Tw[4][3][32 * TS];
  __shared__ __attribute__((aligned(16))) float Pc[4][2][Cin];
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wv = threadIdx.x >> 6;
  float* Ts = &Tw[wv][0][0];
  // A fragments: row i = tap * NOUT + o of W, k-step (g, e) -> channel 8g + 4lh + e      (wgt: OHWI [NOUT][3][3][Cin])
  float wf[NG][4];
  {
    const int tap = li / NOUT, o = li - tap * NOUT;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) wf[g][e] = (li < MM) ? wgt[((long)o * 9 + tap) * Cin + 8 * g + 4 * lh + e] : 0.f;
  }
  const int cbs = (W + CW - 1) / CW, rss = (H + NC_ROWS - 1) / NC_ROWS;
  const long units = (long)B * cbs * rss;
  for (long u = (long)blockIdx.x * 4 + wv; u < units; u += (long)gridDim.x * 4) {
    const int b = (int)(u / ((long)cbs * rss));
    const int r2 = (int)(u - (long)b * cbs * rss);
    const int hb = (r2 / cbs) * NC_ROWS, c0 = (r2 % cbs) * CW;
    const int he = min(H, hb + NC_ROWS);
    if (affine) {                                            // this image's prologue coefficients -> LDS (wave-private)
      for (int c = lane; c < Cin; c += 64) { Pc[wv][0][c] = scale[(long)b * ssb + c]; Pc[wv][1][c] = shift[(long)b * ssb + c]; }
      icg_wave_lds_sync();                                   // other lanes of this wave read what this lane stored
    }
    const int col = c0 - 1 + li;                             // this lane's pixel column as the B-operand column
    const bool col_on = (unsigned)col < (unsigned)W;
    for (int h = hb - 1; h <= he; ++h) {                     // T rows hb-1 .. he; output row h-1 once rows h-2 .. h are in the window
      float* Trow = Ts + ((h + 3) % 3) * (32 * TS);
      if ((unsigned)h < (unsigned)H) {
        const float* xp = x + (((long)b * H + h) * W + (col_on ? col : 0)) * Cin + 4 * lh;
        float4 v[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) v[g] = *reinterpret_cast<const float4*>(xp + 8 * g);
        th_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          float4 a = v[g];
          if (affine) {
            const float4 sc = *reinterpret_cast<const float4*>(&Pc[wv][0][8 * g + 4 * lh]);
            const float4 sh = *reinterpret_cast<const float4*>(&Pc[wv][1][8 * g + 4 * lh]);
            a.x = fmaf(a.x, sc.x, sh.x); a.y = fmaf(a.y, sc.y, sh.y); a.z = fmaf(a.z, sc.z, sh.z); a.w = fmaf(a.w, sc.w, sh.w);
          }
          if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
          if (!col_on) a = make_float4(0.f, 0.f, 0.f, 0.f);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[g][0], a.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[g][1], a.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[g][2], a.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[g][3], a.w, acc, 0, 0, 0);
        }
        // C layout: column = lane & 31 = pixel of the tile, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = (tap, o)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (i < MM) Trow[li * TS + i] = acc[r];
        }
      } else {
        for (int k = lane; k < 32 * TS; k += 64) Trow[k] = 0.f;
      }
      // wave-private LDS exchange (T rows written by one set of lanes, read by others): no workgroup barrier is needed, but
      // the store -> load order across lanes is stated to the compiler (release fence + wave barrier; both free at run time)
      icg_wave_lds_sync();
      const int ho = h - 1;
      if (ho >= hb && ho < he) {
#pragma unroll
        for (int it = 0; it < (CW * NOUT + 63) / 64; ++it) {
          const int item = lane + 64 * it;
          if (item < CW * NOUT) {
            const int jo = 1 + item / NOUT, o = item - (item / NOUT) * NOUT;
            const int oc = c0 - 1 + jo;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              const float* Tr = Ts + ((ho + r - 1 + 3) % 3) * (32 * TS);
#pragma unroll
              for (int sx = 0; sx < 3; ++sx) sum += Tr[(jo + sx - 1) * TS + (r * 3 + sx) * NOUT + o];
            }
            if (oc < W) out[(((long)b * H + ho) * W + oc) * NOUT + o] = alpha * sum + (bias ? bias[o] : 0.f);
          }
        }
      }
      icg_wave_lds_sync();                                   // the next row's T stores reuse the window slot read above
    }
  }
}

static bool narrow_fprop_mfma_ok(int Cin, int Cout) { return Cout >= 1 && Cout <= 3 && Cin % 32 == 0 && Cin <= 128; }

template <int NOUT>
static void narrow_fprop_mfma_launch(int nct, dim3 grid, hipStream_t st, const float* x, const float* w, const float* bias,
                                     const float* scale, const float* shift, long ssb, float* out, int B, int H, int W, int affine,
                                     int relu, float alpha) {
#define ICG_NFM(NCT_)                                                                                                          \
  hipLaunchKernelGGL((narrow_fprop_mfma_kernel<NOUT, NCT_>), grid, dim3(256), 0, st, x, w, bias, scale, shift, ssb, out, B, H, W, \
                     affine, relu, alpha)
  switch (nct) {
    case 1: ICG_NFM(1); break;
    case 2: ICG_NFM(2); break;
    case 3: ICG_NFM(3); break;
    default: ICG_NFM(4); break;
  }
#undef ICG_NFM
}

int icg_narrow_fprop(const float* x, const float* w, const float* bias, const float* scale, const float* shift, long ssb,
                     float* out, int B, int H, int W, int Cin, int Cout, int affine, int relu, float alpha,
                     hipStream_t st) {
  if (narrow_fprop_mfma_ok(Cin, Cout)) {
    const long units = (long)B * icg_cdiv(W, 30) * icg_cdiv(H, NC_ROWS);
    long blocks = icg_cdiv(units, 4);
    if (blocks > 4096) blocks = 4096;
    const dim3 g((unsigned)blocks);
    if (Cout == 1) narrow_fprop_mfma_launch<1>(Cin / 32, g, st, x, w, bias, scale, shift, ssb, out, B, H, W, affine, relu, alpha);
    else if (Cout == 2) narrow_fprop_mfma_launch<2>(Cin / 32, g, st, x, w, bias, scale, shift, ssb, out, B, H, W, affine, relu, alpha);
    else narrow_fprop_mfma_launch<3>(Cin / 32, g, st, x, w, bias, scale, shift, ssb, out, B, H, W, affine, relu, alpha);
    return icg_check_launch();
  }
  const int lp = narrow_lp(Cin);
  const dim3 grid((unsigned)narrow_blocks(B, H, W, lp, 1 << 20));
  switch (Cout) {
    case 1: narrow_fprop_lp<1>(lp, grid, st, x, w, bias, scale, shift, ssb, out, B, H, W, Cin, affine, relu, alpha); break;
    case 2: narrow_fprop_lp<2>(lp, grid, st, x, w, bias, scale, shift, ssb, out, B, H, W, Cin, affine, relu, alpha); break;
    case 3: narrow_fprop_lp<3>(lp, grid, st, x, w, bias, scale, shift, ssb, out, B, H, W, Cin, affine, relu, alpha); break;
    default: narrow_fprop_lp<4>(lp, grid, st, x, w, bias, scale, shift, ssb, out, B, H, W, Cin, affine, relu, alpha); break;
  }
  return icg_check_launch();
}

// ---- MFMA form of the thin-input kernels (9 * CIN <= 32 and Cout = 32 * NT, NT <= 4: the from-RGB layer of every D) ----------
// With K = 9 * CIN <= 32 the convolution is out[P][Cout] = Xcol[P][32] W^T[32][Cout] per 32-pixel tile: 16 k-steps of
// v_mfma_f32_32x32x2_f32 per column tile instead of 27 packed FMAs per output quad, so both kernels become what they should be --
// one HBM pass over the Cout-wide tensor (the 3-channel image stays in L1/L2).  Operand fragments come straight from global
// memory: A[i = lane & 31][k = 2t + (lane >> 5)] is one image value per lane and k-step (zero for padding and for k >= 9 CIN),
// B is the weight fragment held in registers (fprop) or one dy value per lane, k-step and column tile (wgrad: 128-byte rows).

template <int CIN, int NT>
__global__ __launch_bounds__(256) void thin_fprop_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                              const float* __restrict__ bias, float* __restrict__ out, int B,
                                                              int H, int W, float alpha, int R) {
  constexpr int Cout = 32 * NT;
  const int KK = R * R * CIN, PD = R >> 1;                 // R = 3, or 1 (the 1x1 shortcut of D's first block: K = CIN)
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  // weight fragments: B[k][j] = wgt[32 jt + j][k] (OHWI: [Cout][3][3][CIN] = [Cout][KK])
  float wf[16][NT];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = 2 * t + lh;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) wf[t][jt] = (k < KK) ? wgt[(long)(32 * jt + li) * KK + k] : 0.f;
  }
  float bv[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) bv[jt] = bias ? bias[32 * jt + li] : 0.f;
  const long P = (long)B * H * W;
  const long ntile = (P + 31) / 32;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long)gridDim.x * 4;
  auto gather = [&](long tile, float (&af)[16]) {          // the A fragments of a 32-pixel tile
    const long pa = min(tile * 32 + li, P - 1);            // this lane's pixel as the A-operand row (clamped: masked at the store)
    const int w = (int)(pa % W);
    const long t2 = pa / W;
    const int h = (int)(t2 % H);
    const float* xb = x + (t2 - h) * (long)W * CIN;          // image base
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k = 2 * t + lh;
      const int tap = k / CIN, c = k - tap * CIN;
      const int r = tap / R, sx = tap - R * r;
      const int hi = h + r - PD, wi = w + sx - PD;
      const bool ok = (k < KK) && ((unsigned)hi < (unsigned)H) && ((unsigned)wi < (unsigned)W);
      const float v = xb[((long)(ok ? hi : h) * W + (ok ? wi : w)) * CIN + (k < KK ? c : 0)];
      af[t] = ok ? v : 0.f;
    }
  };
  float af[16], afn[16];
  if (wave < ntile) gather(wave, af);
  for (long tile = wave; tile < ntile; tile += nwave) {
    gather(min(tile + nwave, ntile - 1), afn);             // next tile's operands in flight under this tile's MFMAs and stores
    th_f32x16 acc[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], wf[t][jt], acc[jt], 0, 0, 0);
    // C layout: column = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the tile)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long pp = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (pp < P) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) out[pp * Cout + 32 * jt + li] = alpha * acc[jt][r] + bv[jt];
      }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) af[t] = afn[t];
  }
}

// slab[block][k = tap * CIN + c][co] = sum over the block's image rows of x[pix + tap][c] * dy[pix][co]    (HWIO rows k < 9 CIN)
// NARROW = 0: x is the CIN-channel image, dy the 32 NT-channel tensor (from-RGB weight gradient), rows k = (tap, c).
// NARROW = 1: the roles are swapped (to-RGB weight gradient, CIN = its 1..3 OUTPUT channels): `x` is dy [.., CIN], `dy` is the
// layer's 32 NT-channel input with the per-sample affine + ReLU prologue applied to the fragment, the tap shift changes sign
// (dw[tap][ci][co] = sum_q act(x)[q][ci] * dy[q - tap][co]) and the slab is written as [tap][ci][co].
template <int CIN, int NT, int NARROW>
__global__ __launch_bounds__(256) void thin_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ slabs, int B, int H, int W,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              long ssb, int affine, int relu, int R) {
  constexpr int Cout = 32 * NT, U = 8;                      // U k-steps (2 pixels each) of loads in flight ahead of their MFMAs
  const int KK = R * R * CIN, PD = R >> 1;
  __shared__ float red[32 * Cout];
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wv = threadIdx.x >> 6;
  // A[i][k]: row i = (tap, c) of the weight gradient, k = pixel
  const int tap = li / CIN, c = li - tap * CIN;
  const int sg = NARROW ? -1 : 1;
  const int dr = sg * (tap / R - PD), ds = sg * (tap - R * (tap / R) - PD);
  const bool row_on = li < KK;
  th_f32x16 acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
  const long rows = (long)B * H;
  const int wsteps = (W + 1) / 2;                            // k-steps per image row
  for (long row = (long)blockIdx.x * 4 + wv; row < rows; row += (long)gridDim.x * 4) {
    const int h = (int)(row % H);
    const int hi = h + dr;
    const bool hok = row_on && ((unsigned)hi < (unsigned)H);
    const float* xr = x + (row + (hok ? dr : 0)) * (long)W * CIN + c;     // row h + dr of the same image (clamped when masked)
    const float* dr_ = dy + row * (long)W * Cout + li;
    float psc[NT], psh[NT];                                  // NARROW: prologue coefficients of this lane's channels in this image
    if (NARROW && affine) {
      const long bimg = row / H;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) { psc[jt] = scale[bimg * ssb + 32 * jt + li]; psh[jt] = shift[bimg * ssb + 32 * jt + li]; }
    }
    float av[U], bfr[U][NT];
    auto fetch = [&](int st, int u) {                        // operands of k-step st into slot u
      const int wq = 2 * st + lh;                            // this lane's pixel column
      const int wi = wq + ds;
      const bool aok = hok && (wq < W) && ((unsigned)wi < (unsigned)W);
      const float v = xr[(long)(aok ? wi : 0) * CIN];
      av[u] = aok ? v : 0.f;
      const bool bok = wq < W;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        float g = dr_[(long)(bok ? wq : 0) * Cout + 32 * jt];
        if (NARROW) {
          if (affine) g = fmaf(g, psc[jt], psh[jt]);
          if (relu) g = fmaxf(g, 0.f);
        }
        bfr[u][jt] = bok ? g : 0.f;
      }
    };
    for (int s0 = 0; s0 < wsteps; s0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < wsteps) fetch(s0 + u, u);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < wsteps) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bfr[u][jt], acc[jt], 0, 0, 0);
        }
    }
  }
  // combine the block's four wavefronts in fixed order, write the rows k < KK of the slab
  float* slab = slabs + (long)blockIdx.x * KK * Cout;
  for (int src = 1; src < 4; ++src) {
    __syncthreads();
    if (wv == src) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * lh) * Cout + 32 * jt + li] = acc[jt][r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] += red[((r & 3) + 8 * (r >> 2) + 4 * lh) * Cout + 32 * jt + li];
    }
  }
  if (wv == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (k < KK) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          if (NARROW) slab[((long)(k / CIN) * Cout + 32 * jt + li) * CIN + (k % CIN)] = acc[jt][r];
          else slab[(long)k * Cout + 32 * jt + li] = acc[jt][r];
        }
      }
    }
  }
}

static bool thin_mfma_ok(int Cin, int Cout) { return 9 * Cin <= 32 && Cout % 32 == 0 && Cout <= 128; }     // (R = 1: K = Cin, also fits)

template <int CIN>
static void thin_mfma_launch(bool wgrad, int nt, dim3 grid, hipStream_t st, const float* x, const float* w_or_dy, const float* bias,
                             float* out, int B, int H, int W, float alpha, int R) {
#define ICG_THM(NT_)                                                                                                            \
  if (wgrad) hipLaunchKernelGGL((thin_wgrad_mfma_kernel<CIN, NT_, 0>), grid, dim3(256), 0, st, x, w_or_dy, out, B, H, W,         \
                                (const float*)nullptr, (const float*)nullptr, 0L, 0, 0, R);                                     \
  else hipLaunchKernelGGL((thin_fprop_mfma_kernel<CIN, NT_>), grid, dim3(256), 0, st, x, w_or_dy, bias, out, B, H, W, alpha, R)
  switch (nt) {
    case 1: ICG_THM(1); break;
    case 2: ICG_THM(2); break;
    case 3: ICG_THM(3); break;
    default: ICG_THM(4); break;
  }
#undef ICG_THM
}

static void thin_mfma_dispatch(bool wgrad, int Cin, int nt, dim3 grid, hipStream_t st, const float* x, const float* w_or_dy,
                               const float* bias, float* out, int B, int H, int W, float alpha, int R) {
  switch (Cin) {
    case 1: thin_mfma_launch<1>(wgrad, nt, grid, st, x, w_or_dy, bias, out, B, H, W, alpha, R); break;
    case 2: thin_mfma_launch<2>(wgrad, nt, grid, st, x, w_or_dy, bias, out, B, H, W, alpha, R); break;
    default: thin_mfma_launch<3>(wgrad, nt, grid, st, x, w_or_dy, bias, out, B, H, W, alpha, R); break;
  }
}


// to-RGB weight gradient on the same kernel (NARROW = 1): Cout in 1..3 output channels, Cin = 32 NT <= 128 input channels
static bool narrow_wgrad_mfma_ok(int Cin, int Cout) { return Cout >= 1 && Cout <= 3 && Cin % 32 == 0 && Cin <= 128; }

template <int NOUT>
static void narrow_wgrad_mfma_launch(int nt, dim3 grid, hipStream_t st, const float* x, const float* dy, float* slabs, int B, int H,
                                     int W, const float* scale, const float* shift, long ssb, int affine, int relu) {
#define ICG_NWM(NT_) \
  hipLaunchKernelGGL((thin_wgrad_mfma_kernel<NOUT, NT_, 1>), grid, dim3(256), 0, st, dy, x, slabs, B, H, W, scale, shift, ssb, affine, relu, 3)
  switch (nt) {
    case 1: ICG_NWM(1); break;
    case 2: ICG_NWM(2); break;
    case 3: ICG_NWM(3); break;
    default: ICG_NWM(4); break;
  }
#undef ICG_NWM
}

size_t icg_narrow_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const int lp = narrow_lp(Cin);
  return (size_t)narrow_blocks(B, H, W, lp, 768) * 9 * Cin * Cout * sizeof(float);
}

template <int NOUT>
static void narrow_wgrad_lp(int lp, dim3 grid, hipStream_t st, const float* x, const float* dy, const float* scale,
                            const float* shift, long ssb, float* slabs, int B, int H, int W, int Cin, int affine,
                            int relu) {
#define ICG_NW(LP_) \
  hipLaunchKernelGGL((narrow_wgrad_kernel<NOUT, LP_>), grid, dim3(256), 0, st, x, dy, scale, shift, ssb, slabs, B, H, W, Cin, \
                     affine, relu)
  switch (lp) {
    case 4: ICG_NW(4); break;
    case 8: ICG_NW(8); break;
    case 16: ICG_NW(16); break;
    case 32: ICG_NW(32); break;
    default: ICG_NW(64); break;
  }
#undef ICG_NW
}

int icg_narrow_wgrad(const float* x, const float* dy, const float* scale, const float* shift, long ssb, float* dw,
                     void* workspace, int B, int H, int W, int Cin, int Cout, int affine, int relu, hipStream_t st) {
  const int lp = narrow_lp(Cin);
  int blocks = narrow_blocks(B, H, W, lp, 768);
  float* slabs = (float*)workspace;
  if (narrow_wgrad_mfma_ok(Cin, Cout)) {
    const long rows4 = ((long)B * H + 3) / 4;                 // one image row per wavefront and pass
    if (blocks > rows4) blocks = (int)rows4;
    const dim3 g((unsigned)blocks);
    if (Cout == 1) narrow_wgrad_mfma_launch<1>(Cin / 32, g, st, x, dy, slabs, B, H, W, scale, shift, ssb, affine, relu);
    else if (Cout == 2) narrow_wgrad_mfma_launch<2>(Cin / 32, g, st, x, dy, slabs, B, H, W, scale, shift, ssb, affine, relu);
    else narrow_wgrad_mfma_launch<3>(Cin / 32, g, st, x, dy, slabs, B, H, W, scale, shift, ssb, affine, relu);
    const long n = 9L * Cin * Cout;
    hipLaunchKernelGGL(narrow_reduce_kernel, dim3((unsigned)icg_cdiv(n, 32)), dim3(256), 0, st, (const float*)slabs, dw, n, blocks);
    return icg_check_launch();
  }
  const dim3 grid((unsigned)blocks);
  switch (Cout) {
    case 1: narrow_wgrad_lp<1>(lp, grid, st, x, dy, scale, shift, ssb, slabs, B, H, W, Cin, affine, relu); break;
    case 2: narrow_wgrad_lp<2>(lp, grid, st, x, dy, scale, shift, ssb, slabs, B, H, W, Cin, affine, relu); break;
    case 3: narrow_wgrad_lp<3>(lp, grid, st, x, dy, scale, shift, ssb, slabs, B, H, W, Cin, affine, relu); break;
    default: narrow_wgrad_lp<4>(lp, grid, st, x, dy, scale, shift, ssb, slabs, B, H, W, Cin, affine, relu); break;
  }
  const long n = 9L * Cin * Cout;
  hipLaunchKernelGGL(narrow_reduce_kernel, dim3((unsigned)icg_cdiv(n, 32)), dim3(256), 0, st, (const float*)slabs, dw, n,
                     blocks);
  return icg_check_launch();
}

// ---- skinny linear layers: out[M][N] = x[M][K] w[N][K]^T with M = batch rows (<= 256) and K not a multiple of 4 -----------------
// (the conditional-BN gain / bias projections of the generator, layers.py:367-374: K = 657 = 17 z + 128 class + 512 feature
// inputs, N = the block's channel count, M = 64.  On the implicit-GEMM kernel these are one row of tiles walking K with the
// scalar gather loader: 0.1-0.3 ms each, 60 per step.)  One block per 8 output columns, lane = row, the K range split over the
// block's 4 wavefronts (fixed-order LDS combine): x is read with per-lane row streams, w through wave-uniform loads; the weight gradient dw[K][N] = x^T dy is one thread per
// (k, 4 columns) looping over the M rows.
template <int NB>
__global__ __launch_bounds__(256) void skinny_fprop_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int M, int N,
                                                           int K, float alpha) {
  __shared__ float red[3][64][NB];                 // partial sums of waves 1..3 (the K range is split over the 4 waves)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.y * 64 + lane;
  const int n0 = blockIdx.x * NB;
  const int kq = ((K + 3) / 4 + 3) & ~3;           // K range of a wave, a multiple of 4
  const int kb = min(wv * kq, K), ke = min(kb + kq, K);
  const float* xr = x + (long)min(row, M - 1) * K;
  const float* wr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) wr[j] = w + (long)min(n0 + j, N - 1) * K;
  float acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = 0.f;
  int k = kb;
  for (; k + 4 <= ke; k += 4) {
    const float x0 = xr[k], x1 = xr[k + 1], x2 = xr[k + 2], x3 = xr[k + 3];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      acc[j] = fmaf(x0, wr[j][k], acc[j]);
      acc[j] = fmaf(x1, wr[j][k + 1], acc[j]);
      acc[j] = fmaf(x2, wr[j][k + 2], acc[j]);
      acc[j] = fmaf(x3, wr[j][k + 3], acc[j]);
    }
  }
  for (; k < ke; ++k) {
    const float x0 = xr[k];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = fmaf(x0, wr[j][k], acc[j]);
  }
  if (wv > 0) {
#pragma unroll
    for (int j = 0; j < NB; ++j) red[wv - 1][lane][j] = acc[j];
  }
  __syncthreads();
  if (wv == 0 && row < M) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float s = ((acc[j] + red[0][lane][j]) + red[1][lane][j]) + red[2][lane][j];      // fixed order
      if (n0 + j < N) out[(long)row * N + n0 + j] = alpha * s + (bias ? bias[n0 + j] : 0.f);
    }
  }
}

// dw[k][n..n+3] = sum_b x[b][k] dy[b][n..n+3]   (HWIO with R = 1: [K][N])
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, int M, int N4, int K) {
  const long total = (long)K * N4;
  const long gstride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const int n4 = (int)(i % N4);
    const int k = (int)(i / N4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* g = reinterpret_cast<const float4*>(dy) + n4;
    for (int b = 0; b < M; ++b) {
      const float xv = x[(long)b * K + k];
      const float4 d = g[(long)b * N4];
      acc.x = fmaf(xv, d.x, acc.x); acc.y = fmaf(xv, d.y, acc.y); acc.z = fmaf(xv, d.z, acc.z); acc.w = fmaf(xv, d.w, acc.w);
    }
    reinterpret_cast<float4*>(dw)[i] = acc;
  }
}

bool icg_skinny_ok(long M, int Cin, int R) { return R == 1 && M <= 256 && (Cin % 4) != 0 && Cin >= 16; }

int icg_skinny_fprop(const float* x, const float* w, const float* bias, float* out, int M, int N, int K, float alpha,
                     hipStream_t st) {
  hipLaunchKernelGGL((skinny_fprop_kernel<8>), dim3((unsigned)icg_cdiv(N, 8), (unsigned)icg_cdiv(M, 64)), dim3(256), 0, st, x, w,
                     bias, out, M, N, K, alpha);
  return icg_check_launch();
}

int icg_skinny_wgrad(const float* x, const float* dy, float* dw, int M, int N, int K, hipStream_t st) {
  const long total = (long)K * (N / 4);
  long nb = icg_cdiv(total, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(skinny_wgrad_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, dy, dw, M, N / 4, K);
  return icg_check_launch();
}

// ---- 3x3 convolutions with <= 4 INPUT channels (the discriminator's from-RGB layer, BigGAN.py:480-492 / layers.py:587-600 with
// preactivation = False) and their weight gradient.  K = 27 is too short for the implicit GEMM (scalar gather loader, 16-wide
// K tiles): lanes own output-channel quads, a group of LP lanes walks down an image column with the 3 x 3 x Cin window in
// registers (9*Cin dwords per lane, 3*Cin new ones per step, the same addresses for all lanes of a group: one L1 request);
// HBM-bound on the Cout-wide tensor.  No prologue (the layer reads the raw image).
template <int CIN, int LP>
__global__ __launch_bounds__(256) void thin_fprop_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                         const float* __restrict__ bias, float* __restrict__ out, int B, int H,
                                                         int W, int Cout, float alpha) {
  constexpr int GPW = 256 / LP;
  const int q = threadIdx.x % LP, grp = threadIdx.x / LP;
  const int Q = Cout >> 2;
  const bool lane_on = q < Q;
  const int qa = lane_on ? q : Q - 1;
  nc_v2 wlo[9][CIN], whi[9][CIN];                    // w[co..co+3][tap][ci]  (OHWI: [Cout][3][3][CIN])
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float* wp = wgt + ((long)(4 * qa) * 9 + t) * CIN + c;
      wlo[t][c] = nc_v2{wp[0], wp[9 * CIN]};
      whi[t][c] = nc_v2{wp[18 * CIN], wp[27 * CIN]};
    }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + 4 * qa);
  const unsigned cbs = (unsigned)(W + GPW - 1) / GPW, rss = (unsigned)(H + NC_ROWS - 1) / NC_ROWS;
  const unsigned units = (unsigned)B * cbs * rss;
  for (unsigned u = blockIdx.x; u < units; u += gridDim.x) {
    const unsigned b = u / (cbs * rss), r2 = u - b * cbs * rss;
    const int hb = (int)(r2 / cbs) * NC_ROWS, w0 = (int)(r2 % cbs) * GPW + grp;
    const int he = min(H, hb + NC_ROWS);
    const bool col_on = w0 < W;
    const float* xb = x + (long)b * H * W * CIN;
    auto load_row = [&](int hi, float (&v)[3][CIN]) {
      const int hc = min(max(hi, 0), H - 1);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = w0 + s - 1;
        const int wc = min(max(wi, 0), W - 1);
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float t = xb[((long)hc * W + wc) * CIN + c];
          v[s][c] = ok ? t : 0.f;
        }
      }
    };
    float win[3][3][CIN], nxt[3][CIN];
    load_row(hb - 1, win[0]);
    load_row(hb, win[1]);
    load_row(hb + 1, win[2]);
    for (int h = hb; h < he; ++h) {
      load_row(h + 2, nxt);
      nc_v2 lo = nc_v2{0.f, 0.f}, hi = nc_v2{0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float a = win[t / 3][t % 3][c];
          const nc_v2 aa = nc_v2{a, a};
          lo = __builtin_elementwise_fma(aa, wlo[t][c], lo);
          hi = __builtin_elementwise_fma(aa, whi[t][c], hi);
        }
      if (lane_on && col_on) {
        const long p = ((long)b * H + h) * W + w0;
        *reinterpret_cast<float4*>(out + p * Cout + 4 * q) =
            make_float4(alpha * lo.x + bv.x, alpha * lo.y + bv.y, alpha * hi.x + bv.z, alpha * hi.y + bv.w);
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < CIN; ++c) { win[0][s][c] = win[1][s][c]; win[1][s][c] = win[2][s][c]; win[2][s][c] = nxt[s][c]; }
    }
  }
}

// slab[block][tap][ci][co] = sum over the block's pixels of x[pix + tap][ci] * dy[pix][co]      (HWIO)
template <int CIN, int LP>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ slabs, int B, int H, int W, int Cout) {
  constexpr int GPW = 256 / LP;
  __shared__ float red[256 * 4];
  const int q = threadIdx.x % LP, grp = threadIdx.x / LP;
  const int Q = Cout >> 2;
  const bool lane_on = q < Q;
  const int qa = lane_on ? q : Q - 1;
  nc_v2 alo[9][CIN], ahi[9][CIN];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c) { alo[t][c] = nc_v2{0.f, 0.f}; ahi[t][c] = nc_v2{0.f, 0.f}; }
  const unsigned cbs = (unsigned)(W + GPW - 1) / GPW, rss = (unsigned)(H + NC_ROWS - 1) / NC_ROWS;
  const unsigned units = (unsigned)B * cbs * rss;
  for (unsigned u = blockIdx.x; u < units; u += gridDim.x) {
    const unsigned b = u / (cbs * rss), r2 = u - b * cbs * rss;
    const int hb = (int)(r2 / cbs) * NC_ROWS, w0 = (int)(r2 % cbs) * GPW + grp;
    const int he = min(H, hb + NC_ROWS);
    const bool col_on = w0 < W;
    const float* xb = x + (long)b * H * W * CIN;
    auto load_row = [&](int hi, float (&v)[3][CIN]) {
      const int hc = min(max(hi, 0), H - 1);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wi = w0 + s - 1;
        const int wc = min(max(wi, 0), W - 1);
        const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float t = xb[((long)hc * W + wc) * CIN + c];
          v[s][c] = ok ? t : 0.f;
        }
      }
    };
    float win[3][3][CIN], nxt[3][CIN];
    load_row(hb - 1, win[0]);
    load_row(hb, win[1]);
    load_row(hb + 1, win[2]);
    for (int h = hb; h < he; ++h) {
      load_row(h + 2, nxt);
      const long p = ((long)b * H + h) * W + min(w0, W - 1);
      float4 g = *reinterpret_cast<const float4*>(dy + p * Cout + 4 * qa);
      if (!col_on) g = make_float4(0.f, 0.f, 0.f, 0.f);
      const nc_v2 glo = nc_v2{g.x, g.y}, ghi = nc_v2{g.z, g.w};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float a = win[t / 3][t % 3][c];
          const nc_v2 aa = nc_v2{a, a};
          alo[t][c] = __builtin_elementwise_fma(aa, glo, alo[t][c]);
          ahi[t][c] = __builtin_elementwise_fma(aa, ghi, ahi[t][c]);
        }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < CIN; ++c) { win[0][s][c] = win[1][s][c]; win[1][s][c] = win[2][s][c]; win[2][s][c] = nxt[s][c]; }
    }
  }
  // combine the GPW pixel groups of the block (fixed order); lane q of group 0 writes its output-channel quad
  float* slab = slabs + (long)blockIdx.x * 9 * CIN * Cout;
  float4* red4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      __syncthreads();
      red4[threadIdx.x] = make_float4(alo[t][c].x, alo[t][c].y, ahi[t][c].x, ahi[t][c].y);
      __syncthreads();
      if (grp == 0 && lane_on) {
        float4 s = red4[q];
        for (int gI = 1; gI < GPW; ++gI) {
          const float4 v = red4[gI * LP + q];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(slab + ((long)t * CIN + c) * Cout + 4 * q) = s;
      }
    }
}


bool icg_thin_conv_ok(int Cin, int Cout, int R) {
  if (R == 1) return Cin >= 1 && Cin <= 3 && thin_mfma_ok(Cin, Cout);       // 1x1: MFMA form only
  return R == 3 && Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && Cout >= 16 && Cout <= 256;
}

static int thin_lp(int Cout) {
  int lp = 4;
  while (lp < Cout / 4) lp <<= 1;
  return lp;
}

size_t icg_thin_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  return (size_t)narrow_blocks(B, H, W, thin_lp(Cout), 768) * 9 * Cin * Cout * sizeof(float);
}

template <int CIN>
static void thin_launch(bool wgrad, int lp, dim3 grid, hipStream_t st, const float* x, const float* w_or_dy, const float* bias,
                        float* out, int B, int H, int W, int Cout, float alpha) {
#define ICG_TH(LP_)                                                                                                              \
  if (wgrad) hipLaunchKernelGGL((thin_wgrad_kernel<CIN, LP_>), grid, dim3(256), 0, st, x, w_or_dy, out, B, H, W, Cout);            \
  else hipLaunchKernelGGL((thin_fprop_kernel<CIN, LP_>), grid, dim3(256), 0, st, x, w_or_dy, bias, out, B, H, W, Cout, alpha)
  switch (lp) {
    case 4: ICG_TH(4); break;
    case 8: ICG_TH(8); break;
    case 16: ICG_TH(16); break;
    case 32: ICG_TH(32); break;
    default: ICG_TH(64); break;
  }
#undef ICG_TH
}

static void thin_dispatch(bool wgrad, int Cin, int lp, dim3 grid, hipStream_t st, const float* x, const float* w_or_dy,
                          const float* bias, float* out, int B, int H, int W, int Cout, float alpha) {
  switch (Cin) {
    case 1: thin_launch<1>(wgrad, lp, grid, st, x, w_or_dy, bias, out, B, H, W, Cout, alpha); break;
    case 2: thin_launch<2>(wgrad, lp, grid, st, x, w_or_dy, bias, out, B, H, W, Cout, alpha); break;
    case 3: thin_launch<3>(wgrad, lp, grid, st, x, w_or_dy, bias, out, B, H, W, Cout, alpha); break;
    default: thin_launch<4>(wgrad, lp, grid, st, x, w_or_dy, bias, out, B, H, W, Cout, alpha); break;
  }
}

int icg_thin_fprop(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int Cin, int Cout, int R,
                   float alpha, hipStream_t st) {
  if (thin_mfma_ok(Cin, Cout) && Cin <= 3) {
    const long tiles = ((long)B * H * W + 31) / 32;
    long blocks = (tiles + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    thin_mfma_dispatch(false, Cin, Cout / 32, dim3((unsigned)blocks), st, x, w, bias, out, B, H, W, alpha, R);
    return icg_check_launch();
  }
  const int lp = thin_lp(Cout);
  thin_dispatch(false, Cin, lp, dim3((unsigned)narrow_blocks(B, H, W, lp, 1 << 20)), st, x, w, bias, out, B, H, W, Cout, alpha);
  return icg_check_launch();
}

int icg_thin_wgrad(const float* x, const float* dy, float* dw, void* workspace, int B, int H, int W, int Cin, int Cout, int R,
                   hipStream_t st) {
  const int lp = thin_lp(Cout);
  int blocks = narrow_blocks(B, H, W, lp, 768);
  float* slabs = (float*)workspace;
  if (thin_mfma_ok(Cin, Cout) && Cin <= 3) {
    const long rows4 = ((long)B * H + 3) / 4;                 // one image row per wavefront and pass
    if (blocks > rows4) blocks = (int)rows4;
    thin_mfma_dispatch(true, Cin, Cout / 32, dim3((unsigned)blocks), st, x, dy, nullptr, slabs, B, H, W, 1.f, R);
  } else
  thin_dispatch(true, Cin, lp, dim3((unsigned)blocks), st, x, dy, nullptr, slabs, B, H, W, Cout, 1.f);
  const long n = (long)R * R * Cin * Cout;
  hipLaunchKernelGGL(narrow_reduce_kernel, dim3((unsigned)icg_cdiv(n, 32)), dim3(256), 0, st, (const float*)slabs, dw, n, blocks);
  return icg_check_launch();
}
