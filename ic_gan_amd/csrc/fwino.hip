// Fused Winograd F(4x4, 3x3) convolution for the NARROW 3x3 layers (Cin <= 192 per pass, Cout a multiple of 96): input
// transform, the 36 (25) plane GEMMs, output transform and epilogue in ONE kernel -- the V and M planes of the three-kernel
// composite of winograd.hip never touch HBM.  Reference: the 96- / 192-channel 3x3 convolutions of GBlock / DBlock
// (BigGAN_PyTorch/layers.py:144-153 SNConv2d.forward, 542-552 GBlock.forward, 587-613 DBlock.forward) and their data gradients.
//
// Why: at K = Cin = 96 a plane GEMM has 24 FLOP per byte of V + M traffic -- the HBM ridge -- so the composite moves
// 2.25 x (in + out) x 2 bytes through HBM around an MFMA phase that cannot hide them (0.34 - 0.51 of the fp32
// MFMA peak, 101 GB per step in icg_pgemm_nn_stream_kernel<3,2,3> alone).  Here a workgroup owns a 4 x 4 block of output
// tiles (16 x 16 pixels) x 96 output channels and keeps ALL planes' accumulators in registers:
//     36 planes x 16 tiles x 96 columns = 55 296 fp32 = 108 registers per lane over 8 MFMA waves
// (the register file, not the LDS, is what bounds the tile count: 37 tiles would fill the CU's 512 KB).
//
// Warp specialisation (12 waves = 768 threads, 3 per SIMD):
//   * waves 8..11, PRODUCERS: lane = (tile, channel PAIR) of a 32-channel chunk; the 6 x 6 (4 x 4 source pixels, upsample-fused
//     form) window lives in registers as float2: BN / ccbn affine + ReLU + zero padding, B^T d B with v_pk_* arithmetic, the 36
//     values go to LDS in the MFMA operand order (ds_write_b64, conflict-free through an XOR of the tile index with the channel
//     quad); optionally also to HBM as the V planes the weight gradient wants (icg_conv2d_wino4_wgrad_from_v).  The window is
//     REFILLED IN PLACE: the second 1-D pass retires it line by line and each retired line's registers are loaded at once with
//     the matching line of the next chunk's window (buffer loads through one descriptor over x), so the loads have half a
//     transform of distance without a second set of 72 registers; the two pass orders alternate (see transform notes below).
//     Their loads are HBM-latency loads; keeping them in waves of their own keeps the in-order vmcnt queue of the MFMA waves
//     free of them.
//   * waves 0..7, CONSUMERS: wave (pg, ng) owns planes pg, pg+4, ... and 48 of the 96 columns: per plane and 16-channel
//     group ONE ds_read_b128 (the tile operand of four k-steps) and three 16-byte global loads of the fragment-major weights
//     (icg_fwino_pack_kernel: every wave-load is 1 KiB contiguous = 8 full cache lines from L2) feed 12
//     v_mfma_f32_16x16x4_f32.  Weight fragments are prefetched one half-step (12 MFMAs) ahead.
//   * double-buffered V chunk in LDS (2 x 72 KiB), ONE raw s_barrier per 32-channel chunk.
//   * output: accumulators -> LDS (the V buffers are dead by then) in two 48-column rounds, every thread then owns one
//     (tile, column): bias and the residual operand first (all loads in flight), A^T m A from 36 LDS reads, alpha / bias /
//     residual (plain, upsample-on-read, ReLU mask) and the stores.
// Chains are single-level over K <= 192 (48 MFMA steps); the routes through this kernel are gated on that (fwino_applies).
#include "icg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// measurement builds only (tools/build_dbg.sh FWA<n>): 1 producers idle, 2 no MFMAs, 4 no weight loads in the loop, 8 no output stage
#ifndef FWINO_ABLATE
#define FWINO_ABLATE 0
#endif
// FWINO_TRACE (measurement builds): workgroup 0 records shader-clock timestamps of its wave 0 (consumer) and wave 8 (producer) at
// every barrier arrival / departure into the buffer named by the environment variable ICG_FWINO_TRACE_PTR (tools/fwino_trace.py)
#ifndef FWINO_TRACE
#define FWINO_TRACE 0
#endif
#ifndef FWINO_MASKBITS
#define FWINO_MASKBITS 0
#endif

struct FwinoP {
  const float* x;
  const float* Uf;       // fragment-major Winograd weights (icg_fwino_pack_kernel)
  const float* bias;
  const float* res;
  float* out;
  const float* scale;
  const float* shift;
  float* V;              // optional: the V planes [NP*NP][T][K] as a by-product (nullptr: not written)
  unsigned long long* trace;   // FWINO_TRACE builds only
  long ssb;              // per-sample stride of scale / shift (elements)
  long planeV;           // T * K
  int B, H, W;           // resolution of the Winograd domain (the conv's full resolution)
  int K, N;              // Cin, Cout
  int affine, relu, res_mode;
  float alpha;
  int tbw, tbh;          // 4x4-tile blocks per image row / column (W / 16, H / 16)
  int nblk;              // N / 96
  unsigned total;        // workgroups
  int swz;
};

template <int NP> __device__ __forceinline__ constexpr int fw_slot(int k) { return NP == 6 ? k : (k < 2 ? k : k - 1); }
template <int NP> __device__ __forceinline__ constexpr bool fw_has(int k) { return NP == 6 || k != 2; }

// scalar forms of the 1-D transforms of winograd.hip (w4_in6 / w4_in_up / w4_out4 / w4_out_pool), same operation order
__device__ __forceinline__ void fw_in6(const float d[6], float t[6]) {
  t[0] = fmaf(d[2], -5.f, d[0] * 4.f) + d[4];
  const float a = fmaf(d[2], -4.f, d[4]), b = fmaf(d[1], -4.f, d[3]);
  t[1] = a + b;
  t[2] = a - b;
  const float c = d[4] - d[2], e = (d[3] - d[1]) * 2.f;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = fmaf(d[3], -5.f, d[1] * 4.f) + d[5];
}
__device__ __forceinline__ void fw_in_up(const float l[4], float t[6]) {
  t[0] = fmaf(l[1], -5.f, l[0] * 4.f) + l[2];
  t[1] = fmaf(l[1], -8.f, l[2] * 2.f);
  t[2] = 0.f;
  const float c = l[2] - l[1];
  t[3] = c * 3.f;
  t[4] = c * -1.f;
  t[5] = fmaf(l[2], -5.f, l[1] * 4.f) + l[3];
}
__device__ __forceinline__ void fw_out4(const float m[6], float y[4]) {
  const float p = m[1] + m[2], q = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
  y[0] = (m[0] + p) + r;
  y[1] = fmaf(s, 2.f, q);
  y[2] = fmaf(r, 4.f, p);
  y[3] = fmaf(s, 8.f, q) + m[5];
}
__device__ __forceinline__ void fw_out_pool(const float m[6], float y[2]) {
  const float a = m[1] * 2.f;
  y[0] = fmaf(m[3], 3.f, m[0] + a) - m[4];
  y[1] = fmaf(m[4], -4.f, fmaf(m[3], 12.f, a)) + m[5];
}

__device__ __forceinline__ void fw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#if FWINO_TRACE
#define FW_BARRIER()                                                                                  \
  do {                                                                                                \
    if (tr_on && tr_n < 250) tr_buf[tr_n++] = clock64();                                              \
    fw_barrier();                                                                                     \
    if (tr_on && tr_n < 250) tr_buf[tr_n++] = clock64();                                              \
  } while (0)
#define FW_MARK()                                                                                     \
  do {                                                                                                \
    if (tr_on && tr_n < 250) tr_buf[tr_n++] = clock64() | (1ull << 63);                               \
  } while (0)
#else
#define FW_BARRIER() fw_barrier()
#define FW_MARK() do {} while (0)
#endif

// Uf[((p * NT + jt) * KG + kg) * 64 + l][e] = U[p][16 jt + (l & 15)][16 kg + 4 (l >> 4) + e]      (U: [planes][N][K])
__global__ __launch_bounds__(256) void icg_fwino_pack_kernel(const float* __restrict__ U, float* __restrict__ Uf, int planes, int N,
                                                             int K) {
  const int NT = N >> 4, KG = K >> 4;
  const long total = (long)planes * NT * KG * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    long r = i >> 6;
    const int kg = (int)(r % KG);
    r /= KG;
    const int jt = (int)(r % NT);
    const int pl = (int)(r / NT);
    const float4 v = *reinterpret_cast<const float4*>(U + ((long)pl * N + 16 * jt + (l & 15)) * K + 16 * kg + 4 * (l >> 4));
    reinterpret_cast<float4*>(Uf)[i] = v;
  }
}

// transforms on channel PAIRS (v_pk_* arithmetic): the producers' instruction count is what they cost the MFMA waves they share a
// SIMD with
// FWINO_SCALAR (measurement builds, tools/build_dbg.sh FWS): the same arithmetic on two plain floats (compile with
// -fno-slp-vectorize) -- packed fp32 VALU beside MFMAs has an issue cost of its own (MI355X_MICROARCH.md)
#ifndef FWINO_SCALAR
#define FWINO_SCALAR 0
#endif
#if FWINO_SCALAR
struct f32x2 { float x, y; };
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f32x2 operator*(f32x2 a, float b) { return f32x2{a.x * b, a.y * b}; }
__device__ __forceinline__ f32x2 fw2(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 fw_fma2(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#else
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fw2(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 fw_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
__device__ __forceinline__ f32x2 fw_fma(f32x2 a, float b, f32x2 c) { return fw_fma2(a, fw2(b), c); }
__device__ __forceinline__ void fw_in6(const f32x2 d[6], f32x2 t[6]) {
  t[0] = fw_fma(d[2], -5.f, d[0] * 4.f) + d[4];
  const f32x2 a = fw_fma(d[2], -4.f, d[4]), b = fw_fma(d[1], -4.f, d[3]);
  t[1] = a + b;
  t[2] = a - b;
  const f32x2 c = d[4] - d[2], e = (d[3] - d[1]) * 2.f;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = fw_fma(d[3], -5.f, d[1] * 4.f) + d[5];
}
__device__ __forceinline__ void fw_in_up(const f32x2 l[4], f32x2 t[6]) {
  t[0] = fw_fma(l[1], -5.f, l[0] * 4.f) + l[2];
  t[1] = fw_fma(l[1], -8.f, l[2] * 2.f);
  t[2] = fw2(0.f);
  const f32x2 c = l[2] - l[1];
  t[3] = c * 3.f;
  t[4] = c * -1.f;
  t[5] = fw_fma(l[2], -5.f, l[1] * 4.f) + l[3];
}

// the prologue operands of a layer without ICG_PRE_AFFINE: scale 1, shift 0 (the kernel loads them unconditionally, see load_affine)
#define FW_R8(x) x, x, x, x, x, x, x, x
#define FW_R64(x) FW_R8(x), FW_R8(x), FW_R8(x), FW_R8(x), FW_R8(x), FW_R8(x), FW_R8(x), FW_R8(x)
#define FW_R512(x) FW_R64(x), FW_R64(x), FW_R64(x), FW_R64(x), FW_R64(x), FW_R64(x), FW_R64(x), FW_R64(x)
__device__ float g_fw_ones[2048] = {FW_R512(1.f), FW_R512(1.f), FW_R512(1.f), FW_R512(1.f)};
__device__ float g_fw_zeros[2048] = {0.f};

template <int UP, int POOL, int NP>
__global__ __launch_bounds__(768) void icg_fwino_kernel(FwinoP p) {
  static_assert(!(UP && POOL), "no layer is resampled on both sides");
  static_assert((!UP && !POOL) || NP == 5, "resample-fused forms live in the 25-plane domain");
  constexpr int NPL = NP * NP;                // planes
  constexpr int NPW = (NPL + 3) / 4;          // plane slots per consumer wave
  constexpr int NL = UP ? 4 : 6;              // window size in stored pixels
  constexpr int NO = POOL ? 2 : 4;            // outputs per tile and dimension
  constexpr int VBUF = NPL * 2048;            // one V chunk: planes x 2 channel groups x 64 lanes x 16 B
  constexpr int MROW = 52;                    // padded row of the accumulator exchange (48 columns)
  constexpr int MBYTES = NPL * 16 * MROW * 4;
  constexpr int LDS_BYTES = (2 * VBUF > MBYTES) ? 2 * VBUF : MBYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..11
#if FWINO_TRACE
  const bool tr_on = p.trace && blockIdx.x == 0 && (tid == 0 || tid == 512);
  unsigned long long* tr_buf = p.trace + (tid == 0 ? 0 : 256);
  int tr_n = 0;
#endif

  // ---- this workgroup's RUN of work items (image b, tile block (by, bx), column block nb), XCD-aware as in pgemm.hip's streaming
  // kernel: each XCD owns a contiguous range of the item order and its workgroups stride through it
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

  const int K = p.K, N = p.N, H = p.H, W = p.W;
  const int nc = K >> 5;                                            // 32-channel chunks
  const int th = H >> 2, tw = W >> 2;
  struct Item { int nb, bx, by, b; };
  auto decode = [&](unsigned t) -> Item {
    Item it;
    it.nb = (int)(t % (unsigned)p.nblk);
    unsigned tb = t / (unsigned)p.nblk;
    it.bx = (int)(tb % (unsigned)p.tbw);
    tb /= (unsigned)p.tbw;
    it.by = (int)(tb % (unsigned)p.tbh);
    it.b = (int)(tb / (unsigned)p.tbh);
    return it;
  };

  // ---- output side, shared by both roles: item -> (tile, column) of a 48-column round (768 items per round)
  const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
  auto output_item = [&](const Item& w, int round, int item) {
#if FWINO_ABLATE & 8
    return;
#endif
    asm volatile("" : "+v"(item));              // (per-item address arithmetic stays here: hoisted out of the persistent loop it spills)
    const int otile = item / 48, ocol = item - 48 * otile;
    const int oty = 4 * w.by + (otile >> 2), otx = 4 * w.bx + (otile & 3);
    const int n = w.nb * 96 + round * 48 + ocol;
    const float* mp = reinterpret_cast<const float*>(lds) + otile * MROW + ocol;
    // bias and the residual operand (plain, upsample-on-read, ReLU mask) FIRST and unconditionally -- all of them in flight under
    // the LDS reads and the transform below; without a residual / bias they read one zero.  (Loaded one by one where they are
    // used, each behind an `if (p.res)`, every one of the 16 loads was a full s_waitcnt vmcnt(0) round trip: the output stage
    // of an item cost 18 000 of its 71 000 clocks, tools/fwino_trace.py.)
    // 32-bit byte offsets throughout (out and the residual operand are < 2^32 - 2^24 bytes, icg_fwino_applies); the residual's
    // geometry (same as out / half resolution, nearest upsampling on read / none: offset 0 into one zero) as uniform selects
    const float* const resp = p.res ? p.res : g_fw_zeros;
    const float bv = (p.bias ? p.bias : g_fw_zeros)[p.bias ? n : 0];
    const bool rup = !POOL && p.res_mode == 1;
    const unsigned N4 = (unsigned)N * 4u, rN4 = p.res ? N4 : 0u;
    const int rH = rup ? (H >> 1) : Ho, rW = rup ? (W >> 1) : Wo, rx = rup ? 2 * otx : NO * otx;
    unsigned pb[NO];
    float rv[NO][NO];
#pragma unroll
    for (int a = 0; a < NO; ++a) {
      const int oy = NO * oty + a;
      pb[a] = (unsigned)(((w.b * Ho + oy) * Wo + NO * otx) * N + n) * 4u;
      const unsigned rb = p.res ? (unsigned)(((w.b * rH + (rup ? (oy >> 1) : oy)) * rW + rx) * N + n) * 4u : 0u;
#pragma unroll
      for (int c = 0; c < NO; ++c)
        rv[a][c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(resp) + (rb + (unsigned)(rup ? (c >> 1) : c) * rN4));
    }
    // y = A^T M A accumulated column by column: for column j of M the 1-D transform yy = A^T M[:, j], then
    // y[a][c] += yy[a] * A[j][c] with the constants of A folded (rows of A^T: [1 1 1 1 1 0], [0 1 -1 2 -2 0], [0 1 1 4 4 0],
    // [0 1 -1 8 -8 1]; pooled: [1 2 0 3 -1 0], [0 2 0 12 -4 1]) -- 16 + 6 + 4 live values instead of 36 + 24
    constexpr float AT4[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
    constexpr float ATP[2][6] = {{1, 2, 0, 3, -1, 0}, {0, 2, 0, 12, -4, 1}};
    float y[NO][NO];
#pragma unroll
    for (int a = 0; a < NO; ++a)
#pragma unroll
      for (int c = 0; c < NO; ++c) y[a][c] = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (!fw_has<NP>(j)) continue;
      float col[6], yy[4];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = fw_has<NP>(r) ? mp[(fw_slot<NP>(r) * NP + fw_slot<NP>(j)) * 16 * MROW] : 0.f;
      if constexpr (POOL) fw_out_pool(col, yy); else fw_out4(col, yy);
#pragma unroll
      for (int a = 0; a < NO; ++a)
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          const float cw = POOL ? ATP[c][j] : AT4[c][j];
          if (cw == 1.f) y[a][c] += yy[a];
          else if (cw == -1.f) y[a][c] -= yy[a];
          else if (cw != 0.f) y[a][c] = fmaf(yy[a], cw, y[a][c]);
        }
    }
#pragma unroll
    for (int a = 0; a < NO; ++a) {
#pragma unroll
      for (int c = 0; c < NO; ++c) {
        float v = p.alpha * y[a][c] + bv;
        if (p.res) v = (p.res_mode == 2) ? (rv[a][c] > 0.f ? v : 0.f) : v + rv[a][c];
        *reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + (pb[a] + (unsigned)c * N4)) = v;
      }
    }
  };

  // Barriers per work item, the same sequence in every role: [T(0) written] [chunk 0 consumed / T(1) written] ... [chunk nc-1
  // consumed] [round-0 accumulators handed over] [round-0 items done] [round-1 handed over] [round-1 items done: LDS free].
  if (wv >= 8) {
    // =========================================================== producers ===========================================================
    // lane = (tile ti, channel pair cp) of a 32-channel chunk: ONE item per lane and chunk, everything on float2.
    // Addressing: wave-uniform base pointer + a 32-bit lane offset per access, rebuilt from row / column terms at every use
    // (opaque to the optimiser on purpose: hoisted 64-bit addresses are what spilled this branch).
    // Priority: a producer wave's ~500 instructions per chunk go FIRST on its SIMD.  The two MFMA waves it shares the SIMD with
    // have slack (they wait for this wave at every chunk barrier); measured with tools/fwino_trace.py the other way round
    // (MFMA waves at priority 1): the producer was not issued for ~10 000 clocks after each barrier and the MFMA waves then idled
    // ~15 000 clocks per chunk at the next one.
    __builtin_amdgcn_s_setprio(3);
    const int pl = tid - 512;
    const int Hx = UP ? (H >> 1) : H, Wx = UP ? (W >> 1) : W;       // stored tensor
    const unsigned WxK = (unsigned)(Wx * K);
    // LDS position of a lane's pair inside a plane's 2 KiB: [sg][kq*16 + (tile ^ (kq + 4 sg))][s4], channel = 16 sg + 4 kq + s4

    f32x2 d[NL][NL], d_sc, d_sh;
    // Window loads: buffer loads through ONE descriptor over x, everything in voffset = the lane's window origin + a wave-uniform
    // (row, column) term.  No clamping: rows / columns outside the image land in the neighbouring row / image (valid memory, the
    // transform zeroes them) or outside the tensor, where the range check of the buffer load returns 0 instead of faulting
    // (offsets wrap modulo 2^32; x is < 2^32 - 2^24 bytes, icg_fwino_applies).
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((unsigned)(p.B * Hx * Wx) * (unsigned)K * 4u),
                                                       0x00020000);
    const unsigned RowB = WxK * 4u, ColB = (unsigned)K * 4u;
    // byte offset of window element (0, 0) of this lane's tile in work item w, chunk ck, channel pair cp
    auto win_base = [&](const Item& w, int ck, int cp, int ti) -> unsigned {
      const int txg = 4 * w.bx + (ti & 3), tyg = 4 * w.by + (ti >> 2);
      const int w0 = UP ? 2 * txg - 1 : 4 * txg - 1, h0 = UP ? 2 * tyg - 1 : 4 * tyg - 1;
      unsigned wb = (((unsigned)(w.b * Hx + h0) * (unsigned)Wx + (unsigned)w0) * (unsigned)K + (unsigned)(32 * ck + 2 * cp)) * 4u;
      asm volatile("" : "+v"(wb));               // (opaque: hoisted address arithmetic is what spilled this branch)
      return wb;
    };
    auto load_px = [&](unsigned wb, int r, int s) -> f32x2 {
      return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)(wb + (unsigned)r * RowB + (unsigned)s * ColB), 0, 0));
    };
    // The chunk's prologue scale / shift travel with its window and go FIRST (one order of the in-order vmcnt queue on every path);
    // unconditional -- without ICG_PRE_AFFINE they come from a row of ones / zeros: a load under a run-time flag
    // is a phi of registers, i.e. a copy of the loaded value, i.e. a full wait right behind the load.
    const float* const scale_p = p.affine ? p.scale : g_fw_ones;     // (uniform, once per kernel)
    const float* const shift_p = p.affine ? p.shift : g_fw_zeros;
    const long ssb_p = p.affine ? p.ssb : 0;
    auto load_affine = [&](const Item& w, int ck, int cp) {
      d_sc = *reinterpret_cast<const f32x2*>(scale_p + (long)w.b * ssb_p + 32 * ck + 2 * cp);
      d_sh = *reinterpret_cast<const f32x2*>(shift_p + (long)w.b * ssb_p + 32 * ck + 2 * cp);
    };
    auto load_chunk = [&](const Item& w, int ck) {                   // the whole window at once (first chunk of the run; UP form)
      int cp = pl & 15, ti = pl >> 4;
      asm volatile("" : "+v"(cp), "+v"(ti));
      const unsigned wb = win_base(w, ck, cp, ti);
      load_affine(w, ck, cp);
#pragma unroll
      for (int r = 0; r < NL; ++r)
#pragma unroll
        for (int s = 0; s < NL; ++s) d[r][s] = load_px(wb, r, s);
    };

    // A chunk = FIRST pass (1-D transform in place; prologue: scale / shift, ReLU, zero padding) + SECOND pass (the other dimension:
    // a window line in, a line of V out to LDS / HBM).  The second pass retires the window line by line, and every retired line is
    // refilled at once with the matching line of the NEXT chunk's window (same registers), so that the loads have half a transform
    // of distance instead of none (measured before, tools/fwino_trace.py: the producer waited ~16 000 clocks for its window after
    // every barrier, the MFMA waves the same time for the producer).  A window refilled by COLUMNS can only be transformed
    // columns-first, one refilled by rows rows-first: the two orders alternate (YV); B^T (d B) and (B^T d) B differ in rounding
    // order only.  The 4 x 4 window of the upsample-fused form is dead after its first pass (it expands into Eu) and is refilled as
    // a whole at the start of the second.
    constexpr int EW = UP ? 6 : 1;
    f32x2 Eu[UP ? NL : 1][EW];
    auto first_pass = [&](auto yv, const Item& w) {
      constexpr bool YV = decltype(yv)::value;      // false: rows then columns (window arrived by rows); true: columns then rows
      static_assert(!(UP && YV), "the upsample-fused form has one order");
      int ti = pl >> 4;
      asm volatile("" : "+v"(ti));
      const int txg = 4 * w.bx + (ti & 3), tyg = 4 * w.by + (ti >> 2);
      const int w0 = UP ? 2 * txg - 1 : 4 * txg - 1, h0 = UP ? 2 * tyg - 1 : 4 * tyg - 1;
      const f32x2 sc = d_sc, sh = d_sh;                             // (1, 0) without ICG_PRE_AFFINE
      const float lo = p.relu ? 0.f : -__builtin_inff();
      FW_MARK();
      // (the zero-padding masks run for every tile block: a workgroup-uniform branch around a mask-free copy for the blocks off
      // the image border -- 72 of ~520 instructions -- does not compile in this loop shape: "illegal VGPR to SGPR copy")
#pragma unroll
      for (int a = 0; a < NL; ++a) {                                 // a: window row (rows first) / window column (columns first)
        const bool aok = YV ? (unsigned)(w0 + a) < (unsigned)Wx : (unsigned)(h0 + a) < (unsigned)Hx;
        f32x2 line[NL], t6[6];
#pragma unroll
        for (int b = 0; b < NL; ++b) {
          const bool bok = YV ? (unsigned)(h0 + b) < (unsigned)Hx : (unsigned)(w0 + b) < (unsigned)Wx;
          // branch-free prologue: scale 1 / shift 0 without ICG_PRE_AFFINE (exact), ReLU floor -inf without ICG_PRE_RELU
          // (if-converted run-time flags cost a v_cndmask per value and flag: 144 of this loop's instructions)
          f32x2 v = fw_fma2(YV ? d[b][a] : d[a][b], sc, sh);
          v = f32x2{fmaxf(v.x, lo), fmaxf(v.y, lo)};
          line[b] = (aok && bok) ? v : fw2(0.f);
        }
        if constexpr (UP) {
          fw_in_up(line, t6);
#pragma unroll
          for (int j = 0; j < 6; ++j) Eu[a][j] = t6[j];
        } else {
          fw_in6(line, t6);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            if constexpr (YV) d[j][a] = t6[j]; else d[a][j] = t6[j];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FW_MARK();
    };
    // second pass of chunk (w, ck) into the V buffer at vb; refill with the window of chunk (wl, ckl)
    auto second_pass = [&](auto yv, const Item& w, int ck, unsigned vb, const Item& wl, int ckl) {
      constexpr bool YV = decltype(yv)::value;
      int cp = pl & 15, ti = pl >> 4;
      asm volatile("" : "+v"(cp), "+v"(ti));
      const int sg = cp >> 3, kq = (cp >> 1) & 3;
      const unsigned wpos0 = (unsigned)((sg * 64 + kq * 16 + (ti ^ (kq + 4 * sg))) * 16 + (cp & 1) * 8);
      const int txg = 4 * w.bx + (ti & 3), tyg = 4 * w.by + (ti >> 2);
      unsigned wbn = 0u;
      if constexpr (UP) {
        load_chunk(wl, ckl);
      } else {
        load_affine(wl, ckl, cp);
        wbn = win_base(wl, ckl, cp, ti);
      }
      unsigned wpos = vb + wpos0;
      unsigned vpos = (unsigned)((((w.b * th + tyg) * tw + txg) * K + 2 * cp)) * 4u;       // byte offset inside a V plane
      const unsigned pv4 = (unsigned)p.planeV * 4u;                 // bytes per plane; planes x pv4 < 2^32 (icg_fwino_conv)
      asm volatile("" : "+v"(wpos), "+v"(vpos));
      const bool wantV = (p.V != nullptr) && w.nb == 0;
      char* vplane = reinterpret_cast<char*>(p.V + 32 * ck);        // uniform
#pragma unroll
      for (int j = 0; j < 6; ++j) {             // line j of the transformed domain (rows first: column j; columns first: row j)
        if (fw_has<NP>(j)) {
          f32x2 col[NL], o[6];
#pragma unroll
          for (int r = 0; r < NL; ++r) {
            if constexpr (UP) col[r] = Eu[r][j]; else col[r] = YV ? d[j][r] : d[r][j];
          }
          if constexpr (UP) fw_in_up(col, o); else fw_in6(col, o);
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            if (!fw_has<NP>(r)) continue;
            const int plane = YV ? fw_slot<NP>(j) * NP + fw_slot<NP>(r) : fw_slot<NP>(r) * NP + fw_slot<NP>(j);
            *reinterpret_cast<f32x2*>(lds + wpos + plane * 2048) = o[r];
            if (wantV) *reinterpret_cast<f32x2*>(vplane + (size_t)((unsigned)plane * pv4 + vpos)) = o[r];
          }
        }
        if constexpr (!UP) {                                         // register line j is retired: the next window's line j
#pragma unroll
          for (int q = 0; q < NL; ++q) {
            if constexpr (YV) d[j][q] = load_px(wbn, j, q); else d[q][j] = load_px(wbn, q, j);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      FW_MARK();
    };

    // ---- the run.  Straight-line code per pair of chunks, rotated so that the loop's back edge (and its entry) sit BEHIND a first
    // pass: where two code paths join, the register allocator moves values between the registers each side keeps them in, and a
    // move of a window element that is still in flight is a full wait -- behind a first pass nothing is in flight.  (One call site
    // behind a flag instead of the two orders in sequence: 9 - 36 spilled VGPRs.)
    unsigned v = first;
    Item cur = decode(first), w = cur;                               // w: the item whose chunk ck is in work; cur: the item after it
    int ck = 0;
    bool has_next = v + vstep < last;
    if (has_next) cur = decode(v + vstep);
    // (w, cur, ck are loop-carried state; the compiler wants them in SGPRs at their uses and needs to be told it may)
    auto uni = [&](const Item& x) -> Item {
      Item y;
      y.nb = __builtin_amdgcn_readfirstlane(x.nb); y.bx = __builtin_amdgcn_readfirstlane(x.bx);
      y.by = __builtin_amdgcn_readfirstlane(x.by); y.b = __builtin_amdgcn_readfirstlane(x.b);
      return y;
    };
    auto second = [&](auto yv) {
#if !(FWINO_ABLATE & 1)
      // the refill: the next chunk of this item, else the next item's first one -- in flight during this item's last MFMAs and its
      // output stage -- else (end of the run) this chunk again: unconditional, see load_affine
      const int cku = __builtin_amdgcn_readfirstlane(ck);
      const bool same = cku + 1 < nc;
      const Item wu = uni(w), wl = uni(same ? w : cur);
      second_pass(yv, wu, cku, (cku & 1) ? (unsigned)VBUF : 0u, wl, same ? cku + 1 : (has_next ? 0 : cku));
#endif
    };
    auto first_of = [&](auto yv) {
#if !(FWINO_ABLATE & 1)
      first_pass(yv, uni(w));
#endif
    };
    // after a chunk: barrier k of the item (T(k) written, chunk k - 1 consumed), then the FIRST pass of the chunk that follows
    // (`next_first`); behind an item's last chunk the rest of its barrier sequence and this role's share of the output stage --
    // with the first pass of the next item's chunk 0 (its window arrived long ago) in front of it, in the time the first-round
    // consumers need to hand their accumulators over, instead of behind it where the MFMA waves wait for T(0).  false: run over
    auto post = [&](auto next_first) -> bool {
      FW_BARRIER();
      const bool item_done = ++ck >= nc;
      Item wo = uni(w);                                              // the item whose outputs are due
      bool more = true;
      if (item_done) {
        FW_BARRIER();                                                // chunk nc - 1 consumed
        v += vstep;
        more = v < last;
        if (more) {
          w = cur;
          ck = 0;
          has_next = v + vstep < last;
          if (has_next) cur = decode(v + vstep);
        }
      }
      if (more) next_first();                                        // (ONE call site: see the note on joins above)
      if (item_done) {
        // Round 0 is worked by the 512 threads that hold no accumulators any more (first-round consumers + producers), the
        // second-round consumers wait with theirs: no point of the program has 108 accumulators AND an output transform live.
        FW_BARRIER();
        output_item(wo, 0, tid - 256);                               // items 256..511 (the first-round consumers: 0..255, 512..767)
        FW_BARRIER();
        FW_BARRIER();
        output_item(wo, 1, tid);
        FW_BARRIER();
      }
      return more;
    };
#if !(FWINO_ABLATE & 1)
    load_chunk(w, 0);
#endif
    first_of(std::false_type{});
    for (;;) {
      second(std::false_type{});
      if constexpr (UP) {
        if (!post([&] { first_of(std::false_type{}); })) break;
      } else {
        if (!post([&] { first_of(std::true_type{}); })) break;
        second(std::true_type{});
        if (!post([&] { first_of(std::false_type{}); })) break;
      }
    }
  } else {
    // =========================================================== consumers ===========================================================
    const int pg = wv & 3, ng = wv >> 2;
    const int it = lane & 15, kq = lane >> 4;
    f32x4 acc[NPW][3];
    const unsigned apos[2] = {(unsigned)((kq * 16 + (it ^ kq)) * 16), (unsigned)((64 + kq * 16 + (it ^ (kq + 4))) * 16)};
    // weight fragments: buffer loads through ONE wave-uniform descriptor over Uf (SGPRs), the per-lane part of the address is
    // 16 B x lane in voffset and everything else (column block, column group, plane, chunk) is scalar arithmetic in soffset
    const int NT = N >> 4, KG = K >> 4;
    const unsigned pstride = (unsigned)NT * (unsigned)KG * 1024u;    // bytes per plane of Uf
    const unsigned jstride = (unsigned)KG * 1024u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Uf), 0, (int)((unsigned)NPL * (unsigned)N * (unsigned)K * 4u),
                                                      0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto ubase_of = [&](int nb) -> unsigned {
      return (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(nb * 6 + ng * 3) * jstride + (unsigned)pg * pstride));
    };
    auto load_b = [&](f32x4 (&bf)[3], unsigned ubase, unsigned ckoff, int i, int sgg) {      // ckoff = 2048 x chunk (bytes)
      const unsigned so = ubase + 4u * (unsigned)i * pstride + ckoff + (unsigned)sgg * 1024u;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        bf[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane16, (int)(so + (unsigned)j * jstride), 0));
    };
    auto valid = [&](int i) -> bool { return NPL % 4 == 0 || i < NPW - 1 || pg + 4 * i < NPL; };

    f32x4 bcur[3], acur;
    Item w = decode(first);
    unsigned ubase = ubase_of(w.nb);
    load_b(bcur, ubase, 0u, 0, 0);
    for (unsigned v = first; v < last; v += vstep) {
      const bool has_next = v + vstep < last;
      Item wn = w;
      if (has_next) wn = decode(v + vstep);
      const unsigned ubase_next = ubase_of(wn.nb);
#pragma unroll
      for (int i = 0; i < NPW; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      FW_BARRIER();
      for (int ck = 0; ck < nc; ++ck) {
        unsigned ckoff = (unsigned)ck * 2048u;
        // the prefetch past this chunk: the next chunk, or the first fragments of the next work item
        unsigned ckoff_next = (ck + 1 < nc) ? ckoff + 2048u : 0u, ub_next = (ck + 1 < nc) ? ubase : ubase_next;
        asm volatile("" : "+s"(ckoff), "+s"(ckoff_next), "+s"(ub_next));   // (offsets are rebuilt per chunk, not hoisted)
        const char* vbase = lds + ((ck & 1) ? VBUF : 0) + pg * 2048;
        acur = *reinterpret_cast<const f32x4*>(vbase + apos[0]);
#pragma unroll
        for (int q = 0; q < 2 * NPW; ++q) {
          const int i = q >> 1, sgg = q & 1;
          f32x4 bnext[3], anext;
          const bool lastq = (q + 1 == 2 * NPW);
          const int ni = lastq ? 0 : (q + 1) >> 1, nsg = lastq ? 0 : (q + 1) & 1;
          // the next half-step's operands first: a full half-step (12 MFMAs of this wave + those of its SIMD neighbour) of distance
          if (valid(ni)) {
#if !(FWINO_ABLATE & 4)
            if (lastq) load_b(bnext, ub_next, ckoff_next, 0, 0);
            else load_b(bnext, ubase, ckoff, ni, nsg);
#else
#pragma unroll
            for (int j = 0; j < 3; ++j) bnext[j] = bcur[j];
#endif
            if (!lastq) anext = *reinterpret_cast<const f32x4*>(vbase + ni * 4 * 2048 + apos[nsg]);
          }
          __builtin_amdgcn_sched_barrier(0);
#if !(FWINO_ABLATE & 2)
          if (valid(i)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[j][s], acur[s], acc[i][j], 0, 0, 0);
          }
#endif
#pragma unroll
          for (int j = 0; j < 3; ++j) bcur[j] = bnext[j];
          acur = anext;
          __builtin_amdgcn_sched_barrier(0);    // one half-step of prefetch distance, no more (register budget: 168)
        }
        FW_BARRIER();
      }
      auto hand_over = [&]() {
        int hb = (it * MROW + 4 * kq) * 4;
        asm volatile("" : "+v"(hb));
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          const int plane = pg + 4 * i;
          if (NPL % 4 != 0 && i == NPW - 1 && plane >= NPL) continue;
#pragma unroll
          for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(lds + hb + (plane * 16 * MROW + 16 * j) * 4) = acc[i][j];
        }
      };
      if (ng == 0) {
        hand_over();
        FW_BARRIER();
        output_item(w, 0, tid);                                      // items 0..255 and 512..767
        output_item(w, 0, tid + 512);
        FW_BARRIER();
        FW_BARRIER();
      } else {
        FW_BARRIER();
        FW_BARRIER();
        hand_over();
        FW_BARRIER();
      }
      output_item(w, 1, tid);
      FW_BARRIER();                                                  // the exchange region is the V ring again
      w = wn;
      ubase = ubase_next;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static int fwino_env(const char* name, int dflt) {      // read per call: a measurement / test switch that can be flipped at run time
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Does the fused kernel take this layer on its own?  (H, W: resolution of the Winograd domain.)  K <= 192: single-level chains.
extern "C" int icg_fwino_applies(int B, int H, int W, int Cin, int Cout) {
  if (!fwino_env("ICG_FWINO", 1)) return 0;
  if (Cin % 32 || Cin > fwino_env("ICG_FWINO_MAXK", 192) || Cout % 96 || Cout > fwino_env("ICG_FWINO_MAXN", 192) || H % 16 || W % 16)
    return 0;
  // x behind ONE buffer descriptor with 32-bit offsets, 32-bit byte offsets into out / the residual operand (2^32 - 2^24)
  if ((double)B * H * W * Cin * 4.0 >= 4278190080.0 || (double)B * H * W * Cout * 4.0 >= 4278190080.0) return 0;
  const long wgs = (long)B * (H / 16) * (W / 16) * (Cout / 96);
  return (wgs >= fwino_env("ICG_FWINO_MIN_WGS", 512) && wgs < 0x7fffffffL) ? 1 : 0;
}

extern "C" size_t icg_fwino_weight_bytes(int planes, int Cin, int Cout) { return (size_t)planes * Cin * Cout * sizeof(float); }

// U [planes][Cout][Cin] (icg_wino4_weight_transform / icg_wino4r_weight_transform) -> fragment-major Uf, same size
extern "C" int icg_fwino_pack_weights(const float* U, float* Uf, int planes, int Cin, int Cout, void* stream) {
  ICG_REQUIRE(U && Uf && (planes == 25 || planes == 36) && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0);
  const long total = (long)planes * (Cout / 16) * (Cin / 16) * 64;
  long nb = icg_cdiv(total, 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(icg_fwino_pack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, U, Uf, planes, Cout, Cin);
  return icg_check_launch();
}

void icg_gemm_set_last_variant(int a, int b, int c, int d);     // gemm_conv.hip: the label bench.py reads back


// The fused forward: out = epilogue(conv3x3(prologue(x))) with x / out resampled as (in_up, out_pool) say; Uf from
// icg_fwino_pack_weights; V (optional) receives the transformed input planes [np*np][T][Cin] as a by-product.
int icg_fwino_run(const float* x, int in_up, const float* Uf, const float* bias, const float* residual, int res_mode, float* out,
                  int out_pool, const float* scale, const float* shift, int64_t ssb, int B, int H, int W, int Cin, int Cout,
                  unsigned flags, float alpha, int np, float* V, void* stream) {
  static const bool no_swz = [] { const char* e = getenv("ICG_NO_XCD_SWIZZLE"); return e && e[0] == '1'; }();
  FwinoP p;
  p.trace = nullptr;
#if FWINO_TRACE
  if (const char* e = getenv("ICG_FWINO_TRACE_PTR")) p.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
  p.x = x; p.Uf = Uf; p.bias = bias; p.res = residual; p.out = out; p.scale = scale; p.shift = shift; p.V = V;
  p.ssb = (long)ssb;
  p.planeV = (long)B * (H / 4) * (W / 4) * Cin;
  p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout;
  p.affine = (flags & ICG_PRE_AFFINE) ? 1 : 0;
  p.relu = (flags & ICG_PRE_RELU) ? 1 : 0;
  p.res_mode = res_mode;
  p.alpha = alpha;
  p.tbw = W / 16; p.tbh = H / 16; p.nblk = Cout / 96;
  p.total = (unsigned)((long)B * p.tbw * p.tbh * p.nblk);
  // persistent workgroups: one per CU (147 KiB of LDS and 12 x 168 registers each), striding through the work items of its XCD
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n - n % 8;
  }();
  unsigned grid = p.total < (unsigned)n_cu ? p.total : (unsigned)n_cu;
  p.swz = (p.total >= 16 && grid % 8 == 0 && !no_swz) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(768);
  icg_gemm_set_last_variant(5, in_up ? 1 : 0, out_pool ? 1 : 0, np);
  if (in_up) {
    if (np != 5 || out_pool) return ICG_ERR_ARG;
    hipLaunchKernelGGL((icg_fwino_kernel<1, 0, 5>), g, blk, 0, st, p);
  } else if (out_pool) {
    if (np != 5) return ICG_ERR_ARG;
    hipLaunchKernelGGL((icg_fwino_kernel<0, 1, 5>), g, blk, 0, st, p);
  } else {
    if (np != 6) return ICG_ERR_ARG;
    hipLaunchKernelGGL((icg_fwino_kernel<0, 0, 6>), g, blk, 0, st, p);
  }
  return icg_check_launch();
}

// The kernel as an entry point of its own (tests, microbenchmarks; the layer entry points of winograd.hip route to it by
// icg_fwino_applies): H, W = resolution of the Winograd domain (multiples of 16), Cin % 32 == 0, Cout % 96 == 0;
// (in_up, out_pool) in {(0,0): 36 planes, (1,0) / (0,1): 25 planes}; V: nullptr or [planes][B H/4 W/4][Cin].
extern "C" int icg_fwino_conv(const float* x, const float* Uf, const float* bias, const float* residual, float* out,
                              const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin, int Cout,
                              unsigned flags, float alpha, int in_up, int out_pool, float* V, void* stream) {
  ICG_REQUIRE(x && Uf && out && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ICG_REQUIRE(H % 16 == 0 && W % 16 == 0 && Cin % 32 == 0 && Cout % 96 == 0 && !(in_up && out_pool));
  if (flags & ICG_PRE_AFFINE) ICG_REQUIRE(scale && shift);
  if (flags & ICG_RES_RELU_MASK) ICG_REQUIRE(residual && !(flags & ICG_RES_UPSAMPLE2X));
  const int np = (in_up || out_pool) ? 5 : 6;
  const double T = (double)B * (H / 4) * (W / 4);
  ICG_REQUIRE(T * (Cout / 96) / 16 < 2147483647.0);
  if (V) ICG_REQUIRE((double)np * np * T * Cin * 4.0 < 4294967296.0);
  ICG_REQUIRE((double)B * H * W * Cin * 4.0 < 4278190080.0 && (double)B * H * W * Cout * 4.0 < 4278190080.0 && Cin <= 2048);
  return icg_fwino_run(x, in_up, Uf, bias, residual, icg_res_mode(flags), out, out_pool, scale, shift, ss_bstride, B, H, W, Cin,
                       Cout, flags, alpha, np, V, stream);
}
