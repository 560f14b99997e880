// Gather-GEMM: the one MFMA kernel every 3x3 conv forward and data-gradient is lowered to.
//
//   Y[dst(m), n] = epilogue( sum_{t<ntaps} sum_{c<Cred} X[src(m,t), c] * Wp[n][tap_w[t]][c] )
//
// GEMM view: rows m = output pixels (img,gy,gx), columns n = output channels,
// reduction K = taps x channels, walked in 16-byte "chunks" (8 bf16 / 4 fp32 channels) so that the
// same staging code serves both precisions:
//   bf16 : v_mfma_f32_16x16x32_bf16  (one MFMA per 16-B fragment pair)
//   fp32 : v_mfma_f32_16x16x4_f32    (four MFMAs per 16-B fragment pair; exact fp32, parity mode)
// The K order inside a fragment is a permutation of the natural one; both operands use the same
// permutation, so the sum is unchanged up to fp32 re-association.
//
// Tile: BP pixels x BC channels per 256-thread workgroup (4 waves), K-step = 8 chunks (128 B per
// row).  NHWC makes every 128-B row piece a full contiguous line of one source pixel.  Operands are
// staged global -> registers -> LDS (double buffered, one barrier per K-step; the global loads of
// step k+1 are in flight while step k computes).  LDS rows are 128 B, XOR-swizzled by
// chunk ^= (row>>1)&7 so that the 16 lanes of a ds_read_b128 group hit 16 different 16-B slots.
// MFMA orientation: A = weights (rows = channels), B = pixels: each lane then owns 4 CONSECUTIVE
// channels of one pixel per accumulator fragment.  In the 64-channel wave tiles the weight rows are staged in the
// order perm64 (below), which makes a lane's FOUR fragments 16 consecutive channels: the epilogue then moves 16-byte
// vectors (two stores of 8 bf16 instead of four of 4), and a wave store covers whole 128-B runs of a pixel -- measured
// with tools/store_probe.hip: 6.1 TB/s against 4.6 TB/s (4.0 with a destination pixel stride of 2) for the 8-byte form.
#include "dg_internal.h"

#include <stdlib.h>
#include <type_traits>

// bit mask of the kernel variants the last dg_conv3x3_fwd / _dgrad call of this thread launched (bench.py tags its
// live timings with it): 1 generic, 2 fast, 8 halo, 16 im2col
static thread_local int g_last_kinds = 0;

struct GGArgs {
  const void* x; const void* w; void* y;
  const float* bias; const void* r1; const void* r2; const void* mask;
  const void* mask_bits; void* out_bits;     // 1-bit LeakyReLU masks (u16 per lane: 4 fragments x 4 channels), see dg_epilogue
  void* out_q; void* out_qs;                 // MXFP8 copy of the stored output (dg_epilogue.out_q / out_qs)
  int ldqs, qs_shift;                        // scale bytes per pixel of out_qs; log2(Nout / 16) when that stride is not Nout / 32
  long long ldx, ldw, ldy, ldr1, ldr2, ldmask;
  int M, Hg, Wg, Hs, Ws;
  int cch, kchunks, Cred, ntaps;
  int sy_mul, sx_mul;
  unsigned long long tap_lo; unsigned tap_hi;
  int Nout, Hd, Wd, dy_mul, dx_mul, dy_off, dx_off;
  int src_ps, dst_ps, cps_src_chunks, cps_dst;
  int has_act, accumulate;
  float act_slope, s1, s2, mask_slope;
  unsigned nwg, nct;
  int mask_c0, mask_last;                      // dg_epilogue: mask for channels >= mask_c0 only, applied after the accumulate
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define DG_OOB_OFF 0x80000000u   // voffset >= num_records: buffer loads return 0, buffer stores are dropped

// Output-channel order of a 64-channel wave tile.  An MFMA D fragment gives lane group g rows 4g..4g+3 of 16-row block j.
// Weight-tile row 16j + m holds channel 16*(m >> 2) + 4j + (m & 3) of the tile (the bit pairs [5:4] and [3:2] of the row
// index swapped; an involution), so accumulator acc[j][.][e] of lane group g is channel 16g + 4j + e: 16 consecutive
// channels per lane.  Every kernel with 64-channel wave tiles stages its weight rows through this map; the narrow tiles
// (Nout <= 64) keep the natural order.
__host__ __device__ constexpr int perm64(int r) { return (r & ~63) | ((r & 0x0c) << 2) | ((r >> 2) & 0x0c) | (r & 3); }

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
// 16-byte epilogue vectors of the permuted tiles: 8 bf16 or 4 fp32 consecutive channels
template <typename T> struct EpiV;
template <> struct EpiV<bf16_t> {
  static constexpr int CPU = 8, NU = 2;            // channels per unit, units per lane (16 channels)
  static __device__ __forceinline__ void unpack(const u32x4_t& t, float* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(t[q] << 16); v[2 * q + 1] = __uint_as_float(t[q] & 0xffff0000u); }
  }
  static __device__ __forceinline__ u32x4_t pack(const float* v) {
    u32x4_t t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = pack_bf16x2(v[2 * q], v[2 * q + 1]);
    return t;
  }
};
template <> struct EpiV<float> {
  static constexpr int CPU = 4, NU = 4;
  static __device__ __forceinline__ void unpack(const u32x4_t& t, float* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = __uint_as_float(t[q]);
  }
  static __device__ __forceinline__ u32x4_t pack(const float* v) {
    u32x4_t t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = __float_as_uint(v[q]);
    return t;
  }
};
struct EpiRes { __amdgpu_buffer_rsrc_t rY, r1, r2, rm, rbi, rbo, rq, rqs; int ldy, ld1, ld2, ldm; };
template <typename T> struct EpiIO;
template <> struct EpiIO<bf16_t> {
  typedef u32x2_t V;
  static __device__ __forceinline__ V load(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
  static __device__ __forceinline__ void store(const float* v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    V t;
    t[0] = pack_bf16x2(v[0], v[1]);
    t[1] = pack_bf16x2(v[2], v[3]);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, off, 0, 0);
  }
  static __device__ __forceinline__ void unpack(const V& t, float* v) {
    v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
    v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
  }
};
template <> struct EpiIO<float> {
  typedef u32x4_t V;
  static __device__ __forceinline__ V load(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
  static __device__ __forceinline__ void store(const float* v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    V t;
    t[0] = __float_as_uint(v[0]); t[1] = __float_as_uint(v[1]); t[2] = __float_as_uint(v[2]); t[3] = __float_as_uint(v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 0);
  }
  static __device__ __forceinline__ void unpack(const V& t, float* v) {
    v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
  }
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// ---- One pixel of a permuted 64-channel wave tile: the lane's 16 consecutive channels (fragments f0..f3 = 4 channels
// each) at byte offset `off` of every operand tensor (an out-of-range offset drops the stores and zero-fills the loads).
// `bias` = the lane's 16 bias values, `boff` = byte offset of the lane's 16-bit LeakyReLU' mask word, `mb` = that word of
// mask_bits, loaded by the caller with epi64_bits BEFORE it stores anything: memory operations retire in order, so a load
// issued behind a store waits for the store's round trip to memory.
template <bool LEAN>
__device__ __forceinline__ unsigned epi64_bits(const GGArgs& a, const EpiRes& R, unsigned boff) {
  if (LEAN || a.mask_bits) return __builtin_amdgcn_raw_buffer_load_b16(R.rbi, (LEAN && !a.mask_bits) ? DG_OOB_OFF : boff, 0, 0);
  return 0u;
}
// F >= 0: the activation / mask_bits / out_bits flags as compile-time bits 1 / 2 / 4 (straight-line code, picked once per
// epilogue by halo_epilogue: with run-time flags every pixel row carries ~20 uniform branches and the register moves of
// their merge points, and the epilogue of a 9..18-step tile was bound by instruction issue, not by its stores); F < 0: run-time.
template <typename T, bool LEAN, int F = -1>
__device__ __forceinline__ void epi64_pixel(const GGArgs& a, const EpiRes& R, const f32x4_t& f0, const f32x4_t& f1, const f32x4_t& f2,
                                            const f32x4_t& f3, const float (&bias)[16], unsigned offy, unsigned off1, unsigned off2,
                                            unsigned offm, unsigned boff, unsigned mb, bool mask_on, unsigned* ob_ret = nullptr) {
  typedef EpiV<T> IO;
  constexpr int NU = IO::NU, CPU = IO::CPU;
  u32x4_t v1[NU], v2[NU], vm[NU], va[NU];
  unsigned ob = 0;
  const bool f_act = F < 0 ? a.has_act != 0 : (F & 1) != 0;
  const bool f_mb = F < 0 ? a.mask_bits != nullptr : (F & 2) != 0;
  const bool f_ob = F < 0 ? a.out_bits != nullptr : (F & 4) != 0;
  const bool f_r1 = F < 0 ? !LEAN && a.r1 != nullptr : (F & 8) != 0;
  const bool f_r2 = F < 0 ? !LEAN && a.r2 != nullptr : (F & 16) != 0;
  // mask_on (run-time path): the caller's "this lane's channels are masked" (a.mask && channel >= a.mask_c0)
  const bool f_mk = F < 0 ? !LEAN && mask_on && !a.mask_last : (F & 32) != 0;       // mask before the accumulate
  const bool f_ml = F < 0 ? !LEAN && mask_on && a.mask_last != 0 : (F & 128) != 0;  // mask after it
  const bool f_ac = F < 0 ? !LEAN && a.accumulate != 0 : (F & 64) != 0;
  const bool f_q = F < 0 ? a.out_q != nullptr : (F & 256) != 0;                      // MXFP8 copy of the stored values
  u32x4_t pk[NU];
  // LEAN runs inside a tile loop whose memory operations must be unconditional (see gg_im2col_kernel): absent bit-mask
  // operands become out-of-range offsets (the load returns 0, the store is dropped)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (f_r1) v1[u] = __builtin_amdgcn_raw_buffer_load_b128(R.r1, off1, u * 16, 0);
    if (f_r2) v2[u] = __builtin_amdgcn_raw_buffer_load_b128(R.r2, off2, u * 16, 0);
    if (f_mk || f_ml) vm[u] = __builtin_amdgcn_raw_buffer_load_b128(R.rm, offm, u * 16, 0);
    if (f_ac) va[u] = __builtin_amdgcn_raw_buffer_load_b128(R.rY, offy, u * 16, 0);
  }
  float v[16];
  if (F >= 0 && (F & 512)) {          // the caller's MFMAs already added the bias (gg_im2col_direct_kernel)
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = f0[e]; v[4 + e] = f1[e]; v[8 + e] = f2[e]; v[12 + e] = f3[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = f0[e] + bias[e]; v[4 + e] = f1[e] + bias[4 + e]; v[8 + e] = f2[e] + bias[8 + e]; v[12 + e] = f3[e] + bias[12 + e]; }
  }
  // (pure-ALU parts may sit behind uniform branches: only the memory operations have to be unconditional)
  if (f_act) {
    if (a.act_slope >= 0.f && a.act_slope <= 1.f) {          // max(v, v * slope) == leaky(v) for slopes in [0, 1]: 2 operations, not 3
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = __builtin_fmaxf(v[k], v[k] * a.act_slope);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = leaky(v[k], a.act_slope);
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float r[CPU];
    float* vu = v + u * CPU;
    if (f_r1) {
      IO::unpack(v1[u], r);
#pragma unroll
      for (int e = 0; e < CPU; ++e) vu[e] = vu[e] * a.s1 + r[e];
    }
    if (f_r2) {
      IO::unpack(v2[u], r);
#pragma unroll
      for (int e = 0; e < CPU; ++e) vu[e] = vu[e] * a.s2 + r[e];
    }
    if (f_mk) {
      IO::unpack(vm[u], r);
#pragma unroll
      for (int e = 0; e < CPU; ++e) vu[e] *= leaky_grad(r[e], a.mask_slope);
    }
    if (f_mb) {
#pragma unroll
      for (int e = 0; e < CPU; ++e) vu[e] *= ((mb >> (u * CPU + e)) & 1u) ? 1.f : a.mask_slope;
    }
    if (f_ac) {
      IO::unpack(va[u], r);
#pragma unroll
      for (int e = 0; e < CPU; ++e) vu[e] += r[e];
    }
    if (f_ml) {
      IO::unpack(vm[u], r);
#pragma unroll
      for (int e = 0; e < CPU; ++e) vu[e] *= leaky_grad(r[e], a.mask_slope);
    }
    if (f_ob) {
#pragma unroll
      for (int e = 0; e < CPU; ++e) ob |= (vu[e] > 0.f ? 1u : 0u) << (u * CPU + e);
    }
    pk[u] = IO::pack(vu);
    __builtin_amdgcn_raw_buffer_store_b128(pk[u], R.rY, offy, u * 16, 0);
  }
  if (F >= 0 && (F & 1024)) *ob_ret = ob;      // the caller stores the word itself (gg_im2col_direct_kernel: two words per store)
  else if (F >= 0) { if (f_ob) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ob, R.rbo, boff, 0, 0); }
  else if (LEAN || a.out_bits) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ob, R.rbo, (LEAN && !a.out_bits) ? DG_OOB_OFF : boff, 0, 0);
  if constexpr (sizeof(T) == 2) {
    // MXFP8 copy of what was just stored (the bf16-ROUNDED values, so it equals dg_quant_mxfp8 of the stored tensor): the lane
    // holds 16 consecutive channels, lane ^ 16 the other half of the 32-channel block; one 16-byte store per lane, the scale byte
    // from the lane with the lower half.  q has y's pixel stride (in bytes = elements), scales [pixel][Nout / 32]:
    // byte offsets offy / 2 and boff / 4 ((rel * Nout + channel) / 32).
    if (f_q) {
      float w[16];
      IO::unpack(pk[0], w); IO::unpack(pk[1], w + 8);
      float amax = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) amax = __builtin_fmaxf(amax, __builtin_fabsf(w[k]));
      amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
      const int e = mx_scale_byte(amax);
      const u32x4_t qv = __builtin_bit_cast(u32x4_t, pack_fp8x16(w, mx_inv_scale(e)));
      __builtin_amdgcn_raw_buffer_store_b128(qv, R.rq, offy == DG_OOB_OFF ? DG_OOB_OFF : offy >> 1, 0, 0);
      const bool low_half = ((threadIdx.x >> 4) & 1) == 0;
      unsigned offqs = boff >> 2;                                   // dense scale rows: (rel * Nout + channel) / 32
      if (a.ldqs != (a.Nout >> 5)) {                                // a channel slice of a wider tensor (dense-block slab)
        const unsigned w16 = boff >> 1;                             // rel * (Nout / 16) + channel / 16, Nout / 16 a power of two
        offqs = (w16 >> a.qs_shift) * (unsigned)a.ldqs + ((w16 & ((1u << a.qs_shift) - 1u)) >> 1);
      }
      __builtin_amdgcn_raw_buffer_store_b8((unsigned char)e, R.rqs, (low_half && boff != DG_OOB_OFF) ? offqs : DG_OOB_OFF, 0, 0);
    }
  }
}

// ---- Epilogue of the row-tiled kernels (generic / fast / im2col): like halo_epilogue below, every tensor is
// addressed through a raw buffer descriptor based at the destination pixel of the workgroup's first row plus
// 32-bit per-lane offsets; rows past M / channels past Nout get an out-of-range offset (stores dropped).
// LEAN: bias / activation / bit masks only (no residual, activation-mask or accumulate operands): a third of the
// registers, for the store-bound kernels that need occupancy more than generality.
template <typename T, int BP, int BC, int WP, int WC, bool LEAN = false>
__device__ __forceinline__ void gg_epilogue(const GGArgs& a, f32x4_t (&acc)[WC / 16][WP / 16], int p0, int c0, int wp, int wc,
                                            int l15, int g) {
  typedef EpiIO<T> IO;
  typedef typename IO::V V;
  constexpr int FP = WP / 16, FC = WC / 16;
  constexpr int ES = (int)sizeof(T);
  auto dest_pixel = [&](int m) -> long long {
    const int gx = m % a.Wg, t = m / a.Wg;
    const int gy = t % a.Hg, n = t / a.Hg;
    const int py = a.dst_ps ? gy * 2 : gy * a.dy_mul + a.dy_off;
    const int px = a.dst_ps ? gx * 2 : gx * a.dx_mul + a.dx_off;
    return ((long long)n * a.Hd + py) * a.Wd + px;
  };
  const long long pb = dest_pixel(p0);                 // workgroup-uniform
  if constexpr (WC == 64) {
    // permuted 64-channel wave tile: the lane's four fragments are the 16 consecutive channels from cb16
    const int cb16 = c0 + wc * 64 + 16 * g;
    const bool cok = cb16 < a.Nout;
    float bias[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = (a.bias && cok) ? *reinterpret_cast<const float4*>(a.bias + cb16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
    }
    int cc0 = cb16, pj0 = 0;
    if (a.dst_ps) { const int q = cb16 / a.cps_dst; cc0 = cb16 - q * a.cps_dst; pj0 = (q >> 1) * a.Wd + (q & 1); }
    auto rsrc = [&](const void* p, long long ld, int es) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + pb * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
    };
    const int ldb = (a.Nout >> 6) * 4, bidx = ((c0 + wc * 64) >> 6) * 4 + g;
    EpiRes R;
    R.rY = rsrc(a.y, a.ldy, ES);
    R.r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1, ES); R.r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2, ES); R.rm = rsrc(a.mask ? a.mask : a.y, a.ldmask, ES);
    R.rbi = rsrc(a.mask_bits ? a.mask_bits : a.y, ldb, 2); R.rbo = rsrc(a.out_bits ? a.out_bits : a.y, ldb, 2);
    R.rq = rsrc(a.out_q ? a.out_q : a.y, a.ldy, 1); R.rqs = rsrc(a.out_qs ? a.out_qs : a.y, a.ldqs, 1);
    R.ldy = (int)a.ldy; R.ld1 = (int)a.ldr1; R.ld2 = (int)a.ldr2; R.ldm = (int)a.ldmask;
    int relv[FP];
    bool okv[FP];
    unsigned mbv[FP];
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const int m = p0 + wp * WP + 16 * i + l15;
      okv[i] = m < a.M && cok;
      relv[i] = okv[i] ? (int)(dest_pixel(m) - pb) : 0;
      mbv[i] = epi64_bits<LEAN>(a, R, okv[i] ? (unsigned)((relv[i] * ldb + bidx) * 2) : DG_OOB_OFF);
    }
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      const bool ok = okv[i];
      const int rel = relv[i], pix = rel + pj0;
      epi64_pixel<T, LEAN>(a, R, acc[0][i], acc[1][i], acc[2][i], acc[3][i], bias,
                           ok ? (unsigned)((pix * R.ldy + cc0) * ES) : DG_OOB_OFF, ok ? (unsigned)((pix * R.ld1 + cc0) * ES) : DG_OOB_OFF,
                           ok ? (unsigned)((pix * R.ld2 + cc0) * ES) : DG_OOB_OFF, ok ? (unsigned)((pix * R.ldm + cc0) * ES) : DG_OOB_OFF,
                           ok ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF, mbv[i], a.mask && cb16 >= a.mask_c0);
    }
    return;
  }
  int cc[FC], pj[FC];
  bool cok[FC];
  float4 bias[FC];
#pragma unroll
  for (int j = 0; j < FC; ++j) {
    const int cj = c0 + wc * WC + 16 * j + 4 * g;
    cok[j] = cj < a.Nout;
    bias[j] = (a.bias && cok[j]) ? *reinterpret_cast<const float4*>(a.bias + cj) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dst_ps) {
      const int q = cj / a.cps_dst;
      cc[j] = cj - q * a.cps_dst; pj[j] = (q >> 1) * a.Wd + (q & 1);
    } else { cc[j] = cj; pj[j] = 0; }
  }
  auto rsrc = [&](const void* p, long long ld) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + pb * ld * ES), 0, (int)DG_OOB_OFF, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rY = rsrc(a.y, a.ldy);
  const __amdgpu_buffer_rsrc_t r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1), r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2),
                               rm = rsrc(a.mask ? a.mask : a.y, a.ldmask);
  const int ldy = (int)a.ldy, ld1 = (int)a.ldr1, ld2 = (int)a.ldr2, ldm = (int)a.ldmask;
  // bit masks: one u16 per (pixel, 64-channel block, lane group g) holding bit 4j+e for channel 16j + 4g + e of the block
  const int ldb = (a.Nout >> 6) * 4, bidx = ((c0 + wc * WC) >> 6) * 4 + g;
  const bool bits_ok = WC == 64 && c0 + wc * WC < a.Nout;
  const __amdgpu_buffer_rsrc_t rbi = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(a.mask_bits ? a.mask_bits : a.y) + pb * ldb * 2), 0, (int)DG_OOB_OFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rbo = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<char*>(a.out_bits ? a.out_bits : a.y) + pb * ldb * 2), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
  for (int i = 0; i < FP; ++i) {
    const int m = p0 + wp * WP + 16 * i + l15;
    const bool pok = m < a.M;
    const int rel = pok ? (int)(dest_pixel(m) - pb) : 0;
    unsigned oyv[FC];
    V v1[FC], v2[FC], vm[FC], va[FC];
    const unsigned boff = (pok && bits_ok) ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF;
    unsigned mb = 0, ob = 0;
    if (a.mask_bits) mb = __builtin_amdgcn_raw_buffer_load_b16(rbi, boff, 0, 0);
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      const bool ok = pok && cok[j];
      const int pix = rel + pj[j];
      oyv[j] = ok ? (unsigned)((pix * ldy + cc[j]) * ES) : DG_OOB_OFF;
      if (!LEAN && a.r1) v1[j] = IO::load(r1, ok ? (unsigned)((pix * ld1 + cc[j]) * ES) : DG_OOB_OFF);
      if (!LEAN && a.r2) v2[j] = IO::load(r2, ok ? (unsigned)((pix * ld2 + cc[j]) * ES) : DG_OOB_OFF);
      if (!LEAN && a.mask) vm[j] = IO::load(rm, ok ? (unsigned)((pix * ldm + cc[j]) * ES) : DG_OOB_OFF);
      if (!LEAN && a.accumulate) va[j] = IO::load(rY, oyv[j]);
    }
    const int cj0 = c0 + wc * WC + 4 * g;           // fragment j covers channels cj0 + 16 j .. + 3 (mask_c0 is a multiple of 16)
#pragma unroll
    for (int j = 0; j < FC; ++j) {
      float v[4] = {acc[j][i][0] + bias[j].x, acc[j][i][1] + bias[j].y, acc[j][i][2] + bias[j].z, acc[j][i][3] + bias[j].w};
      float r[4];
      if (a.has_act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = leaky(v[e], a.act_slope);
      }
      if (!LEAN && a.r1) {
        IO::unpack(v1[j], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * a.s1 + r[e];
      }
      if (!LEAN && a.r2) {
        IO::unpack(v2[j], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * a.s2 + r[e];
      }
      const bool mask_j = !LEAN && a.mask && cj0 + 16 * j >= a.mask_c0;
      if (mask_j && !a.mask_last) {
        IO::unpack(vm[j], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(r[e], a.mask_slope);
      }
      if (a.mask_bits) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= ((mb >> (4 * j + e)) & 1u) ? 1.f : a.mask_slope;
      }
      if (!LEAN && a.accumulate) {
        IO::unpack(va[j], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      if (mask_j && a.mask_last) {
        IO::unpack(vm[j], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(r[e], a.mask_slope);
      }
      if (a.out_bits) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ob |= (v[e] > 0.f ? 1u : 0u) << (4 * j + e);
      }
      IO::store(v, rY, oyv[j]);
    }
    if (a.out_bits) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ob, rbo, boff, 0, 0);
  }
}

template <typename T, int BP, int BC, int WP, int WC>
__global__ __launch_bounds__(256, 2) void gg_kernel(const GGArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int NPW = BP / WP;
  constexpr int FP = WP / 16, FC = WC / 16;
  constexpr int PR = BP / 32;
  constexpr int CR = (BC + 31) / 32;
  constexpr int ROWS = BP + BC;
  static_assert((BP / WP) * (BC / WC) == 4, "4 waves per workgroup");
  __shared__ uint4 smem[2 * ROWS * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct, tile_p = tile / a.nct;
  const int p0 = tile_p * BP, c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w);

  int sy0[PR], sx0[PR], img[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    int m = p0 + r0 + 32 * i;
    if (m < a.M) {
      int gx = m % a.Wg, t = m / a.Wg;
      int gy = t % a.Hg;
      img[i] = t / a.Hg;
      sy0[i] = gy * a.sy_mul;
      sx0[i] = gx * a.sx_mul;
    } else {
      img[i] = -1; sy0[i] = 0; sx0[i] = 0;
    }
  }
  int tap = cc / a.cch, c8 = cc % a.cch;

  uint4 ra[PR], rb[CR];

  auto gload = [&]() {
    const bool kok = tap < a.ntaps;
    unsigned code = 0;
    if (kok) code = tap < 8 ? (unsigned)((a.tap_lo >> (8 * tap)) & 0xffull) : (a.tap_hi & 0xffu);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1, ws = (int)(code >> 4);
    int q = 0, cq = c8;
    if (a.src_ps) { q = c8 / a.cps_src_chunks; cq = c8 - q * a.cps_src_chunks; }
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int sy = sy0[i] + dy, sx = sx0[i] + dx;
      const bool ok = kok && img[i] >= 0 && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      long long off;
      if (!a.src_ps) off = ((long long)(img[i] * a.Hs + sy) * a.Ws + sx) * a.ldx + c8 * EPC;
      else off = (((long long)(img[i] * 2 * a.Hs + 2 * sy + (q >> 1))) * (2 * a.Ws) + 2 * sx + (q & 1)) * a.ldx + cq * EPC;
      // always-valid address + value select: keeps the staging registers out of scratch
      uint4 v = *reinterpret_cast<const uint4*>(X + (ok ? off : 0ll));
      ra[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int row = r0 + 32 * i, n = c0 + (WC == 64 ? perm64(row) : row);
      const bool ok = kok && row < BC && n < a.Nout;
      const long long woff = (long long)n * a.ldw + (long long)ws * a.Cred + c8 * EPC;
      uint4 v = *reinterpret_cast<const uint4*>(Wt + (ok ? woff : 0ll));
      rb[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
    c8 += 8;
    while (c8 >= a.cch) { c8 -= a.cch; ++tap; }
  };
  auto lstore = [&](int buf) {
    uint4* s = smem + buf * ROWS * 8;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int row = r0 + 32 * i;
      s[row * 8 + (cc ^ ((row >> 1) & 7))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int row = r0 + 32 * i;
      if (row < BC) s[(BP + row) * 8 + (cc ^ ((row >> 1) & 7))] = rb[i];
    }
  };

  f32x4_t acc[FC][FP];
#pragma unroll
  for (int j = 0; j < FC; ++j)
#pragma unroll
    for (int i = 0; i < FP; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int wp = wave % NPW, wc = wave / NPW;
  const int l15 = lane & 15, g = lane >> 4;
  const int nk = (a.kchunks + 7) >> 3;

  gload();
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) gload();
    const uint4* s = smem + cur * ROWS * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 4 + g;
      uint4 fa[FC], fb[FP];
#pragma unroll
      for (int j = 0; j < FC; ++j) {
        const int row = wc * WC + 16 * j + l15;
        fa[j] = s[(BP + row) * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int i = 0; i < FP; ++i) {
        const int row = wp * WP + 16 * i + l15;
        fb[i] = s[row * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < FC; ++j)
#pragma unroll
        for (int i = 0; i < FP; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  gg_epilogue<T, BP, BC, WP, WC>(a, acc, p0, c0, wp, wc, l15, g);
}

// ---------------------------------------------------------------------------------------------
// Fast path: every K-step (8 chunks) lies inside ONE tap (chunks-per-tap % 8 == 0, true for every
// wide layer: Cred >= 64 bf16 / 32 fp32).  Then tap, channel offset and the source-pixel shift are
// workgroup-uniform per K-step and live in SGPRs; each thread keeps constant 32-bit byte offsets for
// its rows and a 9-bit tap-validity mask.  Operands are fetched with raw buffer loads whose
// descriptor base is re-pointed per K-step (scalar adds); a padded / out-of-tile row simply gets an
// out-of-range offset and the hardware returns zeros - no per-row address arithmetic in the loop.

template <typename T, int BP, int BC, int WP, int WC>
__global__ __launch_bounds__(256, 2) void gg_fast_kernel(const GGArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int NPW = BP / WP;
  constexpr int FP = WP / 16, FC = WC / 16;
  constexpr int PR = BP / 32;
  constexpr int CR = (BC + 31) / 32;
  constexpr int ROWS = BP + BC;
  static_assert((BP / WP) * (BC / WC) == 4, "4 waves per workgroup");
  __shared__ uint4 smem[2 * ROWS * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct, tile_p = tile / a.nct;
  const int p0 = tile_p * BP, c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;
  const int Wsrc = a.src_ps ? 2 * a.Ws : a.Ws, Hsrc = a.src_ps ? 2 * a.Hs : a.Hs, psm = a.src_ps ? 2 : 1;

  // workgroup-uniform base pixel: first row of the tile minus a margin that covers every tap shift
  long long pbase;
  {
    const int gx = p0 % a.Wg, t = p0 / a.Wg;
    const int gy = t % a.Hg, im = t / a.Hg;
    pbase = ((long long)im * Hsrc + (long long)gy * a.sy_mul * psm) * Wsrc + (long long)gx * a.sx_mul * psm - (2 * Wsrc + 2);
    if (pbase < 0) pbase = 0;
  }
  unsigned rowoff[PR], vmask[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    const int m = p0 + r0 + 32 * i;
    rowoff[i] = 0; vmask[i] = 0;
    if (m < a.M) {
      const int gx = m % a.Wg, t = m / a.Wg;
      const int gy = t % a.Hg, im = t / a.Hg;
      const int sy0 = gy * a.sy_mul, sx0 = gx * a.sx_mul;
      const long long p = ((long long)im * Hsrc + (long long)sy0 * psm) * Wsrc + (long long)sx0 * psm;
      rowoff[i] = (unsigned)((p - pbase) * a.ldx * ES) + cc * 16;
      for (int t2 = 0; t2 < a.ntaps; ++t2) {
        const unsigned code = t2 < 8 ? (unsigned)((a.tap_lo >> (8 * t2)) & 0xffull) : (a.tap_hi & 0xffu);
        const int sy = sy0 + (int)(code & 3u) - 1, sx = sx0 + (int)((code >> 2) & 3u) - 1;
        if ((unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws) vmask[i] |= 1u << t2;
      }
    }
  }
  unsigned woff[CR];
#pragma unroll
  for (int i = 0; i < CR; ++i) {
    const int row = r0 + 32 * i, prow = WC == 64 ? perm64(row) : row;     // LDS row `row` holds output channel c0 + prow
    woff[i] = (row < BC && c0 + prow < a.Nout) ? (unsigned)((long long)prow * a.ldw * ES) + cc * 16 : DG_OOB_OFF;
  }
  const char* Xb = reinterpret_cast<const char*>(a.x) + pbase * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;

  u32x4_t ra[PR], rb[CR];
  int tap = 0, cbase = 0;          // workgroup-uniform K position (SGPRs)
  auto gload = [&]() {
    const unsigned code = tap < 8 ? (unsigned)((a.tap_lo >> (8 * tap)) & 0xffull) : (a.tap_hi & 0xffu);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1, ws = (int)(code >> 4);
    long long xo;
    if (!a.src_ps) xo = ((long long)dy * a.Ws + dx) * a.ldx + cbase * EPC;
    else {
      const int q = cbase / a.cps_src_chunks, cq = cbase - q * a.cps_src_chunks;
      xo = ((long long)(2 * dy + (q >> 1)) * Wsrc + (2 * dx + (q & 1))) * a.ldx + cq * EPC;
    }
    const long long wo = (long long)ws * a.Cred + cbase * EPC;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + xo * ES), 0, (int)DG_OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Wb + wo * ES), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const unsigned vo = ((vmask[i] >> tap) & 1u) ? rowoff[i] : DG_OOB_OFF;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[i], 0, 0);
    cbase += 8;
    if (cbase >= a.cch) { cbase = 0; ++tap; }
  };
  auto lstore = [&](int buf) {
    uint4* s = smem + buf * ROWS * 8;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int row = r0 + 32 * i;
      s[row * 8 + (cc ^ ((row >> 1) & 7))] = __builtin_bit_cast(uint4, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int row = r0 + 32 * i;
      if (row < BC) s[(BP + row) * 8 + (cc ^ ((row >> 1) & 7))] = __builtin_bit_cast(uint4, rb[i]);
    }
  };

  f32x4_t acc[FC][FP];
#pragma unroll
  for (int j = 0; j < FC; ++j)
#pragma unroll
    for (int i = 0; i < FP; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int wp = wave % NPW, wc = wave / NPW;
  const int l15 = lane & 15, g = lane >> 4;
  const int nk = a.kchunks >> 3;

  // write-after-barrier pipeline: the registers filled during step ks-1 are stored at the START of step ks and
  // refilled at once with the loads of step ks+2, so a load has a whole step to arrive
  gload();
  lstore(0);
  if (nk > 1) gload();
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) lstore(cur ^ 1);
    if (ks + 2 < nk) gload();
    const uint4* s = smem + cur * ROWS * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 4 + g;
      uint4 fa[FC], fb[FP];
#pragma unroll
      for (int j = 0; j < FC; ++j) {
        const int row = wc * WC + 16 * j + l15;
        fa[j] = s[(BP + row) * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int i = 0; i < FP; ++i) {
        const int row = wp * WP + 16 * i + l15;
        fb[i] = s[row * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < FC; ++j)
#pragma unroll
        for (int i = 0; i < FP; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
    }
    __syncthreads();
    cur ^= 1;
  }
  gg_epilogue<T, BP, BC, WP, WC>(a, acc, p0, c0, wp, wc, l15, g);
}

// ---- Epilogue of the halo kernels.  The lane's 16 fragments (pixel fragment i = tile row wp*4+i, channel
// fragment j) sit at (relative pixel of i + pixel-shuffle offset of j) * ld + channel: every tensor is addressed
// through a raw buffer descriptor based at the workgroup's first destination pixel plus 32-bit per-lane offsets
// (one multiply-add per fragment and tensor instead of 64-bit index arithmetic), and out-of-tile / out-of-range
// fragments get an out-of-range offset: the hardware drops those stores and returns zeros for those loads, so the
// epilogue has no divergent branches.
template <typename T, int NH>
__device__ __forceinline__ void halo_epilogue(const GGArgs& a, f32x4_t (&acc)[4 * NH][4], int img, int ty0, int tx0, int c0, int wp,
                                              int wc, int l15, int g) {
  constexpr int ES = (int)sizeof(T);
  const int psm = a.dst_ps ? 2 : a.dy_mul, psx = a.dst_ps ? 2 : a.dx_mul;
  const int oy = a.dst_ps ? 0 : a.dy_off, ox = a.dst_ps ? 0 : a.dx_off;
  // workgroup base pixel (scalar) and this lane's relative pixel for tile row wp*4 (+ i rows of pitch `rowp`)
  const long long pb = ((long long)img * a.Hd + (long long)ty0 * psm + oy) * a.Wd + (long long)tx0 * psx + ox;
  const int rel0 = (wp * 4) * psm * a.Wd + l15 * psx;
  const int rowp = psm * a.Wd;
  auto rsrc = [&](const void* p, long long ld, int es) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + pb * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
  };
  const int ldb = (a.Nout >> 6) * 4;
  EpiRes R;
  R.rY = rsrc(a.y, a.ldy, ES);
  R.r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1, ES); R.r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2, ES); R.rm = rsrc(a.mask ? a.mask : a.y, a.ldmask, ES);
  R.rbi = rsrc(a.mask_bits ? a.mask_bits : a.y, ldb, 2); R.rbo = rsrc(a.out_bits ? a.out_bits : a.y, ldb, 2);
  R.rq = rsrc(a.out_q ? a.out_q : a.y, a.ldy, 1); R.rqs = rsrc(a.out_qs ? a.out_qs : a.y, a.ldqs, 1);
  R.ldy = (int)a.ldy; R.ld1 = (int)a.ldr1; R.ld2 = (int)a.ldr2; R.ldm = (int)a.ldmask;
  const bool xok = tx0 + l15 < a.Wg;
  // every mask word of the wave's tile first (NH halves x 4 rows), before the first store
  unsigned mbv[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int cb64 = c0 + (wc + h) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = xok && cb64 + 16 * g < a.Nout && ty0 + wp * 4 + i < a.Hg;
      mbv[h][i] = epi64_bits<false>(a, R, ok ? (unsigned)(((rel0 + i * rowp) * ldb + (cb64 >> 6) * 4 + g) * 2) : DG_OOB_OFF);
    }
  }
  auto run = [&](auto tag, auto htag) {
  constexpr int F = decltype(tag)::value;
  constexpr int h = decltype(htag)::value;          // compile-time: acc[] must never be indexed dynamically
  {
    // permuted wave tile (perm64): the lane's four channel fragments are the 16 consecutive channels from cb16
    const int cb16 = c0 + (wc + h) * 64 + 16 * g;
    const bool cok = cb16 < a.Nout && xok;
    float bias[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = (a.bias && cb16 < a.Nout) ? *reinterpret_cast<const float4*>(a.bias + cb16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
    }
    int cc0 = cb16, pj0 = 0;
    if (a.dst_ps) { const int q = cb16 / a.cps_dst; cc0 = cb16 - q * a.cps_dst; pj0 = (q >> 1) * a.Wd + (q & 1); }
    const int bidx = ((c0 + (wc + h) * 64) >> 6) * 4 + g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cok && ty0 + wp * 4 + i < a.Hg;
      const int rel = rel0 + i * rowp;
      const int pix = rel + pj0;
      epi64_pixel<T, (F >= 0), F>(a, R, acc[4 * h][i], acc[4 * h + 1][i], acc[4 * h + 2][i], acc[4 * h + 3][i], bias,
                            ok ? (unsigned)((pix * R.ldy + cc0) * ES) : DG_OOB_OFF, ok ? (unsigned)((pix * R.ld1 + cc0) * ES) : DG_OOB_OFF,
                            ok ? (unsigned)((pix * R.ld2 + cc0) * ES) : DG_OOB_OFF, ok ? (unsigned)((pix * R.ldm + cc0) * ES) : DG_OOB_OFF,
                            ok ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF, mbv[h][i], a.mask && cb16 >= a.mask_c0);
    }
  }
  };
  // the flag combinations the train step launches most get straight-line instances; everything else the general one.
  // Decoded per 64-channel half: the activation mask may start at channel mask_c0 (a multiple of 64 here, else general path)
  const int key0 = (a.has_act ? 1 : 0) | (a.mask_bits ? 2 : 0) | (a.out_bits ? 4 : 0) | (a.r1 ? 8 : 0) | (a.r2 ? 16 : 0) | (a.accumulate ? 64 : 0) |
                   (a.out_q ? 256 : 0);
  auto dispatch = [&](auto htag) {
    constexpr int h = decltype(htag)::value;
    int key = key0;
    if (a.mask) key = (a.mask_c0 & 63) ? -1 : (c0 + (wc + h) * 64 >= a.mask_c0 ? key0 | (a.mask_last ? 128 : 32) : key0);
    switch (key) {
      case 0: run(std::integral_constant<int, 0>{}, htag); break;      // plain / bias only (data gradients without a mask)
      case 1: run(std::integral_constant<int, 1>{}, htag); break;      // bias + LeakyReLU (generator dense-block convs)
      case 2: run(std::integral_constant<int, 2>{}, htag); break;      // 1-bit mask (critic data gradients, penalty tangent forward)
      case 5: run(std::integral_constant<int, 5>{}, htag); break;      // LeakyReLU + out_bits (critic forward)
      case 8: run(std::integral_constant<int, 8>{}, htag); break;      // residual (generator dense-block output)
      case 24: run(std::integral_constant<int, 24>{}, htag); break;    // two residuals (RRDB output)
      case 32: run(std::integral_constant<int, 32>{}, htag); break;    // activation mask (data gradients of the narrow configs)
      case 64: run(std::integral_constant<int, 64>{}, htag); break;    // accumulate (dense-block data gradients)
      case 128: run(std::integral_constant<int, 128>{}, htag); break;  // mask of the completed top slice (dense block, conv 5's data gradient)
      case 192: run(std::integral_constant<int, 192>{}, htag); break;  // accumulate, then the completed slice's mask (convs 4..2)
      case 257: run(std::integral_constant<int, 257>{}, htag); break;  // fp8 mode: bias + LeakyReLU + MXFP8 copy (generator dense-block convs)
      case 264: run(std::integral_constant<int, 264>{}, htag); break;  // fp8 mode: residual + MXFP8 copy (dense-block output)
      case 280: run(std::integral_constant<int, 280>{}, htag); break;  // fp8 mode: two residuals + MXFP8 copy (RRDB output)
      case 258: run(std::integral_constant<int, 258>{}, htag); break;  // fp8 mode: 1-bit mask + MXFP8 copy (critic data gradients, tangent forward)
      case 261: run(std::integral_constant<int, 261>{}, htag); break;  // fp8 mode: LeakyReLU + out_bits + MXFP8 copy (critic forward)
      default: run(std::integral_constant<int, -1>{}, htag); break;
    }
  };
  dispatch(std::integral_constant<int, 0>{});
  if constexpr (NH == 2) dispatch(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------
// Halo path (unit-stride gathers: stride-1 forward, every data gradient; stride-2 forward through parity planes).  One
// workgroup = a 16x16 tile of the GEMM-row grid of ONE image x 128 output channels.  The 9 taps of a 3x3 stencil read
// overlapping source pixels, so instead of staging a [pixels][K-slice] operand per tap (9x the bytes) the workgroup keeps
// the (16+2)x(16+2) source PATCH of the current channel block in LDS and every tap reads its fragments from the patch at a
// shifted row; only the weights stream per tap-step.  gg_halo128_kernel (8 waves, 128-channel steps, one workgroup per CU)
// came first and stays behind DG_GG_NO4W; gg_halo4w_kernel (4 waves, 64-channel steps, two workgroups per CU) is the default.
// Halo kernel with 128-channel K-steps (16 chunks = 256-byte LDS rows).  In-kernel cycle stamps of the 64-channel
// version show ~700-900 cycles per tap-step that do not shrink with the MFMA work (LDS store + barrier skew between
// the two waves of a SIMD + scalar bookkeeping), against 1024 MFMA-pipe cycles: doubling the channels per step
// doubles the MFMA work those fixed costs are amortised over.  The patch is single-buffered (86 KB with row padding) and
// swapped at the channel-block boundary between two barriers; weights stream by LDS-DMA into two 32-KB slots.
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#ifdef DG_STAMP
// diagnostic build only (make stamp): per-wave cycle sums of the segments of a tap-step, blocks 0/1
__device__ unsigned long long g_stamps[2 * 8 * 8];
extern "C" int dg_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 1;
}
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(v) do { } while (0)
#endif
// S2 = stride-2 forward on the same machinery.  out(y,x) = sum_{r,s} in(2y+r-1, 2x+s-1) w(r,s): split the input into its four
// parity planes (py,px) = (row & 1, col & 1) of 2x2 blocks; tap r reads plane py = (r != 1) at block offset (r == 0 ? -1 : 0),
// so every tap is a UNIT-stride shift in block coordinates.  The K loop runs over (plane, channel block) pairs; the patch
// of a pair is the 17x17 blocks of that plane (gathered with stride 2 straight from the NHWC tensor, 256-byte rows), and
// the pair's taps are the 1 / 2 / 2 / 4 taps that read the plane (a.tap_* are grouped by plane by the launcher).
// MODE 2 (CT) = layers with ONE reduction-channel block (Cin = 128) and several output-channel tiles: a 9-step tile pays its
// prologue (exposed patch latency) and epilogue for only nine tap-steps (~800 TFLOP/s against ~1000 at 18 steps), and the
// channel tiles of a pixel tile all read the same patch.  One workgroup therefore walks ALL channel tiles of its pixel tile:
// the patch is loaded once, the weight stream continues across the tiles (tile-major tap-steps) and every tile ends with
// its own epilogue and a cleared accumulator.
template <typename T, int MODE>
__global__ __launch_bounds__(512) void gg_halo128_kernel(const GGArgs a, int tiles_x, int tiles_y) {
  constexpr bool S2 = MODE == 1, CT = MODE == 2;
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int BC = 128, KC = 16;                                       // 16 chunks per row
  constexpr int NPL = (PROWS * KC + 511) / 512;                          // 11 patch chunks per thread
  constexpr int NWL = BC * KC / 512;                                     // 4 weight chunks per thread
  // LDS rows are 256 B of data + 16 B of padding: 16 consecutive rows then start 4 banks apart, so a fragment read
  // (16 lanes = 16 consecutive rows at one chunk) is conflict-free at ANY row offset without an XOR swizzle, and every
  // per-k-chunk / per-row address is base + immediate (no per-read VALU address arithmetic).
  constexpr int PITCH = KC * 16 + 16;            // 272 bytes
  extern __shared__ __attribute__((aligned(16))) char dsm128[];
  char* const s_patch = dsm128;                  // [PROWS][PITCH]
  // weights: [2][BC][256 B], no padding (written by LDS-DMA, which fills 1 KB contiguously per wave-instruction); the
  // 16-byte chunk c of row r sits at chunk position c ^ (r & 15) -- conflict-free for the 16-row-aligned fragment reads
  char* const s_w = dsm128 + PROWS * PITCH;      // [2][BC][KC * 16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = CT ? 0 : tile % a.nct;
  unsigned rest = CT ? tile : tile / a.nct;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  const int ty0 = (rest % tiles_y) * TH;
  const int img = rest / tiles_y;
  const int c0 = tile_c * BC;
  const int cc = tid & 15, r0 = tid >> 4;        // r0 in [0,32)

  const int sy_base = S2 ? (ty0 > 0 ? 2 * (ty0 - 1) : 0) : (ty0 - 1 > 0 ? ty0 - 1 : 0);
  unsigned woff[NWL];                            // lane's source offset for DMA piece i: row r0+32i, chunk cc ^ (r0 & 15)
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    int row = perm64(r0 + 32 * i);                   // LDS row r0+32i holds output channel c0 + perm64(row)
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;   // rows past Cout: any valid row (their outputs are never stored)
    woff[i] = (unsigned)((long long)row * a.ldw * ES) + ((cc ^ (r0 & 15)) * 16);
  }
  const char* Xb = reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;

  const int ncbr = a.cch / KC;                   // channel blocks of 128 (64 for fp32) reduction channels
  // (plane, channel block) pairs are numbered plane-major; planes hold 1, 2, 2, 4 taps starting at tap 0, 1, 3, 5
  auto plane_of = [&](int vcb) { return S2 ? (int)(vcb >= ncbr) + (int)(vcb >= 2 * ncbr) + (int)(vcb >= 3 * ncbr) : 0; };
  auto ntaps_of = [&](int vcb) { return S2 ? (0x4221 >> (4 * plane_of(vcb))) & 15 : a.ntaps; };   // (CT: every tile has all taps)
  auto tap_code = [&](int vcb, int tap) {
    const int gt = S2 ? ((0x5310 >> (4 * plane_of(vcb))) & 15) + tap : tap;
    return gt < 8 ? (unsigned)((a.tap_lo >> (8 * gt)) & 0xffull) : (a.tap_hi & 0xffu);
  };
  u32x4_t rp[NPL];
  auto load_patch = [&](int vcb) {
    const int plane = plane_of(vcb), cb = vcb - plane * ncbr;
    const int ppy = plane >> 1, ppx = plane & 1;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + (long long)cb * KC * EPC * ES), 0, (int)DG_OOB_OFF, 0x00020000);
    // offsets are recomputed per channel block (once per ntaps steps) instead of living in 11 registers; the empty
    // asm keeps the compiler from hoisting them back out of the step loop
    int r0v = r0;
    asm volatile("" : "+v"(r0v));
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0v + 32 * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      const unsigned off = ok ? (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16 : DG_OOB_OFF;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // one offset live at a time: the kernel sits at the 256-VGPR limit
    }
  };
  char* const st_base = dsm128 + r0 * PITCH + cc * 16;
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + 32 * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(st_base + i * 32 * PITCH) = __builtin_bit_cast(uint4, rp[i]);
    }
  };
  // LDS-DMA of one 128x128 weight tile: 4 pieces of 1 KB per wave.  Issued as inline asm: for the builtin the compiler
  // assumes the DMA may alias every later LDS read and drains vmcnt(0) in front of the fragment prefetch, which serialises
  // the fetch with the MFMA blocks it should hide under.  (Its own vmcnt bookkeeping stays safe: unknown extra loads only
  // make counted waits stricter.)  m0 is not otherwise used by this kernel.
  i32x4_t w_rs;
  int w_dst0 = 0;
  auto dma_setup = [&](int vcb, int tap, int slot) {
    const unsigned code = tap_code(vcb, tap);
    const long long wo = CT ? (long long)(code >> 4) * a.Cred + (long long)vcb * BC * a.ldw
                            : (long long)(code >> 4) * a.Cred + (vcb - plane_of(vcb) * ncbr) * KC * EPC;
    const unsigned long long wbase = (unsigned long long)(Wb + wo * ES);
    w_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wbase);
    w_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32) & 0xffff);
    w_rs[2] = (int)DG_OOB_OFF;
    w_rs[3] = 0x00020000;
    w_dst0 = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_w + slot * (BC * KC * 16) + (wave * 4) * (KC * 16))));
  };
  auto dma_piece = [&](int i) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(w_dst0 + i * 32 * KC * 16), "v"(woff[i]), "s"(w_rs) : "memory");
  };
  auto dma_w = [&](int cb, int tap, int slot) {
    dma_setup(cb, tap, slot);
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma_piece(i);
  };
  auto barrier_all = [&]() {        // LDS-DMA is tracked by vmcnt: drain it before the barrier publishes the tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  auto barrier_keep_patch = [&]() { // the 11 patch loads issued after the DMA pieces may stay in flight
    asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    __syncthreads();
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int wp = wave & 3, wc = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncb = S2 ? 4 * ncbr : (CT ? (int)a.nct : ncbr);        // CT: ncbr == 1, vcb = output-channel tile
  const int nsteps = (CT ? (int)a.nct : ncbr) * a.ntaps;

  const char* fa_k[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fa_k[kk] = s_w + (wc * 64 + l15) * (KC * 16) + (((kk * 4 + g) ^ l15) * 16);
  const char* const fb_lane = s_patch + l15 * PITCH + g * 16;
  auto read_frags = [&](uint4 (&fa)[4], uint4 (&fb)[4], int pa, const char* pb, int kk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const uint4*>(fa_k[kk] + pa + j * 16 * KC * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + kk * 64);
  };
  auto mma_row = [&](const uint4 (&fa)[4], const uint4 (&fb)[4], int j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
  };
  auto mma_block = [&](const uint4 (&fa)[4], const uint4 (&fb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
  };

  // Schedule: ONE barrier per tap-step, in the MIDDLE of the step.  A step's 64 MFMAs per wave are four blocks of 16
  // (k-chunks 0..3).  Fragments of blocks 0/1 are prefetched at the end of the previous step, so MFMAs are ready to issue
  // on both sides of the barrier (in-kernel stamps of the barrier-at-step-end version: ~1000 cycles without MFMAs after
  // each barrier while 8 waves queue LDS stores + 16 fragment reads, and the two waves of a SIMD ran one after the other).
  //   step s:  mma(blk0) | read blk2 | mma(blk1) | read blk3 | BARRIER (all reads of weight slot s&1 done)
  //            LDS-DMA W[s+2] -> slot s&1 | mma(blk2) | read blk0 of step s+1 | mma(blk3) | read blk1 of s+1
  // W[s+1] was DMA'd into slot (s+1)&1 after the barrier of step s-1 and published by the barrier of step s.  At a channel-block boundary the single-buffered patch is
  // rewritten after the barrier (every wave holds its last fragments in registers) and a second barrier publishes it.
  auto adv = [&](int& c_, int& t_) { if (++t_ == ntaps_of(c_)) { t_ = 0; ++c_; } };
  auto patch_ptr = [&](int vcb_, int tap_) -> const char* {
    const unsigned code = tap_code(vcb_, tap_);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
    return fb_lane + ((wp * 4 + 1 + dy) * PW + 1 + dx) * PITCH;
  };
  load_patch(0);
  int cb = 0, tap = 0;
  int cbw = 0, tapw = 0;             // position of the next weight tile to fetch
  dma_w(0, 0, 0);
  adv(cbw, tapw);
  if (nsteps > 1) dma_w(cbw, tapw, 1);
  adv(cbw, tapw);                    // -> W[2]
  store_patch();
  barrier_all();

  uint4 fa0[4], fb0[4], fa1[4], fb1[4];
  const char* pb = patch_ptr(0, 0);
  int pa = 0;                        // byte offset of the weight slot being read
  read_frags(fa0, fb0, pa, pb, 0);
  read_frags(fa1, fb1, pa, pb, 1);
#ifdef DG_STAMP
  unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sAB = 0, sBC = 0, sCD = 0, sDE = 0, tL0 = 0;
  STAMP(tL0);
#endif
  for (int s = 0; s < nsteps; ++s) {
    STAMP(tA);
    const bool more = s + 1 < nsteps;
    int ntap = tap + 1, ncbn = cb;
    const int ntaps_cb = ntaps_of(cb);
    if (ntap == ntaps_cb) { ntap = 0; ncbn = cb + 1; }
    const bool swap = !CT && ntap == 0 && more;
    // a one-tap block (plane (0,0) of a stride-2 launch) has no earlier step of its own to fetch the next patch in:
    // fetched here and stored after this step's barrier (latency exposed, 1 step in 9)
    if (S2 && ntaps_cb == 1 && cb + 1 < ncb) load_patch(cb + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_block(fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fa0, fb0, pa, pb, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_block(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fa1, fb1, pa, pb, 3);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(tB);
    if (!CT && tap == 1 && cb + 1 < ncb && ntaps_cb > 1) barrier_keep_patch(); else barrier_all();
    STAMP(tC);
    const bool fetch = s + 2 < nsteps;
    if (fetch) dma_setup(cbw, tapw, s & 1);
    adv(cbw, tapw);
    pa = ((s + 1) & 1) * (BC * KC * 16);
    pb = patch_ptr(ncbn < ncb ? ncbn : 0, ntap);
    if (swap) store_patch();
    // the MFMA blocks stay in straight-line code (no accumulator phis): only the small side operations are conditional.
    // One DMA piece goes behind each row of MFMAs, so its issue cost hides in the gaps between this wave's MFMAs.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      mma_row(fa0, fb0, j);
      __builtin_amdgcn_sched_barrier(0);
      if (fetch) dma_piece(j);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!CT && tap == 0 && cb + 1 < ncb && ntaps_cb > 1) load_patch(cb + 1);   // a whole channel block ahead; kept out of the next vmcnt wait
    __builtin_amdgcn_sched_barrier(0);
    if (swap) barrier_all();                 // publish the new patch before the next step's fragments are read
    read_frags(fa0, fb0, pa, pb, 0);         // (after the last step: a harmless read of valid LDS)
    __builtin_amdgcn_sched_barrier(0);
    mma_block(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fa1, fb1, pa, pb, 1);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(tD);
#ifdef DG_STAMP
    tE = tD;
    sAB += tB - tA; sBC += tC - tB; sCD += tD - tC; sDE += tE - tD;
#endif
    if constexpr (CT) {
      if (ntap == 0) {                         // last tap of an output-channel tile: its outputs are complete
        halo_epilogue<T, 1>(a, acc, img, ty0, tx0, cb * BC, wp, wc, l15, g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    tap = ntap; cb = ncbn;
  }
#ifdef DG_STAMP
  unsigned long long tL1, tX;
  STAMP(tL1);
#endif

  // (16-byte stores come from the perm64 channel order, not from staging the tile through LDS: see epi64_pixel)
  if constexpr (!CT) halo_epilogue<T, 1>(a, acc, img, ty0, tx0, c0, wp, wc, l15, g);
#ifdef DG_STAMP
  STAMP(tX);
  if (blockIdx.x < 2 && lane == 0) {
    unsigned long long* o = g_stamps + (blockIdx.x * 8 + wave) * 8;
    o[0] = sAB; o[1] = sBC; o[2] = sCD; o[3] = sDE; o[4] = (unsigned long long)nsteps; o[5] = tL1 - tL0; o[6] = tX - tL1; o[7] = tL0;
  }
#endif
}

template <typename T, int MODE>
static int gg_launch_halo128(GGArgs& a, int N, hipStream_t st) {
  constexpr int LDS_BYTES = 324 * 272 + 2 * 128 * 256;
  DG_SET_MAX_LDS_ONCE((&gg_halo128_kernel<T, MODE>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = (unsigned)((a.Nout + 127) / 128);
  a.nwg = (MODE == 2 ? 1u : a.nct) * (unsigned)(tiles_x * tiles_y * N);
  g_last_kinds |= 8;
  hipLaunchKernelGGL((gg_halo128_kernel<T, MODE>), dim3(a.nwg), dim3(512), LDS_BYTES, st, a, tiles_x, tiles_y);
  return dg_check_launch();
}

// Stride-2 forward on the halo kernel: regroup the nine taps (source offsets dy,dx in {-1,0,1} around (2y,2x)) by the parity
// plane they read, plane-major order (0,0) (0,1) (1,0) (1,1) = 1 + 2 + 2 + 4 taps, and re-express each as a block offset
// in {-1, 0}.  Returns false (caller falls back to the per-tap kernel) unless the taps are exactly the 3x3 stencil.
static bool regroup_taps_by_plane(GGArgs& a) {
  if (a.ntaps != 9) return false;
  unsigned codes[9], out[9];
  bool seen[3][3] = {};
  for (int t = 0; t < 9; ++t) {
    codes[t] = t < 8 ? (unsigned)((a.tap_lo >> (8 * t)) & 0xffull) : (a.tap_hi & 0xffu);
    const int dy = (int)(codes[t] & 3u) - 1, dx = (int)((codes[t] >> 2) & 3u) - 1;
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || seen[dy + 1][dx + 1]) return false;
    seen[dy + 1][dx + 1] = true;
  }
  int n = 0;
  for (int plane = 0; plane < 4; ++plane)
    for (int t = 0; t < 9; ++t) {
      const int dy = (int)(codes[t] & 3u) - 1, dx = (int)((codes[t] >> 2) & 3u) - 1;
      if ((dy != 0) * 2 + (dx != 0) != plane) continue;
      const unsigned by = dy == -1 ? 0u : 1u, bx = dx == -1 ? 0u : 1u;      // block offset + 1
      out[n++] = by | (bx << 2) | (codes[t] & 0xf0u);
    }
  a.tap_lo = 0; a.tap_hi = 0;
  for (int t = 0; t < 9; ++t) {
    if (t < 8) a.tap_lo |= (unsigned long long)out[t] << (8 * t); else a.tap_hi = out[t];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Four-wave halo kernel, TWO workgroups per CU.  Same tile as gg_halo128_kernel (16x16 pixels x 128 output channels) but
// 64-channel K-steps, so patch (324 x 144 B) + two 16-KB weight slots = 78 KB and a second, independent workgroup shares
// the CU: its tap-step loop runs while this one sits in its prologue (exposed patch latency) or in its store-bound
// epilogue (then ~14k cycles per tile; 7-11k since the 16-byte, flag-specialised epilogue), which is where 9..18-step tiles
// lose 30-50 % of their time; and
// the two waves of a SIMD now belong to different workgroups (no shared barrier, no lock-step).  Each wave owns 4 tile
// rows x all 128 channels (8 x 4 accumulator fragments, 3 LDS fragment reads per 8 MFMAs instead of 4).
// One barrier per step, at its top:  BARRIER | DMA W[s+2] -> slot s&1 | mma(k0) | read k0 of s+1 | mma(k1) | read k1 of s+1.
// S2: stride-2 forward over the four parity planes of the input (see gg_halo128_kernel): K loop over (plane, channel block)
// pairs with 1 / 2 / 2 / 4 taps, the plane's patch gathered with stride 2.
// PS: pixel-shuffled source (data gradient of an up-sampling conv): virtual pixel (y, x), channel quarter q = stored pixel
// (2y + (q >> 1), 2x + (q & 1)); a 64-channel block lies inside one quarter, so its patch is a stride-2 gather like S2's.
// NW = 8: the same kernel with EIGHT waves and a 16x16-pixel x 256-channel tile (waves 0-3 the first 128 channels, waves 4-7
// the second), one workgroup per CU.  Both channel halves read ONE patch, so the patch bytes per flop halve -- which is what
// bounds the stride-2 forward (a stride-2 tile reads 4x the input pixels of a stride-1 tile: 227 flop per patch byte at 128
// channels, and the per-CU global->LDS path sustains only ~10-12 B/clk) -- at the price of the second, independent workgroup.
template <typename T, bool S2, bool PS = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void gg_halo4w_kernel(const GGArgs a, int tiles_x, int tiles_y) {
  constexpr int NT = 64 * NW, RPP = NT / 8;                              // threads; patch rows staged per pass
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int BC = 32 * NW, KC = 8;                                    // 8 chunks per row: 64 bf16 / 32 fp32 channels
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B patch rows
  constexpr int WROW = KC * 16;                                          // 128 B weight rows, chunk c of row r at c ^ ((r >> 1) & 7)
  constexpr int NPL = (PROWS * KC + NT - 1) / NT;                        // 11 (6) patch chunks per thread
  constexpr int NWL = BC * KC / NT;                                      // 4 weight pieces per wave and step
  extern __shared__ __attribute__((aligned(16))) char dsm4w[];
  char* const s_patch = dsm4w;                    // [PROWS][PITCH]
  char* const s_w = dsm4w + PROWS * PITCH;        // [2][BC][WROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct;
  unsigned rest = tile / a.nct;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  const int ty0 = (rest % tiles_y) * TH;
  const int img = rest / tiles_y;
  const int c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0, RPP)
  const int wq = wave & 3, wh = wave >> 2;        // tile rows 4*wq.., channel half wh (0 unless NW = 8)
  const int sy_base = S2 ? (ty0 > 0 ? 2 * (ty0 - 1) : 0) : (ty0 - 1 > 0 ? ty0 - 1 : 0);
  const char* Xb = PS ? reinterpret_cast<const char*>(a.x) + ((long long)img * 2 * a.Hs + 2 * sy_base) * (2 * a.Ws) * a.ldx * ES
                      : reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncbr = a.cch / KC;                    // real channel blocks
  const int ncb = S2 ? 4 * ncbr : ncbr;           // (plane, channel block) pairs, plane-major
  const int nsteps = ncbr * a.ntaps;
  auto plane_of = [&](int vcb) { return S2 ? (int)(vcb >= ncbr) + (int)(vcb >= 2 * ncbr) + (int)(vcb >= 3 * ncbr) : 0; };
  auto ntaps_of = [&](int vcb) { return S2 ? (0x4221 >> (4 * plane_of(vcb))) & 15 : a.ntaps; };

  unsigned woff[NWL];                             // DMA piece i of this wave: rows wave*32 + 8i .. +7, lane = (row, physical chunk)
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    int row = wave * 32 + i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    row = perm64(row);                              // LDS row holds output channel c0 + perm64(row)
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;
    woff[i] = (unsigned)((long long)row * a.ldw * ES) + logical * 16;
  }
  auto tap_code = [&](int vcb, int tap) {
    const int gt = S2 ? ((0x5310 >> (4 * plane_of(vcb))) & 15) + tap : tap;
    return gt < 8 ? (unsigned)((a.tap_lo >> (8 * gt)) & 0xffull) : (a.tap_hi & 0xffu);
  };
  u32x4_t rp[NPL];
  auto load_patch = [&](int vcb) {
    const int plane = plane_of(vcb), cb = vcb - plane * ncbr;
    const int ppy = plane >> 1, ppx = plane & 1;
    const int psq = PS ? (cb * KC) / a.cps_src_chunks : 0;                 // channel quarter of this block
    const int psy = psq >> 1, psx = psq & 1;
    const long long cboff = PS ? (long long)(cb * KC - psq * a.cps_src_chunks) * EPC * ES : (long long)cb * KC * EPC * ES;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + cboff), 0, (int)DG_OOB_OFF, 0x00020000);
    int r0v = r0;
    asm volatile("" : "+v"(r0v));
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0v + RPP * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      const unsigned off = !ok ? DG_OOB_OFF
                           : PS ? (unsigned)(((2 * (sy - sy_base) + psy) * (2 * a.Ws) + 2 * sx + psx) * a.ldx * ES) + cc * 16
                                : (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  char* const st_base = s_patch + r0 * PITCH + cc * 16;
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + RPP * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(st_base + i * RPP * PITCH) = __builtin_bit_cast(uint4, rp[i]);
    }
  };
  typedef int i32x4h_t __attribute__((ext_vector_type(4)));
  i32x4h_t w_rs;
  int w_dst0 = 0;
  auto dma_setup = [&](int vcb, int tap, int slot) {
    const unsigned code = tap_code(vcb, tap);
    const long long wo = (long long)(code >> 4) * a.Cred + (vcb - plane_of(vcb) * ncbr) * KC * EPC;
    const unsigned long long wbase = (unsigned long long)(Wb + wo * ES);
    w_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wbase);
    w_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32) & 0xffff);
    w_rs[2] = (int)DG_OOB_OFF;
    w_rs[3] = 0x00020000;
    w_dst0 = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_w + slot * (BC * WROW) + (wave * 32) * WROW)));
  };
  auto dma_piece = [&](int i) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(w_dst0 + i * 8 * WROW), "v"(woff[i]), "s"(w_rs) : "memory");
  };
  auto dma_w = [&](int cb, int tap, int slot) {
    dma_setup(cb, tap, slot);
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma_piece(i);
  };
  auto barrier_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };
  auto barrier_keep_patch = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPL) : "memory"); __syncthreads(); };

  f32x4_t acc[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const char* fa_k[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fa_k[kk] = s_w + (wh * 128 + l15) * WROW + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) * 16);
  const char* const fb_lane = s_patch + l15 * PITCH + g * 16;
  auto read_frags = [&](uint4 (&fa)[8], uint4 (&fb)[4], int pa, const char* pb, int kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = *reinterpret_cast<const uint4*>(fa_k[kk] + pa + j * 16 * WROW);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + kk * 64);
  };
  auto mma_rows = [&](const uint4 (&fa)[8], const uint4 (&fb)[4], int j0) {     // two weight fragments x four pixel rows
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
  };
  auto adv = [&](int& c_, int& t_) { if (++t_ == ntaps_of(c_)) { t_ = 0; ++c_; } };
  auto patch_ptr = [&](int vcb_, int tap_) -> const char* {
    const unsigned code = tap_code(vcb_, tap_);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
    return fb_lane + ((wq * 4 + 1 + dy) * PW + 1 + dx) * PITCH;
  };

#ifdef DG_STAMP
  unsigned long long tK0 = 0, tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sAB = 0, sBC = 0, sCD = 0, sDE = 0, tL0 = 0;
  STAMP(tK0);
#endif
  load_patch(0);
  int cb = 0, tap = 0, cbw = 0, tapw = 0;
  dma_w(0, 0, 0);
  adv(cbw, tapw);
  if (nsteps > 1) dma_w(cbw, tapw, 1);
  adv(cbw, tapw);                    // -> W[2]
  store_patch();
  barrier_all();
  // ONE fragment set (accumulators 128 + fragments 48 + patch staging 44 registers): a wave waits for its LDS reads in
  // the open, which is what the second workgroup on the CU is there to cover.
  uint4 fa[8], fb[4];
  const char* pb = patch_ptr(0, 0);
  int pa = 0;
  read_frags(fa, fb, pa, pb, 0);
  STAMP(tL0);
  for (int s = 0; s < nsteps; ++s) {
    STAMP(tA);
    const bool more = s + 1 < nsteps;
    int ntap = tap + 1, ncbn = cb;
    const int ntaps_cb = ntaps_of(cb);
    if (ntap == ntaps_cb) { ntap = 0; ncbn = cb + 1; }
    const bool swap = ntap == 0 && more;
    const bool patch_now = cb + 1 < ncb && (ntaps_cb == 1 || tap == 0);     // fetch the next block's patch during its predecessor's first step
    const bool fetch = s + 2 < nsteps;
    // k-block 0 (fragments read at the end of the previous step), then k-block 1: after it every wave has read all it
    // needs of this step, so the barrier below frees slot s&1 (and, at a block end, the patch); W[s+1] has landed by then
    // k-block 0, and the weight fragments of k-block 1 re-read row pair by row pair right behind the MFMAs that consumed their
    // k-block-0 contents (that latency hides under the remaining rows; only the four pixel fragments are read in the open at
    // the end): +0.5-1.5 % per launch against reading all twelve fragments after the block (the chip gives about half of a
    // cycle saving back as clock).  The asm fences keep each read in its slot and each MFMA pair in front of it.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mma_rows(fa, fb, 2 * q);
      asm volatile("" : "+v"(acc[2 * q][0]), "+v"(acc[2 * q][1]), "+v"(acc[2 * q][2]), "+v"(acc[2 * q][3]),
                        "+v"(acc[2 * q + 1][0]), "+v"(acc[2 * q + 1][1]), "+v"(acc[2 * q + 1][2]), "+v"(acc[2 * q + 1][3]) :: "memory");
      __builtin_amdgcn_sched_barrier(0);
      fa[2 * q] = *reinterpret_cast<const uint4*>(fa_k[1] + pa + (2 * q) * 16 * WROW);
      fa[2 * q + 1] = *reinterpret_cast<const uint4*>(fa_k[1] + pa + (2 * q + 1) * 16 * WROW);
      asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + 64);
    __builtin_amdgcn_sched_barrier(0);
    if (patch_now) load_patch(cb + 1);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(tB);
#pragma unroll
    for (int q = 0; q < 4; ++q) mma_rows(fa, fb, 2 * q);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(tC);
    if (patch_now && !swap) barrier_keep_patch(); else barrier_all();
    STAMP(tD);
    pa = ((s + 1) & 1) * (BC * WROW);
    pb = patch_ptr(ncbn < ncb ? ncbn : 0, ntap);
    if (swap) {                      // channel-block boundary: the single-buffered patch is rewritten, then published
      store_patch();
      barrier_all();
    }
    // next step's first fragments go out before the DMA pieces: the ~700 cycles a wave spends issuing those then cover
    // the LDS read latency instead of preceding it
    if (more) read_frags(fa, fb, pa, pb, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (fetch) { dma_setup(cbw, tapw, s & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_piece(q);
    }
    adv(cbw, tapw);
    __builtin_amdgcn_sched_barrier(0);
    tap = ntap; cb = ncbn;
    STAMP(tE);
#ifdef DG_STAMP
    sAB += tB - tA; sBC += tC - tB; sCD += tD - tC; sDE += tE - tD;
#endif
  }
#ifdef DG_STAMP
  unsigned long long tL1, tX;
  STAMP(tL1);
#endif
  halo_epilogue<T, 2>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g);
#ifdef DG_STAMP
  STAMP(tX);
  if (blockIdx.x < 2 && lane == 0) {
    unsigned long long* o = g_stamps + (blockIdx.x * 8 + wave) * 8;
    o[0] = sAB; o[1] = sBC; o[2] = sCD; o[3] = sDE; o[4] = (unsigned long long)nsteps; o[5] = tL1 - tL0; o[6] = tX - tL1; o[7] = tL0 - tK0;
  }
#endif
}

template <typename T, bool S2, bool PS = false, int NW = 4>
static int gg_launch_halo4w(GGArgs& a, int N, hipStream_t st) {
  constexpr int BC = 32 * NW;
  constexpr int LDS_BYTES = 324 * 144 + 2 * BC * 128;
  DG_SET_MAX_LDS_ONCE((&gg_halo4w_kernel<T, S2, PS, NW>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = (unsigned)((a.Nout + BC - 1) / BC);
  a.nwg = a.nct * (unsigned)(tiles_x * tiles_y * N);
  g_last_kinds |= 8;
  hipLaunchKernelGGL((gg_halo4w_kernel<T, S2, PS, NW>), dim3(a.nwg), dim3(64 * NW), LDS_BYTES, st, a, tiles_x, tiles_y);
  return dg_check_launch();
}

#ifdef DG_LC_EXPERIMENT
// EXPERIMENT (-DDG_LC_EXPERIMENT, DG_GG_LC=1): the four-wave halo kernel's tile with EIGHT waves in one workgroup per CU --
// waves 0-3 COMPUTE (fragment reads, MFMAs, epilogue), waves 4-7 LOAD (weight DMA, patch loads / LDS stores): can one compute
// wave per SIMD, freed of every vector-memory instruction in the loop, feed the matrix pipe as well as two waves that do both?
__global__ __launch_bounds__(512, 1) void gg_halo_lc_kernel(const GGArgs a, int tiles_x, int tiles_y) {
  typedef bf16_t T;
  constexpr bool S2 = false, PS = false;
  constexpr int NW = 4;
  constexpr int NT = 256, RPP = NT / 8;                                  // LOADER threads; patch rows staged per pass                              // threads; patch rows staged per pass
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int BC = 32 * NW, KC = 8;                                    // 8 chunks per row: 64 bf16 / 32 fp32 channels
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B patch rows
  constexpr int WROW = KC * 16;                                          // 128 B weight rows, chunk c of row r at c ^ ((r >> 1) & 7)
  constexpr int NPL = (PROWS * KC + NT - 1) / NT;                        // 11 (6) patch chunks per thread
  constexpr int NWL = BC * KC / NT;                                      // 4 weight pieces per wave and step
  extern __shared__ __attribute__((aligned(16))) char dsm4w[];
  char* const s_patch = dsm4w;                    // [PROWS][PITCH]
  char* const s_w = dsm4w + PROWS * PITCH;        // [2][BC][WROW]

  const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool compute = wave8 < 4;
  const int wave = wave8 & 3, tid = threadIdx.x & 255;      // index inside the role's four waves
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct;
  unsigned rest = tile / a.nct;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  const int ty0 = (rest % tiles_y) * TH;
  const int img = rest / tiles_y;
  const int c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0, RPP)
  const int wq = wave & 3, wh = wave >> 2;        // tile rows 4*wq.., channel half wh (0 unless NW = 8)
  const int sy_base = S2 ? (ty0 > 0 ? 2 * (ty0 - 1) : 0) : (ty0 - 1 > 0 ? ty0 - 1 : 0);
  const char* Xb = PS ? reinterpret_cast<const char*>(a.x) + ((long long)img * 2 * a.Hs + 2 * sy_base) * (2 * a.Ws) * a.ldx * ES
                      : reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncbr = a.cch / KC;                    // real channel blocks
  const int ncb = S2 ? 4 * ncbr : ncbr;           // (plane, channel block) pairs, plane-major
  const int nsteps = ncbr * a.ntaps;
  auto plane_of = [&](int vcb) { return S2 ? (int)(vcb >= ncbr) + (int)(vcb >= 2 * ncbr) + (int)(vcb >= 3 * ncbr) : 0; };
  auto ntaps_of = [&](int vcb) { return S2 ? (0x4221 >> (4 * plane_of(vcb))) & 15 : a.ntaps; };

  unsigned woff[NWL];                             // DMA piece i of this wave: rows wave*32 + 8i .. +7, lane = (row, physical chunk)
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    int row = wave * 32 + i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    row = perm64(row);                              // LDS row holds output channel c0 + perm64(row)
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;
    woff[i] = (unsigned)((long long)row * a.ldw * ES) + logical * 16;
  }
  auto tap_code = [&](int vcb, int tap) {
    const int gt = S2 ? ((0x5310 >> (4 * plane_of(vcb))) & 15) + tap : tap;
    return gt < 8 ? (unsigned)((a.tap_lo >> (8 * gt)) & 0xffull) : (a.tap_hi & 0xffu);
  };
  u32x4_t rp[NPL];
  auto load_patch = [&](int vcb) {
    const int plane = plane_of(vcb), cb = vcb - plane * ncbr;
    const int ppy = plane >> 1, ppx = plane & 1;
    const int psq = PS ? (cb * KC) / a.cps_src_chunks : 0;                 // channel quarter of this block
    const int psy = psq >> 1, psx = psq & 1;
    const long long cboff = PS ? (long long)(cb * KC - psq * a.cps_src_chunks) * EPC * ES : (long long)cb * KC * EPC * ES;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + cboff), 0, (int)DG_OOB_OFF, 0x00020000);
    int r0v = r0;
    asm volatile("" : "+v"(r0v));
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0v + RPP * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      const unsigned off = !ok ? DG_OOB_OFF
                           : PS ? (unsigned)(((2 * (sy - sy_base) + psy) * (2 * a.Ws) + 2 * sx + psx) * a.ldx * ES) + cc * 16
                                : (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  char* const st_base = s_patch + r0 * PITCH + cc * 16;
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + RPP * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(st_base + i * RPP * PITCH) = __builtin_bit_cast(uint4, rp[i]);
    }
  };
  typedef int i32x4h_t __attribute__((ext_vector_type(4)));
  i32x4h_t w_rs;
  int w_dst0 = 0;
  auto dma_setup = [&](int vcb, int tap, int slot) {
    const unsigned code = tap_code(vcb, tap);
    const long long wo = (long long)(code >> 4) * a.Cred + (vcb - plane_of(vcb) * ncbr) * KC * EPC;
    const unsigned long long wbase = (unsigned long long)(Wb + wo * ES);
    w_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wbase);
    w_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32) & 0xffff);
    w_rs[2] = (int)DG_OOB_OFF;
    w_rs[3] = 0x00020000;
    w_dst0 = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_w + slot * (BC * WROW) + (wave * 32) * WROW)));
  };
  auto dma_piece = [&](int i) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(w_dst0 + i * 8 * WROW), "v"(woff[i]), "s"(w_rs) : "memory");
  };
  auto dma_w = [&](int cb, int tap, int slot) {
    dma_setup(cb, tap, slot);
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma_piece(i);
  };
  auto barrier_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };
  auto barrier_keep_patch = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPL) : "memory"); __syncthreads(); };

  f32x4_t acc[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const char* fa_k[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fa_k[kk] = s_w + (wh * 128 + l15) * WROW + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) * 16);
  const char* const fb_lane = s_patch + l15 * PITCH + g * 16;
  auto read_frags = [&](uint4 (&fa)[8], uint4 (&fb)[4], int pa, const char* pb, int kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = *reinterpret_cast<const uint4*>(fa_k[kk] + pa + j * 16 * WROW);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + kk * 64);
  };
  auto mma_rows = [&](const uint4 (&fa)[8], const uint4 (&fb)[4], int j0) {     // two weight fragments x four pixel rows
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
  };
  auto adv = [&](int& c_, int& t_) { if (++t_ == ntaps_of(c_)) { t_ = 0; ++c_; } };
  auto patch_ptr = [&](int vcb_, int tap_) -> const char* {
    const unsigned code = tap_code(vcb_, tap_);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
    return fb_lane + ((wq * 4 + 1 + dy) * PW + 1 + dx) * PITCH;
  };

  int cb = 0, tap = 0, cbw = 0, tapw = 0;
  adv(cbw, tapw); adv(cbw, tapw);      // -> W[2]
  auto next_of = [&](int& ntap, int& ncbn, int ntaps_cb) { ntap = tap + 1; ncbn = cb; if (ntap == ntaps_cb) { ntap = 0; ncbn = cb + 1; } };
  if (!compute) {
    // ---------------- loader waves: every vector-memory instruction of the tile loop
    load_patch(0);
    dma_w(0, 0, 0);
    if (nsteps > 1) { int c1 = 0, t1 = 0; adv(c1, t1); dma_w(c1, t1, 1); }
    store_patch();
    barrier_all();
    for (int s = 0; s < nsteps; ++s) {
      const bool more = s + 1 < nsteps;
      int ntap, ncbn;
      const int ntaps_cb = ntaps_of(cb);
      next_of(ntap, ncbn, ntaps_cb);
      const bool swap = ntap == 0 && more;
      const bool patch_now = cb + 1 < ncb && (ntaps_cb == 1 || tap == 0);
      const bool fetch = s + 2 < nsteps;
      if (patch_now) load_patch(cb + 1);
      if (patch_now && !swap) barrier_keep_patch(); else barrier_all();
      if (swap) { store_patch(); barrier_all(); }
      if (fetch) { dma_setup(cbw, tapw, s & 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_piece(q);
      }
      adv(cbw, tapw);
      tap = ntap; cb = ncbn;
    }
    return;
  }
  // ---------------- compute waves: fragment reads, MFMAs, epilogue; no vector-memory instruction until the epilogue
  __syncthreads();
  // TWO fragment sets: k-block 1 of a step is read while k-block 0 feeds the MFMAs, k-block 0 of the next step (behind the
  // barrier that publishes its weight slot) while k-block 1 does -- one wave per SIMD has nobody else to cover its LDS latency
  uint4 fa0[8], fb0[4], fa1[8], fb1[4];
  const char* pb = patch_ptr(0, 0);
  int pa = 0;
  read_frags(fa0, fb0, pa, pb, 0);
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    int ntap, ncbn;
    const int ntaps_cb = ntaps_of(cb);
    next_of(ntap, ncbn, ntaps_cb);
    const bool swap = ntap == 0 && more;
    read_frags(fa1, fb1, pa, pb, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) mma_rows(fa0, fb0, 2 * q);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    pa = ((s + 1) & 1) * (BC * WROW);
    pb = patch_ptr(ncbn < ncb ? ncbn : 0, ntap);
    if (swap) __syncthreads();
    if (more) read_frags(fa0, fb0, pa, pb, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) mma_rows(fa1, fb1, 2 * q);
    __builtin_amdgcn_sched_barrier(0);
    tap = ntap; cb = ncbn;
  }
  halo_epilogue<T, 2>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g);
}

static int gg_launch_halo_lc(GGArgs& a, int N, hipStream_t st) {
  constexpr int LDS_BYTES = 324 * 144 + 2 * 128 * 128;
  DG_SET_MAX_LDS_ONCE((&gg_halo_lc_kernel), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = (unsigned)((a.Nout + 127) / 128);
  a.nwg = a.nct * (unsigned)(tiles_x * tiles_y * N);
  g_last_kinds |= 8;
  hipLaunchKernelGGL(gg_halo_lc_kernel, dim3(a.nwg), dim3(512), LDS_BYTES, st, a, tiles_x, tiles_y);
  return dg_check_launch();
}
#endif

// ---------------------------------------------------------------------------------------------
// MXFP8 version of the four-wave halo kernel (BASELINE configs[4]: the critic's wide layers, critic.py:25-88): operands are
// OCP E4M3 bytes with one E8M0 scale per block of 32 consecutive channels (csrc/quant.hip, which also records how the
// instruction maps operand bytes and scale lanes to K -- measured with tools/fp8_probe2.hip), the MFMA is
// v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulate, 2x the bf16 rate).  A K-step is still 8 chunks = 128 bytes per
// patch / weight row, now 128 channels, and the LDS images, swizzles, DMA pieces and patch pipeline are those of
// gg_halo4w_kernel: a lane's 32-byte operand is chunk g of the row's first and of its second 64-byte half -- exactly the two
// 16-byte fragments the bf16 kernel feeds to two MFMAs -- so one scaled MFMA replaces two bf16 ones at the same LDS bytes
// and the same matrix-pipe cycles for twice the channels.  Beside the operands, each K-step needs 4 scale bytes per patch
// pixel (fetched with the patch into s_ps) and per weight row (a fifth LDS-DMA piece of waves 0/1 into s_ws); lane group g
// reads byte g.  Output: bf16 through the common epilogue (activations, masks, bit masks as in the bf16 kernel).
struct F8Args { const unsigned char* xs; const unsigned char* ws; int ldxs; };   // ldxs: scale bytes per source pixel (Cred/32 unless the source is a slab slice); weights: 9*Cred/32 per row
typedef int i32x8_t __attribute__((ext_vector_type(8)));

template <bool S2, int NW = 4>       // NW = 8: 256-channel tile, two channel halves on one patch (see gg_halo4w_kernel)
__global__ __launch_bounds__(64 * NW, 2) void gg_halo4w_f8_kernel(const GGArgs a, const F8Args f, int tiles_x, int tiles_y) {
  constexpr int EPC = 16, ES = 1;
  constexpr int NT = 64 * NW, RPP = NT / 8;
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int BC = 32 * NW, KC = 8;                                    // 8 chunks per row: 128 fp8 channels
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B patch rows
  constexpr int WROW = KC * 16;                                          // 128 B weight rows, chunk c of row r at c ^ ((r >> 1) & 7)
  constexpr int NPL = (PROWS * KC + NT - 1) / NT;                        // 11 (6) patch chunks per thread
  constexpr int NPS = (PROWS + NT - 1) / NT;                             // 2 (1) patch scale words per thread
  constexpr int NWL = BC * KC / NT;                                      // 4 weight pieces per wave and step
  constexpr int NSW = BC / 64;                                           // waves that carry distinct scale pieces
  extern __shared__ __attribute__((aligned(16))) char dsm4f8[];
  char* const s_patch = dsm4f8;                                // [PROWS][PITCH]
  char* const s_w = dsm4f8 + PROWS * PITCH;                    // [2][BC][WROW]
  char* const s_ws = s_w + 2 * BC * WROW;                      // [2][BC] u32: 4 scale bytes of the row's 128-channel K-step
  char* const s_ps = s_ws + 2 * BC * 4;                        // [PROWS] u32: same for the patch pixels

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct;
  unsigned rest = tile / a.nct;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  const int ty0 = (rest % tiles_y) * TH;
  const int img = rest / tiles_y;
  const int c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0, RPP)
  const int wq = wave & 3, wh = wave >> 2;        // tile rows 4*wq.., channel half wh (0 unless NW = 8)
  const int sy_base = S2 ? (ty0 > 0 ? 2 * (ty0 - 1) : 0) : (ty0 - 1 > 0 ? ty0 - 1 : 0);
  const char* Xb = reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const int ldxs = f.ldxs, ldws = 9 * (a.Cred >> 5);
  const char* XSb = reinterpret_cast<const char*>(f.xs) + ((long long)img * a.Hs + sy_base) * a.Ws * ldxs;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;
  const char* WSb = reinterpret_cast<const char*>(f.ws) + (long long)c0 * ldws;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncbr = a.cch / KC;                    // real 128-channel blocks
  const int ncb = S2 ? 4 * ncbr : ncbr;           // (plane, channel block) pairs, plane-major
  const int nsteps = ncbr * a.ntaps;
  auto plane_of = [&](int vcb) { return S2 ? (int)(vcb >= ncbr) + (int)(vcb >= 2 * ncbr) + (int)(vcb >= 3 * ncbr) : 0; };
  auto ntaps_of = [&](int vcb) { return S2 ? (0x4221 >> (4 * plane_of(vcb))) & 15 : a.ntaps; };

  unsigned woff[NWL];                             // DMA piece i of this wave: rows wave*32 + 8i .. +7, lane = (row, physical chunk)
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    int row = wave * 32 + i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    row = perm64(row);                              // LDS row holds output channel c0 + perm64(row)
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;
    woff[i] = (unsigned)((long long)row * a.ldw * ES) + logical * 16;
  }
  unsigned wsoff;                                 // scale piece (waves 0 and 1): LDS row wave*64 + lane
  {
    int row = perm64((wave & (NSW - 1)) * 64 + lane);
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;
    wsoff = (unsigned)(row * ldws);
  }
  auto tap_code = [&](int vcb, int tap) {
    const int gt = S2 ? ((0x5310 >> (4 * plane_of(vcb))) & 15) + tap : tap;
    return gt < 8 ? (unsigned)((a.tap_lo >> (8 * gt)) & 0xffull) : (a.tap_hi & 0xffu);
  };
  u32x4_t rp[NPL];
  unsigned rps[NPS];
  auto load_patch = [&](int vcb) {
    const int plane = plane_of(vcb), cb = vcb - plane * ncbr;
    const int ppy = plane >> 1, ppx = plane & 1;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + (long long)cb * KC * EPC * ES), 0, (int)DG_OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc((void*)(XSb + cb * 4), 0, (int)DG_OOB_OFF, 0x00020000);
    int r0v = r0;
    asm volatile("" : "+v"(r0v));
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0v + RPP * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      const unsigned off = !ok ? DG_OOB_OFF : (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NPS; ++i) {               // scale words of patch rows tid (and tid + 256)
      const int pr = tid + NT * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      rps[i] = __builtin_amdgcn_raw_buffer_load_b32(rxs, ok ? (unsigned)(((sy - sy_base) * a.Ws + sx) * ldxs) : DG_OOB_OFF, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  char* const st_base = s_patch + r0 * PITCH + cc * 16;
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + RPP * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(st_base + i * RPP * PITCH) = __builtin_bit_cast(uint4, rp[i]);
    }
#pragma unroll
    for (int i = 0; i < NPS; ++i)
      if (tid + NT * i < PROWS) *reinterpret_cast<unsigned*>(s_ps + (tid + NT * i) * 4) = rps[i];
  };
  typedef int i32x4h_t __attribute__((ext_vector_type(4)));
  i32x4h_t w_rs, ws_rs;
  int w_dst0 = 0, ws_dst = 0;
  auto dma_setup = [&](int vcb, int tap, int slot) {
    const unsigned code = tap_code(vcb, tap);
    const int cbr = vcb - plane_of(vcb) * ncbr;
    const long long wo = (long long)(code >> 4) * a.Cred + cbr * KC * EPC;
    const unsigned long long wbase = (unsigned long long)(Wb + wo * ES);
    w_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wbase);
    w_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32) & 0xffff);
    w_rs[2] = (int)DG_OOB_OFF;
    w_rs[3] = 0x00020000;
    w_dst0 = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_w + slot * (BC * WROW) + (wave * 32) * WROW)));
    const unsigned long long sbase = (unsigned long long)(WSb + (long long)(code >> 4) * (a.Cred >> 5) + cbr * 4);
    ws_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)sbase);
    ws_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(sbase >> 32) & 0xffff);
    ws_rs[2] = (int)DG_OOB_OFF;
    ws_rs[3] = 0x00020000;
    ws_dst = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_ws + slot * (BC * 4) + (wave & (NSW - 1)) * 256)));
  };
  auto dma_piece = [&](int i) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(w_dst0 + i * 8 * WROW), "v"(woff[i]), "s"(w_rs) : "memory");
  };
  // every wave issues the scale piece (waves 2/3 re-write the rows of waves 0/1 with the same bytes), so the per-wave count
  // of outstanding vector-memory operations -- what the counted s_waitcnt below relies on -- is the same in all four waves
  auto dma_scales = [&]() {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(ws_dst), "v"(wsoff), "s"(ws_rs) : "memory");
  };
  auto dma_w = [&](int cb, int tap, int slot) {
    dma_setup(cb, tap, slot);
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma_piece(i);
    dma_scales();
  };
  auto barrier_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };
  auto barrier_keep_patch = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPL + NPS) : "memory"); __syncthreads(); };   // patch chunks + scale words

  f32x4_t acc[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const char* fa_k[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fa_k[kk] = s_w + (wh * 128 + l15) * WROW + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) * 16);
  const char* const fb_lane = s_patch + l15 * PITCH + g * 16;
  const char* const sa_lane = s_ws + (wh * 128 + l15) * 4 + g;
  const char* const sb_lane = s_ps + l15 * 4 + g;
  // A lane's 32-byte operand = chunk g of the row's first and second 64-byte half.  All fragments of a step are read and
  // consumed INSIDE the step (nothing is carried around the loop: a loop-carried 8-register value gets split into two
  // 4-register halves by SROA and copied back together, 32 v_mov per step), and no read result is ever passed through an
  // asm statement (that forces a full s_waitcnt lgkmcnt(0) right behind the read -- the first version of this kernel
  // paid two serialised LDS latencies per weight fragment that way).
  auto ld32 = [](const char* p0, const char* p1) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p0), hi = *reinterpret_cast<const uint4*>(p1);
    i32x8_t o;
    o[0] = (int)lo.x; o[1] = (int)lo.y; o[2] = (int)lo.z; o[3] = (int)lo.w;
    o[4] = (int)hi.x; o[5] = (int)hi.y; o[6] = (int)hi.z; o[7] = (int)hi.w;
    return o;
  };
  // one tap-step: B fragments (4 pixel rows) + the first weight fragment are read at the top -- the DMA issue of W[s+1] that
  // follows covers their latency -- then weight fragment j+1 is read while the four MFMAs of fragment j run (two fragment
  // registers sets of 8, ping-pong).  The asm fences keep every read in its slot and every MFMA row between its fences
  // (an MFMA is not a memory operation: without the accumulator operands the compiler sinks MFMAs below later reads).
#define F8_FENCE(j) asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]), "+v"(acc[j][2]), "+v"(acc[j][3]) :: "memory")
  auto adv = [&](int& c_, int& t_) { if (++t_ == ntaps_of(c_)) { t_ = 0; ++c_; } };
  auto patch_row = [&](int vcb_, int tap_) {
    const unsigned code = tap_code(vcb_, tap_);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
    return (wq * 4 + 1 + dy) * PW + 1 + dx;
  };

  load_patch(0);
  int cb = 0, tap = 0, cbw = 0, tapw = 0;
  dma_w(0, 0, 0);
  adv(cbw, tapw);                    // -> W[1], issued at the top of step 0
  store_patch();
  barrier_all();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    int ntap = tap + 1, ncbn = cb;
    const int ntaps_cb = ntaps_of(cb);
    if (ntap == ntaps_cb) { ntap = 0; ncbn = cb + 1; }
    const bool swap = ntap == 0 && more;
    const bool patch_now = cb + 1 < ncb && (ntaps_cb == 1 || tap == 0);     // fetch the next block's patch during its predecessor's first step
    const int slot = s & 1, pa = slot * (BC * WROW);
    const int prow = patch_row(cb, tap);
    const char* const pb = fb_lane + prow * PITCH;
    const char* const sap = sa_lane + slot * (BC * 4);
    i32x8_t fb[4], fa[2];
    int sb[4], sa[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fb[i] = ld32(pb + i * PW * PITCH, pb + i * PW * PITCH + 64);
      sb[i] = *reinterpret_cast<const unsigned char*>(sb_lane + (prow + i * PW) * 4);
    }
    fa[0] = ld32(fa_k[0] + pa, fa_k[1] + pa);
    sa[0] = *reinterpret_cast<const unsigned char*>(sap);
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    if (more) {                      // W[s+1] -> the other slot (free since the barrier that ended step s-1)
      dma_setup(cbw, tapw, slot ^ 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_piece(q);
      dma_scales();
    }
    adv(cbw, tapw);
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    // the next block's patch goes out behind the DMA pieces (the barrier's counted wait then leaves it in flight) and before
    // the MFMA rows: here only the B fragments and one weight fragment are live beside the prefetch registers
    if (patch_now) {
      load_patch(cb + 1);
      asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < 7) {
        fa[(j + 1) & 1] = ld32(fa_k[0] + pa + (j + 1) * 16 * WROW, fa_k[1] + pa + (j + 1) * 16 * WROW);
        sa[(j + 1) & 1] = *reinterpret_cast<const unsigned char*>(sap + (j + 1) * 64);
      }
      F8_FENCE(j); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[j & 1], fb[i], acc[j][i], 0, 0, 0, sa[j & 1], 0, sb[i]);
      F8_FENCE(j); __builtin_amdgcn_sched_barrier(0);
    }
    // every wave has read all it needs of this step: the barrier frees slot s&1 (and, at a block end, the patch); W[s+1] has landed
    if (patch_now && !swap) barrier_keep_patch(); else barrier_all();
    if (swap) {                      // channel-block boundary: the single-buffered patch is rewritten, then published
      store_patch();
      barrier_all();
    }
    tap = ntap; cb = ncbn;
  }
#undef F8_FENCE
  halo_epilogue<bf16_t, 2>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g);
}

template <bool S2, int NW = 4>
static int gg_launch_halo4w_f8(GGArgs& a, const F8Args& f, int N, hipStream_t st) {
  constexpr int BC = 32 * NW;
  constexpr int LDS_BYTES = 324 * 144 + 2 * BC * 128 + 2 * BC * 4 + 324 * 4;
  DG_SET_MAX_LDS_ONCE((&gg_halo4w_f8_kernel<S2, NW>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = (unsigned)((a.Nout + BC - 1) / BC);
  a.nwg = a.nct * (unsigned)(tiles_x * tiles_y * N);
  g_last_kinds |= 32;
  hipLaunchKernelGGL((gg_halo4w_f8_kernel<S2, NW>), dim3(a.nwg), dim3(64 * NW), LDS_BYTES, st, a, f, tiles_x, tiles_y);
  return dg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Halo kernel for layers with <= 16 output channels (generator conv3.2: 128 -> 2 at 1024^2; the critic's first-layer data
// gradient in the penalty): HBM-bound, 1/8 of the MFMA work of a 128-wide tile.  The per-tap kernel re-reads every input
// pixel 9x through L2 (4.9 ms per pass against a 1.7 ms HBM floor); here a 16x16-pixel tile keeps the (16+2)^2 patch of one
// 64-channel block in LDS for all taps, like the wide halo kernel, and ALL nine 16x64 weight tiles of the block beside it
// (67 KB in total, so two workgroups share a CU and one's loads overlap the other's MFMAs: no software pipeline needed).
// 4 waves, wave = 4 tile rows x 16 channels; per block and tap two k-chunks of (1 weight + 4 patch fragment reads, 4 MFMAs).
template <typename T>
__global__ __launch_bounds__(256, 2) void gg_halo16_kernel(const GGArgs a, int tiles_x, int tiles_y) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int KC = 8;                                                  // 16-byte chunks per block row (64 bf16 / 32 fp32 channels)
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B: conflict-free fragment reads at any row offset
  constexpr int NPL = (PROWS * KC + 255) / 256;                          // 11 patch chunks per thread
  constexpr int WROWS = 9 * 16;
  constexpr int NWL = (WROWS * KC + 255) / 256;                          // 5 weight chunks per thread
  extern __shared__ __attribute__((aligned(16))) char dsm16[];
  char* const s_patch = dsm16;                    // [PROWS][PITCH]
  char* const s_w = dsm16 + PROWS * PITCH;        // [9 taps][16 channels][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  unsigned rest = tile;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  const int ty0 = (rest % tiles_y) * TH;
  const int img = rest / tiles_y;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0,32)
  const int sy_base = ty0 - 1 > 0 ? ty0 - 1 : 0;
  const char* Xb = reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w);
  const int l15 = lane & 15, g = lane >> 4;
  const int ncb = a.cch / KC, ntaps = a.ntaps;

  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // the next block's patch and weights are fetched into registers while the current block computes (the kernel is
  // HBM-latency-bound: with the loads issued at the top of their own block every block waited a full memory round trip)
  u32x4_t rp[NPL], rw[NWL];
  auto load_block = [&](int cb) {
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + (long long)cb * KC * EPC * ES), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + 32 * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = ty0 - 1 + py, sx = tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      const unsigned off = ok ? (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16 : DG_OOB_OFF;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
    __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc((void*)(Wb + (long long)cb * KC * EPC * ES), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int wr = r0 + 32 * i;                 // row = tap * 16 + channel
      const int t = wr >> 4, n = wr & 15;
      unsigned off = DG_OOB_OFF;
      if (wr < WROWS && t < ntaps && n < a.Nout) {
        const unsigned code = t < 8 ? (unsigned)((a.tap_lo >> (8 * t)) & 0xffull) : (a.tap_hi & 0xffu);
        off = (unsigned)(((long long)n * a.ldw + (long long)(code >> 4) * a.Cred) * ES) + cc * 16;
      }
      rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rwd, off, 0, 0);
    }
  };
  load_block(0);
  for (int cb = 0; cb < ncb; ++cb) {
    __syncthreads();                                // everybody is done with the previous block's tiles
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + 32 * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(s_patch + pr * PITCH + cc * 16) = __builtin_bit_cast(uint4, rp[i]);
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int wr = r0 + 32 * i;
      if (wr < WROWS) *reinterpret_cast<uint4*>(s_w + wr * PITCH + cc * 16) = __builtin_bit_cast(uint4, rw[i]);
    }
    __syncthreads();
    if (cb + 1 < ncb) load_block(cb + 1);
    for (int t = 0; t < ntaps; ++t) {
      const unsigned code = t < 8 ? (unsigned)((a.tap_lo >> (8 * t)) & 0xffull) : (a.tap_hi & 0xffu);
      const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
      const char* pb = s_patch + ((wave * 4 + 1 + dy) * PW + 1 + dx + l15) * PITCH + g * 16;
      const char* pa = s_w + (t * 16 + l15) * PITCH + g * 16;
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        const uint4 fa = *reinterpret_cast<const uint4*>(pa + kk * 64);
        uint4 fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + kk * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(fa, fb[i], acc[i]);
      }
    }
  }

  // epilogue: lane = pixel (tile row wave*4 + i, column l15), channels 4g .. 4g+3
  typedef EpiIO<T> IO;
  const int psm = a.dy_mul, psx = a.dx_mul;
  const long long pbase = ((long long)img * a.Hd + (long long)ty0 * psm + a.dy_off) * a.Wd + (long long)tx0 * psx + a.dx_off;
  const int cj = 4 * g;
  const bool cok = cj < a.Nout && tx0 + l15 < a.Wg;
  const float4 bias = (a.bias && cj < a.Nout) ? *reinterpret_cast<const float4*>(a.bias + cj) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto rsrc = [&](const void* p, long long ld) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + pbase * ld * ES), 0, (int)DG_OOB_OFF, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rY = rsrc(a.y, a.ldy);
  const __amdgpu_buffer_rsrc_t r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1), r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2),
                               rm = rsrc(a.mask ? a.mask : a.y, a.ldmask);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool ok = cok && ty0 + wave * 4 + i < a.Hg;
    const int pix = (wave * 4 + i) * psm * a.Wd + l15 * psx;
    const unsigned oy = ok ? (unsigned)((pix * (int)a.ldy + cj) * ES) : DG_OOB_OFF;
    float v[4] = {acc[i][0] + bias.x, acc[i][1] + bias.y, acc[i][2] + bias.z, acc[i][3] + bias.w};
    float r[4];
    if (a.has_act) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = leaky(v[e], a.act_slope);
    }
    if (a.r1) {
      IO::unpack(IO::load(r1, ok ? (unsigned)((pix * (int)a.ldr1 + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * a.s1 + r[e];
    }
    if (a.r2) {
      IO::unpack(IO::load(r2, ok ? (unsigned)((pix * (int)a.ldr2 + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * a.s2 + r[e];
    }
    const bool mask_on = a.mask && a.mask_c0 == 0;          // (<= 16 output channels: mask_c0 is 0 or excludes them all)
    if (mask_on && !a.mask_last) {
      IO::unpack(IO::load(rm, ok ? (unsigned)((pix * (int)a.ldmask + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(r[e], a.mask_slope);
    }
    if (a.accumulate) {
      IO::unpack(IO::load(rY, oy), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    if (mask_on && a.mask_last) {
      IO::unpack(IO::load(rm, ok ? (unsigned)((pix * (int)a.ldmask + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(r[e], a.mask_slope);
    }
    IO::store(v, rY, oy);
  }
}

template <typename T>
static int gg_launch_halo16(GGArgs& a, int N, hipStream_t st) {
  constexpr int LDS_BYTES = (324 + 9 * 16) * 144;
  DG_SET_MAX_LDS_ONCE((&gg_halo16_kernel<T>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = 1;
  a.nwg = (unsigned)(tiles_x * tiles_y * N);
  g_last_kinds |= 8;
  hipLaunchKernelGGL((gg_halo16_kernel<T>), dim3(a.nwg), dim3(256), LDS_BYTES, st, a, tiles_x, tiles_y);
  return dg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// im2col path for stride-1 FORWARD layers with <= 2 real input channels (critic features.0 on the
// 1024^2 tiles, generator conv1 with 2 covariates).  K = 9 taps x 2 channels = 18, so
// the layer is bound by writing its output to HBM, not by MFMA: the 18 (padded to 32) K values of every
// pixel are gathered straight from the 2 real channels (4/8-byte loads) instead of walking 9 taps x 16
// padded channels, the weight tile is built once per workgroup, and each workgroup streams several
// 128-pixel tiles.
template <typename T, bool LEAN>
__global__ __launch_bounds__(256, LEAN ? 3 : 2) void gg_im2col_kernel(const GGArgs a, int tiles_per_block) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int KCH = 32 / EPC;         // 16-B chunks per 32-element K row
  constexpr int TPC = EPC / 2;          // taps per chunk (2 channels per tap)
  constexpr int NCH = 128 * KCH / 256;  // chunks per thread per tile
  __shared__ uint4 sW[128 * KCH], sX[128 * KCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.y * 128;
  const T* X = reinterpret_cast<const T*>(a.x);
  const T* Wt = reinterpret_cast<const T*>(a.w);
  auto swz = [](int row) { return KCH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  auto load_pair = [](const T* p, unsigned* w2) {
    if constexpr (sizeof(T) == 2) { w2[0] = *reinterpret_cast<const unsigned*>(p); }
    else { const uint2 v = *reinterpret_cast<const uint2*>(p); w2[0] = v.x; w2[1] = v.y; }
  };
  // weight tile: row = output channel, K element k = tap*2 + ci
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int e = tid + 256 * i, row = e / KCH, col = e % KCH;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    const int n = c0 + perm64(row);                 // 64-channel wave tiles: permuted channel order
    if (n < a.Nout) {
#pragma unroll
      for (int tt = 0; tt < TPC; ++tt) {
        const int tp = col * TPC + tt;
        if (tp < 9) load_pair(Wt + ((long long)n * 9 + tp) * a.Cred, w + tt * (4 / TPC));
      }
    }
    sW[row * KCH + (col ^ swz(row))] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  const int wp = wave & 1, wc = wave >> 1;
  const int l15 = lane & 15, g = lane >> 4;
  constexpr int ES = (int)sizeof(T);
  // Everything below is branch-free per tile (buffer loads / stores with out-of-range offsets for padding, tile ends and
  // channel tails): the compiler can then count the outstanding operations and the LDS write of the next tile's gathered
  // rows waits for the gather loads only (s_waitcnt vmcnt(#stores issued after them)), not for this tile's stores to
  // reach memory -- with per-load branches it fell back to vmcnt(0) at the loop head, i.e. one store round trip per tile.
  // forward, stride 1, plain destination: GEMM row m is both the source and the destination pixel index
  const int col = tid % KCH, row0 = tid / KCH;                   // this thread's K chunk; rows row0 + (256 / KCH) * i
  int tap_rel[TPC];                                              // source pixel shift of the chunk's taps
  int tap_dy[TPC], tap_dx[TPC];
#pragma unroll
  for (int tt = 0; tt < TPC; ++tt) {
    const int tp = col * TPC + tt;
    tap_dy[tt] = tp < 9 ? tp / 3 - 1 : 4;                        // 4: never inside the image
    tap_dx[tt] = tp < 9 ? tp % 3 - 1 : 0;
    tap_rel[tt] = tap_dy[tt] * a.Ws + tap_dx[tt];
  }
  const int cb16 = c0 + wc * 64 + 16 * g;
  const bool cok = cb16 < a.Nout;
  float bias[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b4 = (a.bias && cok) ? *reinterpret_cast<const float4*>(a.bias + cb16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
  }
  const int ldb = (a.Nout >> 6) * 4, bidx = ((c0 + wc * 64) >> 6) * 4 + g;
  // the im2col rows of the NEXT tile are gathered into registers while the current tile's MFMAs and (long) store epilogue
  // run: the 4/8-byte gathers are latency-bound and nothing else would hide them (one barrier pair per tile)
  unsigned gw[NCH][4];
  auto gather = [&](int p0, bool live) {
    long long pbase = (long long)p0 - a.Ws - 1;
    if (pbase < 0) pbase = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.x) + pbase * a.ldx * ES), 0,
                                                                      (int)DG_OOB_OFF, 0x00020000);
    const int mrel = (int)(p0 - pbase);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = row0 + (256 / KCH) * i;
      const unsigned m = (unsigned)(p0 + row);
      const unsigned tq = m / (unsigned)a.Wg;
      const int gx = (int)(m - tq * (unsigned)a.Wg), gy = (int)(tq % (unsigned)a.Hg);
      const bool mok = live & (m < (unsigned)a.M);
#pragma unroll
      for (int tt = 0; tt < TPC; ++tt) {
        const bool ok = (int)mok & (int)((unsigned)(gy + tap_dy[tt]) < (unsigned)a.Hs) & (int)((unsigned)(gx + tap_dx[tt]) < (unsigned)a.Ws);
        const unsigned off = ok ? (unsigned)((mrel + row + tap_rel[tt]) * (int)a.ldx * ES) : DG_OOB_OFF;
        if constexpr (sizeof(T) == 2) gw[i][tt] = __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0);
        else {
          const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
          gw[i][2 * tt] = v[0]; gw[i][2 * tt + 1] = v[1];
        }
      }
    }
  };
  const int tile0 = blockIdx.x * tiles_per_block;
  auto publish = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = row0 + (256 / KCH) * i;
      sX[row * KCH + (col ^ swz(row))] = make_uint4(gw[i][0], gw[i][1], gw[i][2], gw[i][3]);
    }
    __syncthreads();
  };
  gather(tile0 * 128, tile0 * 128 < a.M);
  publish();
  // per tile: mask words | gather of tile t+1 | MFMAs | stores | barrier | gathered rows -> LDS | barrier.  The LDS write
  // sits at the END of the body so that its wait is "all but the 12 stores issued after the gather" on every path
  for (int t = 0; t < tiles_per_block; ++t) {
    const int p0 = (tile0 + t) * 128;
    if (p0 >= a.M) break;
    // operand resources of this tile (destination pixel = GEMM row, everything based at the tile's first pixel) and its
    // mask words, ahead of the next tile's gather: the epilogue then waits for these loads only
    auto rsrc = [&](const void* p, long long ld, int es) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + (long long)p0 * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
    };
    EpiRes R;
    R.rY = rsrc(a.y, a.ldy, ES);
    R.r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1, ES); R.r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2, ES); R.rm = rsrc(a.mask ? a.mask : a.y, a.ldmask, ES);
    R.rbi = rsrc(a.mask_bits ? a.mask_bits : a.y, ldb, 2); R.rbo = rsrc(a.out_bits ? a.out_bits : a.y, ldb, 2);
    R.rq = rsrc(a.out_q ? a.out_q : a.y, a.ldy, 1); R.rqs = rsrc(a.out_qs ? a.out_qs : a.y, a.ldqs, 1);
    R.ldy = (int)a.ldy; R.ld1 = (int)a.ldr1; R.ld2 = (int)a.ldr2; R.ldm = (int)a.ldmask;
    unsigned mbv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rel = wp * 64 + 16 * i + l15;
      mbv[i] = epi64_bits<LEAN>(a, R, (cok && p0 + rel < a.M) ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF);
    }
    gather(p0 + 128, t + 1 < tiles_per_block);
    f32x4_t acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KCH / 4; ++kk) {
      const int ch = kk * 4 + g;
      uint4 fa[4], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int row = wc * 64 + 16 * j + l15; fa[j] = sW[row * KCH + (ch ^ swz(row))]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = wp * 64 + 16 * i + l15; fb[i] = sX[row * KCH + (ch ^ swz(row))]; }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rel = wp * 64 + 16 * i + l15;
      const bool ok = cok && p0 + rel < a.M;
      epi64_pixel<T, LEAN>(a, R, acc[0][i], acc[1][i], acc[2][i], acc[3][i], bias,
                           ok ? (unsigned)((rel * R.ldy + cb16) * ES) : DG_OOB_OFF, ok ? (unsigned)((rel * R.ld1 + cb16) * ES) : DG_OOB_OFF,
                           ok ? (unsigned)((rel * R.ld2 + cb16) * ES) : DG_OOB_OFF, ok ? (unsigned)((rel * R.ldm + cb16) * ES) : DG_OOB_OFF,
                           ok ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF, mbv[i], a.mask && cb16 >= a.mask_c0);
    }
    __syncthreads();
    publish();
  }
}

// ---------------------------------------------------------------------------------------------
// The same layer without LDS and without barriers (bf16, image width a multiple of 16, 128-channel tiles, bias / activation /
// bit-mask / MXFP8-copy epilogues).  gg_im2col_kernel moves every tile through gather -> LDS -> barrier -> MFMA -> stores ->
// barrier with 12 waves per CU; its 2.7-3.4 TB/s of stores is per-tile latency, not bandwidth.  Here a WAVE owns a group of 16
// consecutive pixels of one image row and all 128 channels: the MFMA's B fragment of pixel n is K elements 8g..8g+7 = taps
// 4g..4g+3 x 2 channels, i.e. four dwords of x (lane groups 0-1: taps 0-7, group 2: tap 8, group 3: nothing), the eight A fragments (18 x 128 weights) live in registers for the whole kernel, and the BIAS rides in the two
// spare K slots 18/19 of lane group 2 (weight = bias split into a bf16 high and low part, pixel value = 1.0 twice: exact to 2^-17
// of the bias), so the epilogue is activation + rounding + stores.  Nothing is shared between waves: 16 waves per CU, each with
// the next group's pixels in flight behind the current group's eight 16-byte stores per lane.  Waves sweep the pixel
// groups interleaved (group = iteration * waves + wave), so the chip writes one moving window of the output.
template <int F>
__global__ __launch_bounds__(256, 4) void gg_im2col_direct_kernel(const GGArgs a, int ngroups, bool tiled) {
  typedef bf16_t T;
  constexpr int ES = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.y * 128;
  const T* Wt = reinterpret_cast<const T*>(a.w);
  uint4 fa[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = c0 + (j >> 2) * 64 + perm64((j & 3) * 16 + l15);      // channel of A row l15 of fragment j (Nout % 128 == 0)
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (g < 2) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) w[tt] = *reinterpret_cast<const unsigned*>(Wt + ((long long)n * 9 + 4 * g + tt) * a.Cred);
    } else if (g == 2) {
      w[0] = *reinterpret_cast<const unsigned*>(Wt + ((long long)n * 9 + 8) * a.Cred);
      const float b = a.bias ? a.bias[n] : 0.f;
      const bf16_t hi = f32_to_bf16(b), lo = f32_to_bf16(b - bf16_to_f32(hi));
      w[1] = (unsigned)hi | ((unsigned)lo << 16);
    }
    fa[j] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // The 16 pixels of a group read 3 rows x 18 columns of x = 54 dwords (2 channels each): ONE load per lane brings them in
  // (lane i < 54: row i / 18 - 1, column i % 18 - 1 relative to the group's first pixel; out-of-image -> 0) and four
  // ds_bpermute_b32 hand every lane its taps (lane constants; K slots past the 9 taps point at lane 63, which always holds 0).
  // Four gathers per lane straight from x touched ~48 cache lines per group in the 16-channel-padded layout (15 now).
  const int ld_r = lane / 18 - 1, ld_c = lane % 18 - 1;
  int perm_src[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int tp = 4 * g + tt;
    perm_src[tt] = tp < 9 ? ((tp / 3) * 18 + l15 + tp % 3) * 4 : 63 * 4;
  }
  const unsigned ones = g == 2 ? 0x3f803f80u : 0u;                 // K slots 18 / 19 of the pixel operand: bf16 1.0 twice
  const int ldb = (a.Nout >> 6) * 4;
  unsigned offy[2], boff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    offy[h] = (unsigned)((l15 * (int)a.ldy + c0 + 64 * h + 16 * g) * ES);
    boff[h] = (unsigned)((l15 * ldb + ((c0 + 64 * h) >> 6) * 4 + g) * 2);
  }
  // out_bits: a pixel's record is 8 words (128 channels); lane group g gets words g (first half) and 4 + g (second half) out of
  // the two epilogue calls.  Stored as ONE dword per lane -- group g writes words 2g, 2g+1, fetched from groups 2(g&1), 2(g&1)+1
  // by two ds_bpermute -- so a group's 256 bytes of mask words leave as one contiguous wave store instead of two scattered 2-byte
  // ones (the two 2-byte stores cost 8-14 % of the launch).
  const int ob_src = (l15 + 32 * (g & 1)) * 4, ob_sh = 16 * (g >> 1);
  const unsigned ob_off = (unsigned)((l15 * ldb + (c0 >> 6) * 4 + 2 * g) * 2);
  const float zero16[16] = {};
  auto gather = [&](int grp) -> unsigned {
    const bool live = grp >= 0;
    const unsigned m0 = live ? (unsigned)grp * 16u : 0u;
    const unsigned tq = m0 / (unsigned)a.Wg;
    const int sx = (int)(m0 - tq * (unsigned)a.Wg) + ld_c, sy = (int)(tq % (unsigned)a.Hg) + ld_r;
    long long pbase = (long long)m0 - a.Ws - 1;
    if (pbase < 0) pbase = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.x) + pbase * a.ldx * ES), 0,
                                                                      (int)DG_OOB_OFF, 0x00020000);
    const int mrel = (int)((long long)m0 - pbase) + ld_r * a.Ws + ld_c;
    const bool ok = (int)live & (int)(lane < 54) & (int)((unsigned)sy < (unsigned)a.Hs) & (int)((unsigned)sx < (unsigned)a.Ws);
    return __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (unsigned)(mrel * (int)a.ldx * ES) : DG_OOB_OFF, 0, 0);
  };
  // group order.  tiled (image height a multiple of 16): a workgroup walks 16x16-pixel tiles, wave w rows 4w..4w+3 of each (the
  // write pattern of the tiled conv kernels: 3.6 against 3.3 TB/s for the linear order, whose 4096 waves write one 16-MB window);
  // otherwise wave k of the grid takes groups k, k + waves, ...
  const int nwaves = (int)gridDim.x * 4;
  const int tiles_x = a.Wg / 16, tiles_y = a.Hg / 16, ntiles = tiled ? tiles_x * tiles_y * (a.M / (a.Hg * a.Wg)) : 0;
  auto grp_of = [&](int it) -> int {                 // < 0: past this wave's last group
    if (!tiled) { const int gq = (int)blockIdx.x * 4 + wave + it * nwaves; return gq < ngroups ? gq : -1; }
    unsigned tile = blockIdx.x + (unsigned)(it >> 2) * gridDim.x;
    if (tile >= (unsigned)ntiles) return -1;
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y, img = tile / tiles_y;
    return ((img * a.Hg + ty * 16 + 4 * wave + (it & 3)) * a.Wg + tx * 16) >> 4;
  };
  // a group's pixels are fetched two groups ahead (memory operations retire in order: a load comes back only after the stores
  // issued before it have been acknowledged)
  unsigned x0 = gather(grp_of(0)), x1 = gather(grp_of(1));
  for (int it = 0;; ++it) {
    const int grp = grp_of(it);
    if (grp < 0) break;
    const uint4 fb = make_uint4((unsigned)__builtin_amdgcn_ds_bpermute(perm_src[0], (int)x0),
                                (unsigned)__builtin_amdgcn_ds_bpermute(perm_src[1], (int)x0) | ones,
                                (unsigned)__builtin_amdgcn_ds_bpermute(perm_src[2], (int)x0),
                                (unsigned)__builtin_amdgcn_ds_bpermute(perm_src[3], (int)x0));
    const long long m0 = (long long)grp * 16;
    auto rsrc = [&](const void* p, long long ld, int es) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + m0 * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
    };
    EpiRes R;
    R.rY = rsrc(a.y, a.ldy, ES);
    R.r1 = R.r2 = R.rm = R.rY;
    R.rbi = (F & 2) ? rsrc(a.mask_bits, ldb, 2) : R.rY; R.rbo = (F & 4) ? rsrc(a.out_bits, ldb, 2) : R.rY;
    R.rq = (F & 256) ? rsrc(a.out_q, a.ldy, 1) : R.rY; R.rqs = (F & 256) ? rsrc(a.out_qs, a.ldqs, 1) : R.rY;
    R.ldy = (int)a.ldy; R.ld1 = R.ld2 = R.ldm = 0;
    unsigned mb[2] = {0u, 0u};
    if (F & 2) {                                                   // both mask words before the first store
      mb[0] = __builtin_amdgcn_raw_buffer_load_b16(R.rbi, boff[0], 0, 0);
      mb[1] = __builtin_amdgcn_raw_buffer_load_b16(R.rbi, boff[1], 0, 0);
    }
    const unsigned x2 = gather(grp_of(it + 2));
    unsigned ob[2] = {0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4_t acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        Mma<T>::run(fa[4 * h + j], fb, acc[j]);
      }
      epi64_pixel<T, true, F | 512 | ((F & 4) ? 1024 : 0)>(a, R, acc[0], acc[1], acc[2], acc[3], zero16, offy[h], 0u, 0u, 0u, boff[h], mb[h],
                                                            false, &ob[h]);
    }
    if (F & 4) {
      const int both = (int)(ob[0] | (ob[1] << 16));
      const unsigned lo = ((unsigned)__builtin_amdgcn_ds_bpermute(ob_src, both) >> ob_sh) & 0xffffu;
      const unsigned hi = ((unsigned)__builtin_amdgcn_ds_bpermute(ob_src + 64, both) >> ob_sh) & 0xffffu;
      __builtin_amdgcn_raw_buffer_store_b32(lo | (hi << 16), R.rbo, ob_off, 0, 0);
    }
    x0 = x1; x1 = x2;
  }
}

template <typename T>
static int gg_launch_im2col(GGArgs& a, hipStream_t st) {
  const int tiles = (a.M + 127) / 128;
  int tpb = tiles / 2048;            // a few tiles per workgroup so the weight tile is built rarely
  if (tpb < 1) tpb = 1;
  if (tpb > 16) tpb = 16;
  // big launches: ONE round of the resident workgroups, every workgroup the same number of tiles (16 tiles per workgroup
  // left 16384 workgroups on 768 slots: 21.3 rounds, the last a third full; +2 % at 1024^2)
  const bool lean = !a.r1 && !a.r2 && !a.mask && !a.accumulate;
  if constexpr (sizeof(T) == 2) {
    static const bool no_direct = getenv("DG_GG_NOIM2COLDIRECT") != nullptr;
    const int F = (a.has_act ? 1 : 0) | (a.mask_bits ? 2 : 0) | (a.out_bits ? 4 : 0) | (a.out_q ? 256 : 0);
    if (!no_direct && lean && a.Wg % 16 == 0 && a.Nout % 128 == 0 && a.Hs == a.Hg && a.Ws == a.Wg &&
        (long long)a.M * a.ldy * 2 < (1ll << 46) && (F == 0 || F == 1 || F == 2 || F == 5 || F == 258 || F == 261)) {
      const int ngroups = a.M / 16;
      int nb = a.Hg % 16 == 0 ? ngroups / 16 : (ngroups + 3) / 4;   // tiles of 16 groups / workgroups of 4 groups
      if (nb > 1024) nb = 1024;                                    // 4 workgroups per CU resident, one round
      dim3 grid(nb, a.Nout / 128);
      g_last_kinds |= 16;
      const bool tiled = a.Hg % 16 == 0;
      switch (F) {
        case 0: hipLaunchKernelGGL((gg_im2col_direct_kernel<0>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 1: hipLaunchKernelGGL((gg_im2col_direct_kernel<1>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 2: hipLaunchKernelGGL((gg_im2col_direct_kernel<2>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 5: hipLaunchKernelGGL((gg_im2col_direct_kernel<5>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 258: hipLaunchKernelGGL((gg_im2col_direct_kernel<258>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        default: hipLaunchKernelGGL((gg_im2col_direct_kernel<261>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
      }
      return dg_check_launch();
    }
  }
  static std::atomic<int> occ_cache[2] = {{0}, {0}};
  int occ = occ_cache[lean].load(std::memory_order_relaxed);
  if (!occ) {
    int n = 0;
    const hipError_t e = lean ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gg_im2col_kernel<T, true>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gg_im2col_kernel<T, false>, 256, 0);
    occ = (e == hipSuccess && n > 0) ? n : 2;
    occ_cache[lean].store(occ, std::memory_order_relaxed);
  }
  const int slots = 256 * occ;
  if (tiles >= 8 * slots) tpb = (tiles + slots - 1) / slots;
  static const int tpb_env = getenv("DG_GG_IM2COL_TPB") ? atoi(getenv("DG_GG_IM2COL_TPB")) : 0;
  if (tpb_env > 0) tpb = tpb_env;
  dim3 grid((tiles + tpb - 1) / tpb, (a.Nout + 127) / 128);
  g_last_kinds |= 16;
  if (lean) hipLaunchKernelGGL((gg_im2col_kernel<T, true>), grid, dim3(256), 0, st, a, tpb);
  else hipLaunchKernelGGL((gg_im2col_kernel<T, false>), grid, dim3(256), 0, st, a, tpb);
  return dg_check_launch();
}

// ------------------------------------------------------------------------------------ host side
static int gg_validate(const dg_gg_desc* d, bool f8 = false, bool compact_src = false) {
  if (d->dtype != DG_F32 && d->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  const int epc = f8 ? 16 : d->dtype == DG_F32 ? 4 : 8;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Hg <= 0 || d->Wg <= 0 || d->Hd <= 0 || d->Wd <= 0) return DG_ERR_BAD_SHAPE;
  if (d->Cred <= 0 || d->Cred % 8 || d->Nout <= 0 || d->Nout % 16) return DG_ERR_BAD_SHAPE;
  if (d->ntaps < 1 || d->ntaps > 9) return DG_ERR_BAD_SHAPE;
  // compact_src: the im2col kernel gathers single (channel 0, channel 1) pairs, so its source may be a tensor that stores the
  // real channels only (pixel stride 2: one dword in bf16) instead of the 16-channel padded form -- 8x fewer cache lines per gather
  if ((compact_src ? d->lds % 2 || d->lds < 2 : d->lds % epc) || d->ldd % 4 || d->ldw % epc) return DG_ERR_BAD_SHAPE;
  for (int t = 0; t < d->ntaps; ++t) {
    if (d->tap_dy[t] < -1 || d->tap_dy[t] > 1 || d->tap_dx[t] < -1 || d->tap_dx[t] > 1) return DG_ERR_BAD_ARG;
    if (d->tap_w[t] < 0 || d->tap_w[t] > 8) return DG_ERR_BAD_ARG;
  }
  if (d->src_ps && ((d->Cred / 4) % 8)) return DG_ERR_BAD_SHAPE;
  if (d->dst_ps && ((d->Nout / 4) % 16)) return DG_ERR_BAD_SHAPE;
  if ((long long)d->N * d->Hg * d->Wg >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  // every destination pixel must be inside the destination tensor
  if (d->dst_ps) {
    if (2 * d->Hg > d->Hd || 2 * d->Wg > d->Wd) return DG_ERR_BAD_SHAPE;
  } else {
    if ((d->Hg - 1) * d->dy_mul + d->dy_off >= d->Hd || (d->Wg - 1) * d->dx_mul + d->dx_off >= d->Wd) return DG_ERR_BAD_SHAPE;
    if (d->dy_off < 0 || d->dx_off < 0 || d->dy_mul < 1 || d->dx_mul < 1) return DG_ERR_BAD_SHAPE;
  }
  return DG_OK;
}

template <typename T, int BP, int BC, int WP, int WC>
static int gg_launch_t(GGArgs& a, hipStream_t st) {
  a.nct = (unsigned)((a.Nout + BC - 1) / BC);
  const unsigned npt = (unsigned)((a.M + BP - 1) / BP);
  a.nwg = a.nct * npt;
  static const bool force_generic = getenv("DG_GG_GENERIC") != nullptr;
  const bool fast_ok = a.cch % 8 == 0 && (!a.src_ps || a.cps_src_chunks % 8 == 0) && !force_generic;
  g_last_kinds |= fast_ok ? 2 : 1;
  if (fast_ok)
    hipLaunchKernelGGL((gg_fast_kernel<T, BP, BC, WP, WC>), dim3(a.nwg), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gg_kernel<T, BP, BC, WP, WC>), dim3(a.nwg), dim3(256), 0, st, a);
  return dg_check_launch();
}

template <typename T>
static int gg_launch(GGArgs& a, int N, hipStream_t st) {
  static const bool no_halo = getenv("DG_GG_NOHALO") != nullptr;
  // the patch is sized for tap shifts in [-1, 1] around a unit-stride grid (stride-1 forward, all data gradients)
  static const bool no4w = getenv("DG_GG_NO4W") != nullptr;
  static const bool no4w1 = getenv("DG_GG_NO4W1TAP") != nullptr;
  if (!no_halo && a.sy_mul == 1 && a.sx_mul == 1 && !a.src_ps && a.cch % 8 == 0 && a.Nout > 64 && a.Hg >= 8 && a.Wg >= 8 &&
      (a.ntaps >= 2 || (a.ntaps == 1 && !no4w && !no4w1)) && a.Hs == a.Hg && a.Ws == a.Wg)
  {
    // default: the four-wave kernel, two workgroups per CU (measured +8-22 % over the eight-wave kernel on every layer);
    // it also takes single-tap launches (the 1-tap parity class of a stride-2 data gradient)
#ifdef DG_LC_EXPERIMENT
    if constexpr (sizeof(T) == 2) {
      static const bool lc = getenv("DG_GG_LC") != nullptr;
      if (lc && !no4w && a.cch % 8 == 0) return gg_launch_halo_lc(a, N, st);
    }
#endif
    if (!no4w && a.cch % 8 == 0) return gg_launch_halo4w<T, false>(a, N, st);
    // one reduction block and 2..8 full output-channel tiles: all channel tiles of a pixel tile in one workgroup
    static const bool no_ct = getenv("DG_GG_NOCT") != nullptr;
    if (!no_ct && a.cch == 16 && a.Nout % 128 == 0 && a.Nout >= 256 && a.Nout <= 1024) return gg_launch_halo128<T, 2>(a, N, st);
    if (a.cch % 16 == 0) return gg_launch_halo128<T, 0>(a, N, st);
  }
  // pixel-shuffled sources (data gradients of the up-sampling convs) whose channel quarters hold whole 64-channel blocks
  static const bool no4w_ps = no4w || getenv("DG_GG_NO4WPS") != nullptr;
  if (!no_halo && !no4w_ps && a.sy_mul == 1 && a.sx_mul == 1 && a.src_ps && a.cps_src_chunks % 8 == 0 && a.cch % 8 == 0 && a.Nout > 64 &&
      a.Hg >= 8 && a.Wg >= 8 && a.ntaps >= 2 && a.Hs == a.Hg && a.Ws == a.Wg)
    return gg_launch_halo4w<T, false, true>(a, N, st);
  static const bool no_s2halo = getenv("DG_GG_NOS2HALO") != nullptr;
  if (!no_halo && !no_s2halo && a.sy_mul == 2 && a.sx_mul == 2 && !a.src_ps && a.cch % 16 == 0 && a.Nout > 64 && a.Hg >= 8 &&
      a.Wg >= 8 && a.Hs == 2 * a.Hg && a.Ws == 2 * a.Wg && a.dy_mul == 1 && a.dx_mul == 1)
  {
    GGArgs b = a;
    if (regroup_taps_by_plane(b)) {
      static const bool no4w_s2 = getenv("DG_GG_NO4W") != nullptr || getenv("DG_GG_NO4WS2") != nullptr;
      // >= 256 output channels (bf16): eight waves share one parity-plane patch between two 128-channel halves
      static const bool no8w = getenv("DG_GG_NO8W") != nullptr;
      if constexpr (sizeof(T) == 2) {
        if (!no4w_s2 && !no8w && b.Nout % 256 == 0) return gg_launch_halo4w<T, true, false, 8>(b, N, st);
      }
      if (!no4w_s2) return gg_launch_halo4w<T, true>(b, N, st);
      return gg_launch_halo128<T, 1>(b, N, st);
    }
  }
  static const bool no_halo16 = getenv("DG_GG_NOHALO16") != nullptr;
  if (!no_halo && !no_halo16 && a.Nout <= 16 && a.sy_mul == 1 && a.sx_mul == 1 && !a.src_ps && !a.dst_ps && a.cch % 8 == 0 && a.Hg >= 8 &&
      a.Wg >= 8 && a.ntaps >= 2 && a.Hs == a.Hg && a.Ws == a.Wg && !a.mask_bits && !a.out_bits)
    return gg_launch_halo16<T>(a, N, st);
  if (a.Nout > 64) return gg_launch_t<T, 128, 128, 64, 64>(a, st);
  if (a.Nout > 32) return gg_launch_t<T, 128, 64, 64, 32>(a, st);
  if (a.Nout > 16) return gg_launch_t<T, 128, 32, 32, 32>(a, st);
  return gg_launch_t<T, 128, 16, 32, 16>(a, st);
}

// fp8 launches: only the shapes the four-wave halo kernel takes (the critic's wide layers); everything else is refused
static int gg_launch_f8(GGArgs& a, const F8Args& f, int N, hipStream_t st) {
  if (a.cch % 8 || a.Nout <= 64 || a.Hg < 8 || a.Wg < 8 || a.src_ps || a.dst_ps) return DG_ERR_BAD_SHAPE;
  if (a.sy_mul == 1 && a.sx_mul == 1 && a.Hs == a.Hg && a.Ws == a.Wg) return gg_launch_halo4w_f8<false>(a, f, N, st);
  if (a.sy_mul == 2 && a.sx_mul == 2 && a.Hs == 2 * a.Hg && a.Ws == 2 * a.Wg && a.dy_mul == 1 && a.dx_mul == 1) {
    GGArgs b = a;
    if (regroup_taps_by_plane(b)) {
      static const bool no8w = getenv("DG_GG_NO8W") != nullptr;
      if (!no8w && b.Nout % 256 == 0) return gg_launch_halo4w_f8<true, 8>(b, f, N, st);
      return gg_launch_halo4w_f8<true>(b, f, N, st);
    }
  }
  return DG_ERR_BAD_SHAPE;
}

static int gather_gemm_impl(const dg_gg_desc* d, const dg_epilogue* ep, const void* x, const void* w, void* y, void* stream,
                            bool im2col_small, const dg_f8_operands* f8 = nullptr) {
  if (!d || !x || !w || !y) return DG_ERR_BAD_ARG;
  if (f8 && (d->dtype != DG_BF16 || !f8->xs || !f8->ws || d->Cred % 128)) return DG_ERR_BAD_SHAPE;
  int rc = gg_validate(d, f8 != nullptr, im2col_small);
  if (rc) return rc;
  const int epc = f8 ? 16 : d->dtype == DG_F32 ? 4 : 8;
  GGArgs a{};
  a.x = x; a.w = w; a.y = y;
  a.ldx = d->lds; a.ldw = d->ldw; a.ldy = d->ldd;
  a.M = d->N * d->Hg * d->Wg; a.Hg = d->Hg; a.Wg = d->Wg; a.Hs = d->Hs; a.Ws = d->Ws;
  a.Cred = d->Cred; a.cch = d->Cred / epc; a.ntaps = d->ntaps; a.kchunks = d->ntaps * a.cch;
  a.sy_mul = d->sy_mul; a.sx_mul = d->sx_mul;
  a.tap_lo = 0; a.tap_hi = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    unsigned long long code = (unsigned)(d->tap_dy[t] + 1) | ((unsigned)(d->tap_dx[t] + 1) << 2) | ((unsigned)d->tap_w[t] << 4);
    if (t < 8) a.tap_lo |= code << (8 * t); else a.tap_hi = (unsigned)code;
  }
  a.Nout = d->Nout; a.Hd = d->Hd; a.Wd = d->Wd;
  a.dy_mul = d->dy_mul; a.dx_mul = d->dx_mul; a.dy_off = d->dy_off; a.dx_off = d->dx_off;
  a.src_ps = d->src_ps; a.dst_ps = d->dst_ps;
  a.cps_src_chunks = d->src_ps ? d->Cred / 4 / epc : 1;
  a.cps_dst = d->dst_ps ? d->Nout / 4 : d->Nout;
  a.s1 = a.s2 = 1.f; a.mask_slope = 1.f; a.act_slope = 1.f;
  if (ep) {
    a.bias = ep->bias; a.has_act = ep->has_act; a.act_slope = ep->act_slope;
    a.r1 = ep->r1; a.ldr1 = ep->ldr1; a.s1 = ep->s1;
    a.r2 = ep->r2; a.ldr2 = ep->ldr2; a.s2 = ep->s2;
    a.mask = ep->mask; a.ldmask = ep->ldmask; a.mask_slope = ep->mask_slope;
    a.mask_c0 = ep->mask_c0; a.mask_last = ep->mask_last;
    if (a.mask_c0 < 0 || a.mask_c0 % 16 || ((a.mask_c0 || a.mask_last) && !a.mask)) return DG_ERR_BAD_ARG;
    a.accumulate = ep->accumulate;
    a.mask_bits = ep->mask_bits; a.out_bits = ep->out_bits;
    a.out_q = ep->out_q; a.out_qs = ep->out_qs;
    // the MXFP8 copy is written by the 64-channel wave-tile epilogues of bf16 launches: same shape rules as the bit masks
    if ((a.out_q != nullptr) != (a.out_qs != nullptr)) return DG_ERR_BAD_ARG;
    if (a.out_q && (d->dtype != DG_BF16 || d->Nout < 128 || d->Nout % 64 || d->dst_ps)) return DG_ERR_BAD_SHAPE;
    a.ldqs = ep->ldqs > 0 ? (int)ep->ldqs : d->Nout / 32;
    a.qs_shift = 0;
    if (a.out_q && a.ldqs != d->Nout / 32) {        // strided scale rows: the kernel splits the mask-word index by a shift
      const int n16 = d->Nout / 16;
      if ((n16 & (n16 - 1)) || a.ldqs < d->Nout / 32) return DG_ERR_BAD_SHAPE;
      while ((1 << a.qs_shift) < n16) ++a.qs_shift;
    }
    // bit masks need 64-channel wave tiles (Nout >= 128 selects them in every dispatch path) and plain destinations
    if ((a.mask_bits || a.out_bits) && (d->Nout < 128 || d->Nout % 64 || d->dst_ps || (a.mask_bits && a.mask))) return DG_ERR_BAD_SHAPE;
    if ((a.r1 && a.ldr1 % 4) || (a.r2 && a.ldr2 % 4) || (a.mask && a.ldmask % 4)) return DG_ERR_BAD_SHAPE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (f8) {
    F8Args f{(const unsigned char*)f8->xs, (const unsigned char*)f8->ws, f8->ldxs > 0 ? (int)f8->ldxs : d->Cred / 32};
    if (f.ldxs < d->Cred / 32 || f.ldxs % 4) return DG_ERR_BAD_SHAPE;          // the kernel fetches 4 scale bytes per pixel and K-step as one dword
    return gg_launch_f8(a, f, d->N, st);
  }
  static const bool no_im2col = getenv("DG_GG_NOIM2COL") != nullptr;
  if (im2col_small && !no_im2col) return d->dtype == DG_F32 ? gg_launch_im2col<float>(a, st) : gg_launch_im2col<bf16_t>(a, st);
  return d->dtype == DG_F32 ? gg_launch<float>(a, d->N, st) : gg_launch<bf16_t>(a, d->N, st);
}

extern "C" int dg_gather_gemm(const dg_gg_desc* d, const dg_epilogue* ep, const void* x, const void* w,
                              void* y, void* stream) {
  return gather_gemm_impl(d, ep, x, w, y, stream, false);
}

static int geom_validate(const dg_conv_geom* g) {
  if (!g) return DG_ERR_BAD_ARG;
  if (g->dtype != DG_F32 && g->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  if (g->N <= 0 || g->H <= 0 || g->W <= 0) return DG_ERR_BAD_SHAPE;
  if (g->stride != 1 && g->stride != 2) return DG_ERR_BAD_SHAPE;
  if (g->stride == 2 && ((g->H | g->W) & 1)) return DG_ERR_BAD_SHAPE;
  if (g->Cin <= 0 || g->Cin % 8 || g->Cout <= 0 || g->Cout % 16) return DG_ERR_BAD_SHAPE;
  if (g->pixel_shuffle && (g->stride != 1 || (g->Cout / 4) % 16)) return DG_ERR_BAD_SHAPE;
  return DG_OK;
}

extern "C" int dg_conv3x3_plan(const dg_conv_geom* g, int kind, dg_gg_desc* out) {
  int rc = geom_validate(g);
  if (rc) return rc;
  if (!out || (kind != 0 && kind != 1)) return DG_ERR_BAD_ARG;
  const int Ho = g->H / g->stride, Wo = g->W / g->stride;
  if (kind == 0) {
    dg_gg_desc d{};
    d.dtype = g->dtype; d.N = g->N; d.Hs = g->H; d.Ws = g->W; d.Cred = g->Cin; d.lds = g->ldx; d.src_ps = 0;
    d.Hg = Ho; d.Wg = Wo; d.sy_mul = g->stride; d.sx_mul = g->stride;
    d.ntaps = 9;
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s) { d.tap_dy[r * 3 + s] = r - 1; d.tap_dx[r * 3 + s] = s - 1; d.tap_w[r * 3 + s] = r * 3 + s; }
    d.Nout = g->Cout; d.ldw = 9ll * g->Cin;
    d.ldd = g->ldy; d.dy_mul = d.dx_mul = 1; d.dy_off = d.dx_off = 0;
    if (g->pixel_shuffle) { d.dst_ps = 1; d.Hd = 2 * Ho; d.Wd = 2 * Wo; }
    else { d.dst_ps = 0; d.Hd = Ho; d.Wd = Wo; }
    out[0] = d;
    return 1;
  }
  // data gradient: source = dy over the Ho x Wo output grid, destination = dx over H x W
  int nd = 0;
  const int st = g->stride;
  for (int ph = 0; ph < st; ++ph)
    for (int pw = 0; pw < st; ++pw) {
      dg_gg_desc d{};
      d.dtype = g->dtype; d.N = g->N; d.Hs = Ho; d.Ws = Wo; d.Cred = g->Cout; d.lds = g->ldy;
      d.src_ps = g->pixel_shuffle;
      d.Hg = g->H / st; d.Wg = g->W / st; d.sy_mul = 1; d.sx_mul = 1;
      d.ntaps = 0;
      for (int r = 0; r < 3; ++r) {
        if ((ph + 1 - r) % st) continue;
        for (int s = 0; s < 3; ++s) {
          if ((pw + 1 - s) % st) continue;
          // ho = (hi + 1 - r)/st with hi = gy*st + ph
          d.tap_dy[d.ntaps] = (ph + 1 - r) / st; d.tap_dx[d.ntaps] = (pw + 1 - s) / st;
          d.tap_w[d.ntaps] = r * 3 + s;
          ++d.ntaps;
        }
      }
      d.Nout = g->Cin; d.ldw = 9ll * g->Cout;
      d.Hd = g->H; d.Wd = g->W; d.ldd = g->ldx;
      d.dy_mul = d.dx_mul = st; d.dy_off = ph; d.dx_off = pw; d.dst_ps = 0;
      out[nd++] = d;
    }
  return nd;
}

extern "C" int dg_conv3x3_fwd(const dg_conv_geom* g, const dg_epilogue* ep, const void* x, const void* w_fwd,
                              void* y, void* stream) {
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 0, d);
  if (n < 0) return n;
  const bool small = g->cin_real > 0 && g->cin_real <= 2 && g->stride == 1 && !g->pixel_shuffle;
  return gather_gemm_impl(&d[0], ep, x, w_fwd, y, stream, small);
}

extern "C" int dg_conv3x3_dgrad(const dg_conv_geom* g, const dg_epilogue* ep, const void* dy, const void* w_dgrad,
                                void* dx, void* stream) {
  if (g && g->Cin % 16) return DG_ERR_BAD_SHAPE;  // dx channels are a GEMM N dimension
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 1, d);
  if (n < 0) return n;
  for (int i = 0; i < n; ++i) {
    int rc = dg_gather_gemm(&d[i], ep, dy, w_dgrad, dx, stream);
    if (rc) return rc;
  }
  return DG_OK;
}

// MXFP8 operands (csrc/quant.hip), bf16 output / epilogue tensors.  q->ldxq replaces the source's pixel stride of `g`.
extern "C" int dg_conv3x3_fwd_f8(const dg_conv_geom* g, const dg_epilogue* ep, const dg_f8_operands* q, void* y, void* stream) {
  if (!q || !g || g->dtype != DG_BF16 || g->pixel_shuffle || g->Cin % 128) return DG_ERR_BAD_SHAPE;
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 0, d);
  if (n < 0) return n;
  d[0].lds = q->ldxq;
  return gather_gemm_impl(&d[0], ep, q->xq, q->wq, y, stream, false, q);
}

extern "C" int dg_conv3x3_dgrad_f8(const dg_conv_geom* g, const dg_epilogue* ep, const dg_f8_operands* q, void* dx, void* stream) {
  if (!q || !g || g->dtype != DG_BF16 || g->pixel_shuffle || g->Cout % 128 || g->Cin % 16) return DG_ERR_BAD_SHAPE;
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 1, d);
  if (n < 0) return n;
  for (int i = 0; i < n; ++i) {
    d[i].lds = q->ldxq;
    int rc = gather_gemm_impl(&d[i], ep, q->xq, q->wq, dx, stream, false, q);
    if (rc) return rc;
  }
  return DG_OK;
}

extern "C" int dg_last_conv_kernels(void) { return g_last_kinds; }
