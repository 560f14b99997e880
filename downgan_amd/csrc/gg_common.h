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
// This header: launch arguments, epilogue code and MFMA wrappers shared by the kernel families
//   conv_rows.hip     gg_kernel / gg_fast_kernel   row-tiled gather-GEMM (any shape; pixel-shuffled sources, narrow layers)
//   conv_halo.hip     gg_halo4w_kernel             16x16-pixel tiles with the source patch resident in LDS (the dominant kernel)
//   conv_halo_f8.hip  gg_halo4w_f8_kernel          the same on the block-scaled MXFP8 MFMA
//   conv_small.hip    gg_halo16_kernel, gg_im2col_kernel, gg_im2col_direct_kernel   <= 16 output / <= 2 real input channels (HBM-bound)
//   conv_plan.hip     host side: lowering of a conv layer to descriptors, kernel choice, the C ABI entry points
#pragma once
#include "dg_internal.h"

#include <stdlib.h>
#include <type_traits>


// bit mask of the kernel variants the last dg_conv3x3_fwd / _dgrad call of this thread launched (bench.py tags its
// live timings with it): 1 generic, 2 fast, 8 halo, 16 im2col, 32 fp8 halo (defined in conv_plan.hip)
extern thread_local int g_last_kinds;

struct GGArgs {
  const void* x; const void* w; void* y;
  const float* bias; const void* r1; const void* r2; const void* mask;
  const void* mask_bits; void* out_bits;     // 1-bit LeakyReLU masks (u16 per lane: 4 fragments x 4 channels), see dg_epilogue
  void* out_q; void* out_qs;                 // MXFP8 copy of the stored output (dg_epilogue.out_q / out_qs)
  void* out_u; const unsigned char* out_ue;  // uniform-scale E4M3 copy for the fp8 weight gradient (dg_epilogue.out_u / out_ue)
  int no_y;                                  // dg_epilogue.skip_y: the bf16 output itself is not stored (only its fp8 copies / mask bits are read)
  unsigned* out_amax;                        // dg_epilogue.out_amax: per 32-channel block, atomic max of the bit pattern of the largest |stored value| (first-layer kernel)
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
  int seg;                                     // 1: the four parity classes of a stride-2 data gradient in ONE launch (conv_halo.hip, SEG)
  int dbg;                                     // diagnostic (stamp) builds only: ablation bits from DG_ABL (1: epilogue stores dropped, 2: one workgroup per CU)
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
struct EpiRes { __amdgpu_buffer_rsrc_t rY, r1, r2, rm, rbi, rbo, rq, rqs, ru; int ldy, ld1, ld2, ldm; };
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
                                            unsigned offm, unsigned boff, unsigned mb, bool mask_on, unsigned* ob_ret = nullptr,
                                            float inv_u = 0.f, unsigned* ab_run = nullptr) {
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
  const bool f_u = F < 0 ? a.out_u != nullptr : (F & 2048) != 0;                     // + the uniform-scale copy (inv_u = 2^(127 - exponent))
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
      // (the products as float2 pairs: v_pk_mul_f32, one VALU slot per two values)
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const f32x2_t sl2 = {a.act_slope, a.act_slope};
#pragma unroll
      for (int k = 0; k < 16; k += 2) {
        const f32x2_t p = f32x2_t{v[k], v[k + 1]} * sl2;
        v[k] = __builtin_fmaxf(v[k], p[0]); v[k + 1] = __builtin_fmaxf(v[k + 1], p[1]);
      }
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
    pk[u] = IO::pack(vu);
    if (F >= 0 ? !(F & 4096) : !a.no_y) __builtin_amdgcn_raw_buffer_store_b128(pk[u], R.rY, offy, u * 16, 0);
  }
  if (f_ob) {
    // bit k = (v[k] > 0), shifted in from the top: compare into VCC, then ob = 2 * ob + carry -- two instructions per value where
    // the compiler's compare / select / or takes 2.5 (the epilogues are bound by instruction issue)
#pragma unroll
    for (int k = 15; k >= 0; --k)
      asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(ob) : "v"(v[k]) : "vcc");
  }
  if (F >= 0 && (F & 1024)) *ob_ret = ob;      // the caller stores the word itself (gg_im2col_direct_kernel: two words per store)
  else if (F >= 0) { if (f_ob) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ob, R.rbo, boff, 0, 0); }
  else if (LEAN || a.out_bits) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ob, R.rbo, (LEAN && !a.out_bits) ? DG_OOB_OFF : boff, 0, 0);
  if constexpr (sizeof(T) == 2) {
    // MXFP8 copy of what was just stored (the bf16-ROUNDED values, so it equals dg_quant_mxfp8 of the stored tensor): the lane
    // holds 16 consecutive channels, lane ^ 16 the other half of the 32-channel block; one 16-byte store per lane, the scale byte
    // from the lane with the lower half.  q has y's pixel stride (in bytes = elements), scales [pixel][Nout / 32]:
    // byte offsets offy / 2 and boff / 4 ((rel * Nout + channel) / 32).
    if (f_q || f_u) {
      // largest magnitude of the 16 ROUNDED values as an fp32 bit pattern, taken on the packed bf16 words (v_pk_max_u16: one operation
      // per two values; as unsigned 16-bit numbers NaN > Inf > every finite value, so a NaN / Inf in the block dominates: mx_poison)
      typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
      u16x2_t m2 = {0, 0};
#pragma unroll
      for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
        for (int q = 0; q < 4; ++q) m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2_t, pk[u2][q] & 0x7fff7fffu));
      unsigned ab = (unsigned)(m2[0] > m2[1] ? m2[0] : m2[1]) << 16;
      // the other half of the 32-channel block sits in lane ^ 16, i.e. in the neighbouring 16-lane row: v_permlane16_swap exchanges the
      // odd rows of one register with the even rows of another (of a copy here), after which the two registers hold, lane by lane, the
      // pair's two values -- three VALU operations and no trip through the LDS crossbar (ds_bpermute behind __shfl_xor, and its wait)
      const auto sw = __builtin_amdgcn_permlane16_swap(ab, ab, false, false);
      ab = sw[0] > sw[1] ? sw[0] : sw[1];
      // the converter saturates only under MODE.FP16_OVFL (dg_internal.h): on for the packs, off again before anything else runs
      // (packs straight from the packed bf16 words, dg_internal.h pack_fp8x16_from_bf16: the divisor's exponent field is the scale byte;
      // inv_u = 2^(127 - E_u) arrives as the factor the stand-alone quantiser uses)
      float su = __uint_as_float((254u << 23) - __float_as_uint(inv_u));
      u32x4_t qv = {0u, 0u, 0u, 0u}, uv = {0u, 0u, 0u, 0u};
      DG_FP8_SAT_ON(ab, su);
      const int e = mx_scale_byte(__uint_as_float(ab));
      if (f_q) qv = __builtin_bit_cast(u32x4_t, pack_fp8x16_from_bf16(pk[0], pk[1], __uint_as_float((unsigned)e << 23)));
      if (f_u) uv = __builtin_bit_cast(u32x4_t, pack_fp8x16_from_bf16(pk[0], pk[1], su));
      DG_FP8_SAT_OFF(qv, uv);
      if (__builtin_amdgcn_ballot_w64(ab >= 0x7f800000u)) {          // a NaN / Inf somewhere in the wave (rare: one compare + a scalar branch otherwise)
        qv = __builtin_bit_cast(u32x4_t, mx_poison(__builtin_bit_cast(dg_u32x4_t, qv), ab));
        uv = __builtin_bit_cast(u32x4_t, mx_poison(__builtin_bit_cast(dg_u32x4_t, uv), ab));
      }
      if (f_q) {
        __builtin_amdgcn_raw_buffer_store_b128(qv, R.rq, offy == DG_OOB_OFF ? DG_OOB_OFF : offy >> 1, 0, 0);
        const bool low_half = ((threadIdx.x >> 4) & 1) == 0;
        unsigned offqs = boff >> 2;                                   // dense scale rows: (rel * Nout + channel) / 32
        if (a.qs_shift) {                                             // a channel slice of a wider tensor (dense-block slab) / a pixel-shuffled output
          const unsigned w16 = boff >> 1;                             // pixel * (C / 16) + channel / 16, C / 16 a power of two (C = channels per destination pixel)
          offqs = (w16 >> a.qs_shift) * (unsigned)a.ldqs + ((w16 & ((1u << a.qs_shift) - 1u)) >> 1);
        }
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)e, R.rqs, (low_half && boff != DG_OOB_OFF) ? offqs : DG_OOB_OFF, 0, 0);
      }
      if (f_u)       // the same rounded values on the tensor-wide exponent of their 32-channel block (a non-finite block is poisoned too)
        __builtin_amdgcn_raw_buffer_store_b128(uv, R.ru, offy == DG_OOB_OFF ? DG_OOB_OFF : offy >> 1, 0, 0);
      // running maximum of the block magnitudes this lane has stored (dg_epilogue.out_amax; a dropped pixel's values are not stored)
      if (ab_run && offy != DG_OOB_OFF) *ab_run = *ab_run > ab ? *ab_run : ab;
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
    R.ru = rsrc(a.out_u ? a.out_u : a.y, a.ldy, 1);
    const float inv_u = (a.out_u && cok) ? mx_inv_scale((int)a.out_ue[cb16 >> 5]) : 0.f;
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
                           ok ? (unsigned)((a.dst_ps ? pix * (a.cps_dst >> 4) + (cc0 >> 4) : rel * ldb + bidx) * 2) : DG_OOB_OFF, mbv[i],
                           a.mask && cb16 >= a.mask_c0, nullptr, inv_u);
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

// ---- Epilogue of the halo kernels.  The lane's 16 fragments (pixel fragment i = tile row wp*4+i, channel
// fragment j) sit at (relative pixel of i + pixel-shuffle offset of j) * ld + channel: every tensor is addressed
// through a raw buffer descriptor based at the workgroup's first destination pixel plus 32-bit per-lane offsets
// (one multiply-add per fragment and tensor instead of 64-bit index arithmetic), and out-of-tile / out-of-range
// fragments get an out-of-range offset: the hardware drops those stores and returns zeros for those loads, so the
// epilogue has no divergent branches.
// oy_o / ox_o >= 0: destination offsets of this workgroup instead of a.dy_off / a.dx_off (the parity class of a merged stride-2
// data-gradient launch, conv_halo.hip SEG).
// PREB: the caller's accumulators were initialised with the bias (the halo kernel loads it in its prologue, where the latency
// hides behind the first patch; loading it here put one global round trip per 64-channel half into every tile's epilogue).
template <typename T, int NH, bool PREB = false>
__device__ __forceinline__ void halo_epilogue(const GGArgs& a, f32x4_t (&acc)[4 * NH][4], int img, int ty0, int tx0, int c0, int wp,
                                              int wc, int l15, int g, int oy_o = -1, int ox_o = -1, unsigned long long* est = nullptr) {
#ifdef DG_STAMP
#define EPI_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); est[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define EPI_STAMP(i) do { } while (0)
#endif
  constexpr int ES = (int)sizeof(T);
  EPI_STAMP(0);
  const int psm = a.dst_ps ? 2 : a.dy_mul, psx = a.dst_ps ? 2 : a.dx_mul;
  const int oy = oy_o >= 0 ? oy_o : (a.dst_ps ? 0 : a.dy_off), ox = ox_o >= 0 ? ox_o : (a.dst_ps ? 0 : a.dx_off);
  // workgroup base pixel (scalar) and this lane's relative pixel for tile row wp*4 (+ i rows of pitch `rowp`)
  const long long pb = ((long long)img * a.Hd + (long long)ty0 * psm + oy) * a.Wd + (long long)tx0 * psx + ox;
  const int rel0 = (wp * 4) * psm * a.Wd + l15 * psx;
  const int rowp = psm * a.Wd;
  auto rsrc = [&](const void* p, long long ld, int es) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + pb * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
  };
  const int ldb = (a.Nout >> 6) * 4;
#ifdef DG_STAMP
  const bool xok = tx0 + l15 < a.Wg && !(a.dbg & 1);       // ablation: every epilogue store (and operand load) out of range = dropped
#else
  const bool xok = tx0 + l15 < a.Wg;
#endif
  // every mask word of the wave's tile first (NH halves x 4 rows), before the first store
  unsigned mbv[NH][4];
  if (a.mask_bits) {
    const __amdgpu_buffer_rsrc_t rbi = rsrc(a.mask_bits, ldb, 2);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int cb64 = c0 + (wc + h) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = xok && cb64 + 16 * g < a.Nout && ty0 + wp * 4 + i < a.Hg;
        mbv[h][i] = __builtin_amdgcn_raw_buffer_load_b16(rbi, ok ? (unsigned)(((rel0 + i * rowp) * ldb + (cb64 >> 6) * 4 + g) * 2) : DG_OOB_OFF, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) mbv[h][i] = 0u;
  }
  EPI_STAMP(1);
  auto run = [&](auto tag, auto htag) {
  constexpr int F = decltype(tag)::value;
  constexpr int h = decltype(htag)::value;          // compile-time: acc[] must never be indexed dynamically
  {
    // the descriptors are built HERE, per straight-line instance: each instance computes only the ones its flags read (built
    // ahead of the dispatch all eight -- 64-bit multiplies each -- sat in front of every epilogue: ~1000 of its ~1800 set-up cycles)
    EpiRes R;
    R.rY = rsrc(a.y, a.ldy, ES);
    R.r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1, ES); R.r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2, ES); R.rm = rsrc(a.mask ? a.mask : a.y, a.ldmask, ES);
    R.rbi = R.rY; R.rbo = rsrc(a.out_bits ? a.out_bits : a.y, ldb, 2);
    R.rq = rsrc(a.out_q ? a.out_q : a.y, a.ldy, 1); R.rqs = rsrc(a.out_qs ? a.out_qs : a.y, a.ldqs, 1);
    R.ru = rsrc(a.out_u ? a.out_u : a.y, a.ldy, 1);
    R.ldy = (int)a.ldy; R.ld1 = (int)a.ldr1; R.ld2 = (int)a.ldr2; R.ldm = (int)a.ldmask;
    // permuted wave tile (perm64): the lane's four channel fragments are the 16 consecutive channels from cb16
    const int cb16 = c0 + (wc + h) * 64 + 16 * g;
    const bool cok = cb16 < a.Nout && xok;
    float inv_u = 0.f;
    if ((F < 0 || (F & 2048)) && a.out_u && cb16 < a.Nout) inv_u = mx_inv_scale((int)a.out_ue[cb16 >> 5]);
    float bias[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!PREB) { if (a.bias && cb16 < a.Nout) b4 = *reinterpret_cast<const float4*>(a.bias + cb16 + 4 * q); }
      bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
    }
    int cc0 = cb16, pj0 = 0;
    if (a.dst_ps) { const int q = cb16 / a.cps_dst; cc0 = cb16 - q * a.cps_dst; pj0 = (q >> 1) * a.Wd + (q & 1); }
    const int bidx = ((c0 + (wc + h) * 64) >> 6) * 4 + g;
    constexpr int FE = (PREB && F >= 0) ? (F | 512) : F;      // 512: the accumulators already hold the bias
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cok && ty0 + wp * 4 + i < a.Hg;
      const int rel = rel0 + i * rowp;
      const int pix = rel + pj0;
      epi64_pixel<T, (F >= 0), FE>(a, R, acc[4 * h][i], acc[4 * h + 1][i], acc[4 * h + 2][i], acc[4 * h + 3][i], bias,
                            ok ? (unsigned)((pix * R.ldy + cc0) * ES) : DG_OOB_OFF, ok ? (unsigned)((pix * R.ld1 + cc0) * ES) : DG_OOB_OFF,
                            ok ? (unsigned)((pix * R.ld2 + cc0) * ES) : DG_OOB_OFF, ok ? (unsigned)((pix * R.ldm + cc0) * ES) : DG_OOB_OFF,
                            // (pixel-shuffled output: no bit masks; the MXFP8 scale byte's (destination pixel, 16-channel group) index)
                            ok ? (unsigned)((a.dst_ps ? pix * (a.cps_dst >> 4) + (cc0 >> 4) : rel * ldb + bidx) * 2) : DG_OOB_OFF, mbv[h][i],
                            a.mask && cb16 >= a.mask_c0, nullptr, inv_u);
    }
  }
  };
  // the flag combinations the train step launches most get straight-line instances; everything else the general one.
  // Decoded per 64-channel half: the activation mask may start at channel mask_c0 (a multiple of 64 here, else general path)
  const int key0 = (a.has_act ? 1 : 0) | (a.mask_bits ? 2 : 0) | (a.out_bits ? 4 : 0) | (a.r1 ? 8 : 0) | (a.r2 ? 16 : 0) | (a.accumulate ? 64 : 0) |
                   (a.out_q ? 256 : 0) | (a.out_u ? 2048 : 0) | (a.no_y ? 4096 : 0);
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
      case 288: run(std::integral_constant<int, 288>{}, htag); break;  // fp8 mode: activation mask + MXFP8 copy (generator dense-block data gradients)
      case 2305: run(std::integral_constant<int, 2305>{}, htag); break; // 257 / 264 / 280 / 288 + the uniform-scale copy (+ 2048) for the dense blocks' fp8 weight gradient
      case 2312: run(std::integral_constant<int, 2312>{}, htag); break;
      case 2328: run(std::integral_constant<int, 2328>{}, htag); break;
      case 2336: run(std::integral_constant<int, 2336>{}, htag); break;
      case 258: run(std::integral_constant<int, 258>{}, htag); break;  // fp8 mode: 1-bit mask + MXFP8 copy (critic data gradients, tangent forward)
      case 261: run(std::integral_constant<int, 261>{}, htag); break;  // fp8 mode: LeakyReLU + out_bits + MXFP8 copy (critic forward)
      case 2306: run(std::integral_constant<int, 2306>{}, htag); break; // ... + the uniform-scale copy for the fp8 weight gradient (258 + 2048)
      case 2309: run(std::integral_constant<int, 2309>{}, htag); break; // (261 + 2048)
      case 4353: run(std::integral_constant<int, 4353>{}, htag); break; // 257 without the bf16 store (up-sampling convs of a forward nobody differentiates)
      case 4354: run(std::integral_constant<int, 4354>{}, htag); break; // 258 / 261 / 2306 / 2309 without the bf16 store (+ 4096: dg_epilogue.skip_y)
      case 4357: run(std::integral_constant<int, 4357>{}, htag); break;
      case 6402: run(std::integral_constant<int, 6402>{}, htag); break;
      case 6405: run(std::integral_constant<int, 6405>{}, htag); break;
      default: run(std::integral_constant<int, -1>{}, htag); break;
    }
  };
  dispatch(std::integral_constant<int, 0>{});
  EPI_STAMP(2);
  if constexpr (NH == 2) dispatch(std::integral_constant<int, 1>{});
  EPI_STAMP(3);
#undef EPI_STAMP
}

struct F8Args { const unsigned char* xs; const unsigned char* ws; int ldxs; };   // ldxs: scale bytes per source pixel (Cred/32 unless the source is a slab slice); weights: 9*Cred/32 per row

// ---- launchers of the kernel families (one translation unit each); dtype = DG_F32 / DG_BF16, N = images
int gg_launch_rows(GGArgs& a, int dtype, hipStream_t st);
bool gg_regroup_taps_by_plane(GGArgs& a);
bool gg_halo_row_step_ok(long long Ws, long long ldx, int dtype, int mult);
int gg_launch_halo(GGArgs& a, int dtype, int N, bool s2, bool ps, int nw, hipStream_t st);
int gg_launch_halo_f8(GGArgs& a, const F8Args& f, int N, bool s2, int nw, hipStream_t st);
int gg_launch_halo16(GGArgs& a, int dtype, int N, hipStream_t st);
int gg_launch_im2col(GGArgs& a, int dtype, hipStream_t st);
