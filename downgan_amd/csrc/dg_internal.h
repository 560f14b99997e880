// Internal helpers shared by the HIP translation units of libdowngan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/downgan_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int EPC = 4; static constexpr int id = DG_F32; };
template <> struct DT<bf16_t> { static constexpr int EPC = 8; static constexpr int id = DG_BF16; };

__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
// round-to-nearest-even on the gfx950 converter (v_cvt_pk_bf16_f32: two values per instruction; the integer trick costs
// ~7 VALU operations per value, which made the store-bound epilogues VALU-bound)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ float ld_elem(const float* p) { return *p; }
__device__ __forceinline__ float ld_elem(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void st_elem(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_elem(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// 4 consecutive elements <-> float[4]
__device__ __forceinline__ void ld4(const float* p, float* v) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float* v) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float* v) {
  uint2 t;
  t.x = pack_bf16x2(v[0], v[1]);
  t.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float leaky_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16 fp32 values scaled by `inv` (a power of two), saturated at +-448 and rounded to nearest even -> 16 OCP E4M3 bytes
typedef __attribute__((ext_vector_type(4))) unsigned int dg_u32x4_t;
// v_cvt_pk_fp8_f32 does not saturate by itself (>= 480 and Inf -> NaN, tools/fp8_cvt_probe.hip) -- unless the wave's MODE.FP16_OVFL bit is
// set: then every finite overflow becomes +-448 (0x7e / 0xfe), Inf and NaN stay NaN (tools/fp8_cvt_probe2.hip).  pack_fp8x16 leaves the
// clamp -- one v_med3 per value, a sixth of an fp8 epilogue's VALU work -- to the converter, so every caller brackets its packs with
// DG_FP8_SAT_ON / DG_FP8_SAT_OFF.  The bit must NOT be left on: under it the MFMAs (bf16 and scaled fp8 alike) turn a NaN in an
// operand or in the accumulator input into a number and an Inf into 3.4e38 (tools/fp8_cvt_probe3.hip) -- a NaN bias no longer reached
// the output -- and v_cvt_pk_bf16_f32 rounds a finite |x| >= 3.39e38 to the largest finite bf16 instead of Inf.
// The macros tie the mode switch to the data: ON rewrites values every pack below depends on, OFF the packed results, so the
// compiler cannot move a conversion out of the bracket (an asm volatile alone orders memory operations, not VALU ones).
// (kernels without MFMAs or bf16 conversions -- the stand-alone quantisers -- switch it on once at their entry)
__device__ __forceinline__ void dg_fp8_saturate_whole_kernel() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 1" ::: "memory"); }
#define DG_FP8_SAT_ON(dep0, dep1) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 1" : "+v"(dep0), "+v"(dep1))
#define DG_FP8_SAT_OFF(res0, res1) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0\n\ts_nop 1" : "+v"(res0), "+v"(res1))
__device__ __forceinline__ dg_u32x4_t pack_fp8x16(const float* v, float inv) {     // (between DG_FP8_SAT_ON and DG_FP8_SAT_OFF)
  dg_u32x4_t o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q] * inv, v[4 * q + 1] * inv, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q + 2] * inv, v[4 * q + 3] * inv, w, true);
    o[q] = (unsigned)w;
  }
  return o;
}
// A block that holds a NaN or an Inf is written as 16 x 0x7F (E4M3 NaN): the clamp above would launder a NaN into -448 and an
// Inf into +-448, and every later fp8 layer reads only the quantised copy.  `amax_bits` = unsigned maximum of the block's
// magnitude bit patterns (equal to the bits of the float maximum for finite values; any NaN / Inf pattern is >= 0x7f800000).
// The same 16 values straight from their PACKED bf16 form (the words a conv epilogue has just stored: p0 = values 0-7, p1 = 8-15):
// v_cvt_scalef32_pk_fp8_bf16 divides two bf16 by 2^(E - 127), E = the exponent field of `scale` (mantissa ignored, field 0 = 2^-127),
// and converts -- no unpacking to fp32, no multiply: one VALU slot per two values instead of four.  Under MODE.FP16_OVFL it equals
// pack_fp8x16 of the unpacked values with factor 2^(127 - E) for every finite bf16 pattern whose quotient is representable in fp32
// (tools/fp8_cvt_probe4.hip: all 65536 patterns x six exponents), saturates where that product would have overflowed to Inf (as the
// emulation's clamp does), and turns +-Inf into +-448 -- blocks holding an Inf or a NaN are poisoned by the caller anyway.  Byte order
// and word select as v_cvt_pk_fp8_f32 (tools/fp8_cvt_probe5.hip).
__device__ __forceinline__ dg_u32x4_t pack_fp8x16_from_bf16(const dg_u32x4_t& p0, const dg_u32x4_t& p1, float scale) {
  typedef __attribute__((ext_vector_type(2))) __bf16 dg_bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) short dg_s16x2_t;
  dg_u32x4_t o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned d0 = j < 2 ? p0[2 * j] : p1[2 * j - 4], d1 = j < 2 ? p0[2 * j + 1] : p1[2 * j - 3];
    dg_s16x2_t w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(dg_bf16x2_t, d0), scale, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(dg_bf16x2_t, d1), scale, true);
    o[j] = __builtin_bit_cast(unsigned, w);
  }
  return o;
}
__device__ __forceinline__ dg_u32x4_t mx_poison(dg_u32x4_t q, unsigned amax_bits) {
  if (amax_bits >= 0x7f800000u) q = dg_u32x4_t{0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu};
  return q;
}
__device__ __forceinline__ unsigned mx_amax_bits16(const float* v) {
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { const unsigned u = __float_as_uint(v[k]) & 0x7fffffffu; m = u > m ? u : m; }
  return m;
}
// E8M0 scale byte of an MX block with maximum magnitude amax: 2^(floor(log2 amax) - 8) (8 = emax of E4M3), biased by 127
__device__ __forceinline__ int mx_scale_byte(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
  return e < 0 ? 0 : e;
}
__device__ __forceinline__ float mx_inv_scale(int e) { return __uint_as_float((unsigned)(254 - e) << 23); }   // 2^(127 - e), exact

static inline int dg_check_launch() { return hipGetLastError() == hipSuccess ? DG_OK : DG_ERR_LAUNCH; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: a launcher that set it once per process would launch
// > 64 KB of dynamic LDS without it on a second device of the same process.  Bit d of `done` = "device d is configured";
// two threads racing through the first call both run the (idempotent) setter.  The only mutable state of the library are
// these per-kernel device masks and the cached occupancy answers (std::atomic, same value on every MI355X).
static inline int dg_set_max_lds_once(std::atomic<unsigned long long>& done, const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DG_ERR_LAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return DG_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return DG_ERR_LAUNCH;
  done.fetch_or(bit, std::memory_order_release);
  return DG_OK;
}
#define DG_SET_MAX_LDS_ONCE(kernel, bytes)                                                                        \
  do {                                                                                                            \
    static std::atomic<unsigned long long> dg_done_{0};                                                           \
    if (dg_set_max_lds_once(dg_done_, reinterpret_cast<const void*>(kernel), (bytes)) != DG_OK) return DG_ERR_LAUNCH; \
  } while (0)

// Deterministic-reduction mode (debug.hip, dg_set_deterministic_workspace): a launch whose workgroups accumulate partial results
// into one fp32 target region takes a plan of `copies` zero-filled copies of that region inside the caller's workspace, points
// workgroup / split / wave c at copy c (DG_DET_PTR: same address arithmetic, other base) and has the copies added to the target in
// index order afterwards.  ws == nullptr: mode off, or a single copy (unique writers): accumulate into the target directly.
struct DetPlan { float* ws; long long stride; int copies; };
bool dg_det_on();
int dg_det_begin(long long stride, int want, hipStream_t st, DetPlan* p);
int dg_det_reduce(const DetPlan& p, long long off, float* target, long long n, hipStream_t st);

// bijective XCD-aware remap of a linear block id: blocks that share an XCD (id % 8 equal) get a
// contiguous range of logical tiles, so neighbouring tiles share that XCD's L2 (speed only).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
