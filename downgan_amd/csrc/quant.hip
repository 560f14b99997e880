// MXFP8 quantisation for the fp8 conv path (BASELINE.json configs[4]; reference math: DoWnGAN/networks/critic.py:20-88).
//
// Element format OCP FP8 E4M3 (gfx950's native fp8), one shared E8M0 scale (a power of two) per block of 32 CONSECUTIVE
// reduction channels (the OCP MX layout) -- the operand format of v_mfma_scale_f32_16x16x128_f8f6f4.  How the instruction
// maps registers to K (measured with tools/fp8_probe2.hip, not documented in the guides): lane (row r = l & 15, group
// g = l >> 4) holds 32 operand bytes; byte j is K index 64 * (j / 16) + 16 * g + (j % 16), and the scale of K block kb
// (= K / 32) of row r is byte `opsel` of the scale register of lane r + 16 * kb.  So a lane that loads the 16-byte chunk g of
// the first and of the second 64-byte half of a 128-channel row -- what the bf16 conv kernel's two fragment reads do -- feeds
// channels 16g.. and 64 + 16g.. in place, the four scale blocks are channels [0,32) [32,64) [64,96) [96,128) of the row, and
// lane (r, g) supplies the scale of block g.
// scale byte s: value 2^(s - 127); chosen as 2^(floor(log2(amax)) - 8) (8 = emax of E4M3, OCP MX rule); elements are
// x / scale rounded to nearest even, saturated at +-448.  A block that holds a NaN or an Inf becomes 32 x NaN (0x7F), so a
// non-finite activation or adjoint stays visible down the fp8 chain instead of being clamped into a finite value.
#include "dg_internal.h"

typedef __attribute__((ext_vector_type(4))) unsigned int q_u32x4_t;

template <typename T> struct QLoad;
template <> struct QLoad<bf16_t> {   // 16 channels = 32 bytes
  static __device__ __forceinline__ void run(const bf16_t* p, float* v) {
    const q_u32x4_t a = *reinterpret_cast<const q_u32x4_t*>(p), b = *reinterpret_cast<const q_u32x4_t*>(p + 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[2 * q] = __uint_as_float(a[q] << 16); v[2 * q + 1] = __uint_as_float(a[q] & 0xffff0000u);
      v[8 + 2 * q] = __uint_as_float(b[q] << 16); v[9 + 2 * q] = __uint_as_float(b[q] & 0xffff0000u);
    }
  }
};
template <> struct QLoad<float> {
  static __device__ __forceinline__ void run(const float* p, float* v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = reinterpret_cast<const float4*>(p)[q];
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
};

// one thread = one block of one row: rows x (C / 32) threads
template <typename T>
__global__ void __launch_bounds__(256) quant_mxfp8_kernel(const T* __restrict__ src, long long rows, long long ld, int C,
                                                          unsigned char* __restrict__ q, long long ldq, unsigned char* __restrict__ sc,
                                                          long long ldqs) {
  const int nb = C >> 5;
  const long long total = rows * nb;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / nb;
    const int b = (int)(t - r * nb);
    const int c_lo = b * 32;
    float v[32];
    QLoad<T>::run(src + r * ld + c_lo, v);
    QLoad<T>::run(src + r * ld + c_lo + 16, v + 16);
    const unsigned a0 = mx_amax_bits16(v), a1 = mx_amax_bits16(v + 16);
    const unsigned ab = a0 > a1 ? a0 : a1;                       // a NaN / Inf anywhere in the block dominates
    const int e = mx_scale_byte(__uint_as_float(ab));
    const float inv = mx_inv_scale(e);
    *reinterpret_cast<q_u32x4_t*>(q + r * ldq + c_lo) = mx_poison(pack_fp8x16(v, inv), ab);
    *reinterpret_cast<q_u32x4_t*>(q + r * ldq + c_lo + 16) = mx_poison(pack_fp8x16(v + 16, inv), ab);
    sc[r * ldqs + b] = (unsigned char)e;
  }
}

extern "C" int dg_quant_mxfp8(int src_dtype, const void* src, int64_t rows, int64_t ld, int C, void* q, int64_t ldq, void* scales,
                              int64_t ldqs, void* stream) {
  if (ldqs <= 0) ldqs = C / 32;
  if (!src || !q || !scales || rows <= 0 || C <= 0 || C % 128 || ld < C || ldq < C || ldq % 16 || ldqs < C / 32) return DG_ERR_BAD_SHAPE;
  if ((src_dtype == DG_BF16 && ld % 8) || (src_dtype == DG_F32 && ld % 4)) return DG_ERR_BAD_SHAPE;
  const long long total = (long long)rows * (C / 32);
  long long nb = (total + 255) / 256;
  if (nb > 65536) nb = 65536;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (src_dtype == DG_BF16)
    hipLaunchKernelGGL(quant_mxfp8_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)src, (long long)rows, (long long)ld, C,
                       (unsigned char*)q, (long long)ldq, (unsigned char*)scales, (long long)ldqs);
  else if (src_dtype == DG_F32)
    hipLaunchKernelGGL(quant_mxfp8_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)src, (long long)rows, (long long)ld, C,
                       (unsigned char*)q, (long long)ldq, (unsigned char*)scales, (long long)ldqs);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
