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

// ---- per-block maximum of the MXFP8 scale bytes of a tensor (rows x nblocks bytes): the exponents of its uniform-scale copy.
// Threads walk dwords of the byte matrix (nblocks is a multiple of 4: channel counts are multiples of 128) with a running
// byte-wise maximum, then one atomicMax per block into the caller's 64-dword scratch (zero at rest: the finish kernel clears it).
__global__ void __launch_bounds__(256) block_exp_max_kernel(const unsigned* __restrict__ sc, long long rows, long long ld4, int nb4,
                                                            unsigned* __restrict__ tmp) {
  const int col = threadIdx.x % nb4, sub = threadIdx.x / nb4, nsub = 256 / nb4;
  if (sub >= nsub) return;
  unsigned m = 0;
  for (long long r = (long long)blockIdx.x * nsub + sub; r < rows; r += (long long)gridDim.x * nsub) {
    const unsigned v = sc[r * ld4 + col];
    unsigned r4 = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) { const unsigned x = (v >> (8 * b)) & 0xffu, y = (m >> (8 * b)) & 0xffu; r4 |= (x > y ? x : y) << (8 * b); }
    m = r4;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) atomicMax(tmp + col * 4 + b, (m >> (8 * b)) & 0xffu);
}
__global__ void block_exp_finish_kernel(unsigned* __restrict__ tmp, int nblocks, int margin, unsigned char* __restrict__ out) {
  const int b = threadIdx.x;
  if (b < nblocks) { const unsigned v = tmp[b] + (unsigned)margin; out[b] = (unsigned char)(v > 254u ? 254u : v); tmp[b] = 0u; }
}

extern "C" int dg_block_exp_max(const void* scales, int64_t rows, int64_t ld, int nblocks, int margin, void* out, void* scratch, void* stream) {
  if (!scales || !out || !scratch) return DG_ERR_BAD_ARG;
  if (rows <= 0 || nblocks <= 0 || nblocks > 64 || nblocks % 4 || ld < nblocks || ld % 4 || margin < 0 || margin > 8) return DG_ERR_BAD_SHAPE;
  if (reinterpret_cast<uintptr_t>(scales) % 4 || reinterpret_cast<uintptr_t>(scratch) % 4) return DG_ERR_BAD_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb4 = nblocks / 4, nsub = 256 / nb4;
  long long nwg = (rows + (long long)nsub * 64 - 1) / ((long long)nsub * 64);
  if (nwg > 2048) nwg = 2048;
  if (nwg < 1) nwg = 1;
  hipLaunchKernelGGL(block_exp_max_kernel, dim3((unsigned)nwg), dim3(256), 0, st, (const unsigned*)scales, (long long)rows, (long long)(ld / 4), nb4,
                     (unsigned*)scratch);
  hipLaunchKernelGGL(block_exp_finish_kernel, dim3(1), dim3(64), 0, st, (unsigned*)scratch, nblocks, margin, (unsigned char*)out);
  return dg_check_launch();
}
