// Frequency separation (DoWnGAN/GAN/wasserstein_fs.py:36-46,73-86; hyperparams.py:31-35):
//   low(x) = AvgPool2d(5, stride 1, padding 0)(ReplicationPad2d(2)(x)),  high(x) = x - low(x)
// on native NHWC tensors, plus the adjoint low^T needed by the generator's backward (the reference gets it from autograd).
// HBM/L1-bound stencils: one 16-byte channel chunk per thread, 25 clamped neighbours summed in fp32 in window order
// (row-major, like ATen's avg_pool2d) and divided by 25.
#include "dg_internal.h"

template <typename T> __device__ __forceinline__ void ldc_fs(const T* p, float* v);
template <> __device__ __forceinline__ void ldc_fs<float>(const float* p, float* v) { ld4(p, v); }
template <> __device__ __forceinline__ void ldc_fs<bf16_t>(const bf16_t* p, float* v) { ld4(p, v); ld4(p + 4, v + 4); }
template <typename T> __device__ __forceinline__ void stc_fs(T* p, const float* v);
template <> __device__ __forceinline__ void stc_fs<float>(float* p, const float* v) { st4(p, v); }
template <> __device__ __forceinline__ void stc_fs<bf16_t>(bf16_t* p, const float* v) { st4(p, v); st4(p + 4, v + 4); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T>
__global__ void lowpass5_kernel(const T* x, long long ldx, int N, int H, int W, int cch, T* low, long long ldl, T* high, long long ldh) {
  constexpr int EPC = DT<T>::EPC;
  const long long total = (long long)N * H * W * cch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cch) * EPC;
    const long long p = i / cch;
    const int xw = (int)(p % W);
    const long long t = p / W;
    const int yh = (int)(t % H);
    const long long n = t / H;
    const T* img = x + n * H * W * ldx + c;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
      const int yy = clampi(yh + dy, 0, H - 1);
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        const int xx = clampi(xw + dx, 0, W - 1);
        float v[EPC];
        ldc_fs<T>(img + ((long long)yy * W + xx) * ldx, v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] += v[e];
      }
    }
    float ctr[EPC];
    ldc_fs<T>(img + ((long long)yh * W + xw) * ldx, ctr);
    float lo[EPC], hi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { lo[e] = s[e] / 25.f; hi[e] = ctr[e] - lo[e]; }
    if (low) stc_fs<T>(low + p * ldl + c, lo);
    if (high) stc_fs<T>(high + p * ldh + c, hi);
  }
}

// out[q] = (1/25) * sum over p of g[p] * #{(dy,dx) in [-2,2]^2 : clamp(p + d) == q}: interior pixels gather their 5x5
// neighbourhood, border pixels also collect what replication padding folded onto them.
__device__ __forceinline__ int fold_count(int q, int p, int L) {
  // number of offsets d in [-2, 2] with clamp(p + d, 0, L-1) == q
  if (q > 0 && q < L - 1) return (p - q <= 2 && q - p <= 2) ? 1 : 0;
  int cnt = 0;
#pragma unroll
  for (int d = -2; d <= 2; ++d) cnt += (clampi(p + d, 0, L - 1) == q) ? 1 : 0;
  return cnt;
}
template <typename T>
__global__ void lowpass5_adjoint_kernel(const T* g, long long ldg, int N, int H, int W, int cch, T* out, long long ldo) {
  constexpr int EPC = DT<T>::EPC;
  const long long total = (long long)N * H * W * cch;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cch) * EPC;
    const long long p = i / cch;
    const int qx = (int)(p % W);
    const long long t = p / W;
    const int qy = (int)(t % H);
    const long long n = t / H;
    const T* img = g + n * H * W * ldg + c;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    const int y0 = qy - 2 < 0 ? 0 : qy - 2, y1 = qy + 2 > H - 1 ? H - 1 : qy + 2;
    const int x0 = qx - 2 < 0 ? 0 : qx - 2, x1 = qx + 2 > W - 1 ? W - 1 : qx + 2;
    for (int py = y0; py <= y1; ++py) {
      const int wy = fold_count(qy, py, H);
      for (int px = x0; px <= x1; ++px) {
        const float w = (float)(wy * fold_count(qx, px, W));
        float v[EPC];
        ldc_fs<T>(img + ((long long)py * W + px) * ldg, v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] += w * v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] /= 25.f;
    stc_fs<T>(out + p * ldo + c, s);
  }
}

static inline unsigned fs_blocks(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b > 65535 ? 65535 : (b < 1 ? 1 : b));
}
extern "C" int dg_lowpass5(int dtype, const void* x, int64_t ldx, int N, int H, int W, int C, void* low, int64_t ldl, void* high,
                           int64_t ldh, void* stream) {
  if (!x || (!low && !high) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || ldx < C || (low && ldl < C) || (high && ldh < C))
    return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) {
    const int cch = C / 4;
    hipLaunchKernelGGL(lowpass5_kernel<float>, dim3(fs_blocks((long long)N * H * W * cch)), dim3(256), 0, st, (const float*)x, (long long)ldx, N, H, W, cch, (float*)low, (long long)ldl, (float*)high, (long long)ldh);
  } else if (dtype == DG_BF16) {
    const int cch = C / 8;
    hipLaunchKernelGGL(lowpass5_kernel<bf16_t>, dim3(fs_blocks((long long)N * H * W * cch)), dim3(256), 0, st, (const bf16_t*)x, (long long)ldx, N, H, W, cch, (bf16_t*)low, (long long)ldl, (bf16_t*)high, (long long)ldh);
  } else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
extern "C" int dg_lowpass5_adjoint(int dtype, const void* g, int64_t ldg, int N, int H, int W, int C, void* out, int64_t ldo,
                                   void* stream) {
  if (!g || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || ldg < C || ldo < C) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) {
    const int cch = C / 4;
    hipLaunchKernelGGL(lowpass5_adjoint_kernel<float>, dim3(fs_blocks((long long)N * H * W * cch)), dim3(256), 0, st, (const float*)g, (long long)ldg, N, H, W, cch, (float*)out, (long long)ldo);
  } else if (dtype == DG_BF16) {
    const int cch = C / 8;
    hipLaunchKernelGGL(lowpass5_adjoint_kernel<bf16_t>, dim3(fs_blocks((long long)N * H * W * cch)), dim3(256), 0, st, (const bf16_t*)g, (long long)ldg, N, H, W, cch, (bf16_t*)out, (long long)ldo);
  } else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
