// HBM-bound elementwise / reduction kernels of the train step: 16 B per lane, grid-stride, fp32 math.
#include <cstdlib>
#include "dg_internal.h"
#include <type_traits>

#include <math.h>

static inline unsigned ew_blocks(long long nthreads) {
  long long b = (nthreads + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;  // 256 CUs x 16 blocks, grid-stride over the rest
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T> __device__ __forceinline__ void ldc(const T* p, float* v) {
  ld4(p, v);
  if constexpr (DT<T>::EPC == 8) ld4(p + 4, v + 4);
}
template <typename T> __device__ __forceinline__ void stc(T* p, const float* v) {
  st4(p, v);
  if constexpr (DT<T>::EPC == 8) st4(p + 4, v + 4);
}

// ------------------------------------------------------------------ mask multiply (in place)
template <typename T>
__global__ void mask_mul_kernel(T* u, long long ldu, const T* y, long long ldy, long long rows, int cchunks, float slope) {
  constexpr int EPC = DT<T>::EPC;
  const long long total = rows * cchunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cchunks; const int c = (int)(i % cchunks) * EPC;
    float a[EPC], b[EPC];
    ldc(u + r * ldu + c, a); ldc(y + r * ldy + c, b);
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[e] *= leaky_grad(b[e], slope);
    stc(u + r * ldu + c, a);
  }
}
extern "C" int dg_mask_mul(int dtype, void* u, int64_t ldu, const void* y, int64_t ldy, int64_t rows, int C, float slope,
                           void* stream) {
  if (!u || !y || rows <= 0 || C <= 0 || C % 8 || ldu % 8 || ldy % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) hipLaunchKernelGGL(mask_mul_kernel<float>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, st, (float*)u, ldu, (const float*)y, ldy, rows, C / 4, slope);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(mask_mul_kernel<bf16_t>, dim3(ew_blocks(rows * (C / 8))), dim3(256), 0, st, (bf16_t*)u, ldu, (const bf16_t*)y, ldy, rows, C / 8, slope);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// ------------------------------------------------------------------ out = a*x + b*y
template <typename T>
__global__ void axpby_kernel(T* out, long long ldo, const T* x, long long ldx, float a, const T* y, long long ldy, float b,
                             long long rows, int cchunks) {
  constexpr int EPC = DT<T>::EPC;
  const long long total = rows * cchunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cchunks; const int c = (int)(i % cchunks) * EPC;
    float xv[EPC], yv[EPC];
    ldc(x + r * ldx + c, xv);
    if (y) {
      ldc(y + r * ldy + c, yv);
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[e] = a * xv[e] + b * yv[e];
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[e] = a * xv[e];
    }
    stc(out + r * ldo + c, xv);
  }
}
extern "C" int dg_axpby(int dtype, void* out, int64_t ldo, const void* x, int64_t ldx, float a, const void* y, int64_t ldy,
                        float b, int64_t rows, int C, void* stream) {
  if (!out || !x || rows <= 0 || C <= 0 || C % 8 || ldo % 8 || ldx % 8 || (y && ldy % 8)) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) hipLaunchKernelGGL(axpby_kernel<float>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, st, (float*)out, ldo, (const float*)x, ldx, a, (const float*)y, ldy, b, rows, C / 4);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(axpby_kernel<bf16_t>, dim3(ew_blocks(rows * (C / 8))), dim3(256), 0, st, (bf16_t*)out, ldo, (const bf16_t*)x, ldx, a, (const bf16_t*)y, ldy, b, rows, C / 8);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// ------------------------------------------------------------------ GP interpolate / per-image scale
template <typename T>
__global__ void interp_kernel(const T* real, const T* fake, const float* alpha, T* xhat, long long chunks_per_img) {
  constexpr int EPC = DT<T>::EPC;
  const int b = blockIdx.y;
  const float al = alpha[b];
  const long long base = (long long)b * chunks_per_img * EPC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks_per_img; i += (long long)gridDim.x * blockDim.x) {
    float r[EPC], f[EPC];
    ldc(real + base + i * EPC, r); ldc(fake + base + i * EPC, f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) r[e] = al * r[e] + (1.f - al) * f[e];  // wasserstein.py:94
    stc(xhat + base + i * EPC, r);
  }
}
extern "C" int dg_gp_interp(int dtype, const void* real, const void* fake, const float* alpha, void* xhat, int B,
                            int64_t per_img, void* stream) {
  if (!real || !fake || !alpha || !xhat || B <= 0 || per_img <= 0 || per_img % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) { dim3 g(ew_blocks(per_img / 4) , B); hipLaunchKernelGGL(interp_kernel<float>, g, dim3(256), 0, st, (const float*)real, (const float*)fake, alpha, (float*)xhat, per_img / 4); }
  else if (dtype == DG_BF16) { dim3 g(ew_blocks(per_img / 8), B); hipLaunchKernelGGL(interp_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)real, (const bf16_t*)fake, alpha, (bf16_t*)xhat, per_img / 8); }
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// The 2-channel fields of the fine grid (wasserstein.py:94: real, fake, interpolate) are stored 16 channels wide like every
// activation; the critic's first layer reads the two real channels only, and its gathers are what bounds it (csrc/conv_small.hip,
// gg_im2col_direct_kernel: 0.79 -> 0.62 ms per 8 images from a compact [pixel][2] source).  These two kernels write the COMPACT
// forms where the bytes are produced anyway: the interpolate (plus compact copies of its two inputs, which it has in registers)
// and the penalty's scaled gradient.  One thread = 16 bytes of each output (4 bf16 / 2 fp32 pixels).
template <typename T>
__global__ void interp_c2_kernel(const T* real, const T* fake, long long ld, const float* alpha, T* xhat_c, T* real_c, T* fake_c,
                                 long long pix_per_img) {
  constexpr int PPT = 8 / (int)sizeof(T);            // pixels per thread
  typedef typename std::conditional<sizeof(T) == 2, unsigned, uint2>::type pair_t;
  const int b = blockIdx.y;
  const float al = alpha[b];
  const long long pbase = (long long)b * pix_per_img;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pix_per_img / PPT; i += (long long)gridDim.x * blockDim.x) {
    const long long p0 = pbase + i * PPT;
    pair_t r[PPT], f[PPT], x[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      r[k] = *reinterpret_cast<const pair_t*>(real + (p0 + k) * ld);
      f[k] = *reinterpret_cast<const pair_t*>(fake + (p0 + k) * ld);
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const T* rp = reinterpret_cast<const T*>(&r[k]);
      const T* fp = reinterpret_cast<const T*>(&f[k]);
      T* xp = reinterpret_cast<T*>(&x[k]);
      st_elem(xp, al * ld_elem(rp) + (1.f - al) * ld_elem(fp));              // wasserstein.py:94
      st_elem(xp + 1, al * ld_elem(rp + 1) + (1.f - al) * ld_elem(fp + 1));
    }
    *reinterpret_cast<uint4*>(xhat_c + p0 * 2) = *reinterpret_cast<const uint4*>(x);
    if (real_c) *reinterpret_cast<uint4*>(real_c + p0 * 2) = *reinterpret_cast<const uint4*>(r);
    if (fake_c) *reinterpret_cast<uint4*>(fake_c + p0 * 2) = *reinterpret_cast<const uint4*>(f);
  }
}
extern "C" int dg_gp_interp_c2(int dtype, const void* real, const void* fake, int64_t ld, const float* alpha, void* xhat_c,
                               void* real_c, void* fake_c, int B, int64_t pix_per_img, void* stream) {
  if (!real || !fake || !alpha || !xhat_c) return DG_ERR_BAD_ARG;
  if (B <= 0 || pix_per_img <= 0 || pix_per_img % 4 || ld < 2 || ld % 2) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) { dim3 g(ew_blocks(pix_per_img / 2), B); hipLaunchKernelGGL(interp_c2_kernel<float>, g, dim3(256), 0, st, (const float*)real, (const float*)fake, (long long)ld, alpha, (float*)xhat_c, (float*)real_c, (float*)fake_c, (long long)pix_per_img); }
  else if (dtype == DG_BF16) { dim3 g(ew_blocks(pix_per_img / 4), B); hipLaunchKernelGGL(interp_c2_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)real, (const bf16_t*)fake, (long long)ld, alpha, (bf16_t*)xhat_c, (bf16_t*)real_c, (bf16_t*)fake_c, (long long)pix_per_img); }
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

template <typename T>
__global__ void scale_rows_c2_kernel(const T* g, long long ld, const float* coef, T* out_c, long long pix_per_img) {
  constexpr int PPT = 8 / (int)sizeof(T);
  typedef typename std::conditional<sizeof(T) == 2, unsigned, uint2>::type pair_t;
  const int b = blockIdx.y;
  const float cf = coef[b];
  const long long pbase = (long long)b * pix_per_img;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pix_per_img / PPT; i += (long long)gridDim.x * blockDim.x) {
    const long long p0 = pbase + i * PPT;
    pair_t v[PPT], x[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) v[k] = *reinterpret_cast<const pair_t*>(g + (p0 + k) * ld);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const T* vp = reinterpret_cast<const T*>(&v[k]);
      T* xp = reinterpret_cast<T*>(&x[k]);
      st_elem(xp, ld_elem(vp) * cf);
      st_elem(xp + 1, ld_elem(vp + 1) * cf);
    }
    *reinterpret_cast<uint4*>(out_c + p0 * 2) = *reinterpret_cast<const uint4*>(x);
  }
}
extern "C" int dg_scale_rows_c2(int dtype, const void* g, int64_t ld, const float* coef, void* out_c, int B, int64_t pix_per_img,
                                void* stream) {
  if (!g || !coef || !out_c) return DG_ERR_BAD_ARG;
  if (B <= 0 || pix_per_img <= 0 || pix_per_img % 4 || ld < 2 || ld % 2) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) { dim3 gr(ew_blocks(pix_per_img / 2), B); hipLaunchKernelGGL(scale_rows_c2_kernel<float>, gr, dim3(256), 0, st, (const float*)g, (long long)ld, coef, (float*)out_c, (long long)pix_per_img); }
  else if (dtype == DG_BF16) { dim3 gr(ew_blocks(pix_per_img / 4), B); hipLaunchKernelGGL(scale_rows_c2_kernel<bf16_t>, gr, dim3(256), 0, st, (const bf16_t*)g, (long long)ld, coef, (bf16_t*)out_c, (long long)pix_per_img); }
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

template <typename T>
__global__ void scale_rows_kernel(const T* g, const float* coef, T* out, long long chunks_per_img) {
  constexpr int EPC = DT<T>::EPC;
  const int b = blockIdx.y;
  const float cf = coef[b];
  const long long base = (long long)b * chunks_per_img * EPC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks_per_img; i += (long long)gridDim.x * blockDim.x) {
    float r[EPC];
    ldc(g + base + i * EPC, r);
#pragma unroll
    for (int e = 0; e < EPC; ++e) r[e] *= cf;
    stc(out + base + i * EPC, r);
  }
}
extern "C" int dg_scale_rows(int dtype, const void* g, const float* coef, void* out, int B, int64_t per_img, void* stream) {
  if (!g || !coef || !out || B <= 0 || per_img <= 0 || per_img % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) { dim3 gr(ew_blocks(per_img / 4), B); hipLaunchKernelGGL(scale_rows_kernel<float>, gr, dim3(256), 0, st, (const float*)g, coef, (float*)out, per_img / 4); }
  else if (dtype == DG_BF16) { dim3 gr(ew_blocks(per_img / 8), B); hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, gr, dim3(256), 0, st, (const bf16_t*)g, coef, (bf16_t*)out, per_img / 8); }
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// ------------------------------------------------------------------ per-image sum of squares (wave-shuffle reduce)
// (det_ws / det_stride, here and below: deterministic mode -- block c accumulates into copy c of the target inside the workspace,
// dg_internal.h DetPlan)
template <typename T>
__global__ void sumsq_kernel(const T* g, long long chunks_per_img, float* ss, float* det_ws, long long det_stride) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float part[4];
  const int b = blockIdx.y;
  const long long base = (long long)b * chunks_per_img * EPC;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks_per_img; i += (long long)gridDim.x * blockDim.x) {
    float r[EPC];
    ldc(g + base + i * EPC, r);
#pragma unroll
    for (int e = 0; e < EPC; ++e) s = fmaf(r[e], r[e], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd((det_ws ? det_ws + blockIdx.x * det_stride : ss) + b, part[0] + part[1] + part[2] + part[3]);
}
extern "C" int dg_sumsq_rows(int dtype, const void* g, int B, int64_t per_img, float* ss, void* stream) {
  if (!g || !ss || B <= 0 || per_img <= 0 || per_img % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int epc = dtype == DG_F32 ? 4 : 8;
  long long nb = (per_img / epc + 256 * 8 - 1) / (256 * 8);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  if (dtype != DG_F32 && dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  DetPlan plan;
  if (dg_det_begin(B, (int)nb, st, &plan) != DG_OK) return DG_ERR_LAUNCH;
  nb = plan.copies;
  dim3 gr((unsigned)nb, B);
  if (dtype == DG_F32) hipLaunchKernelGGL(sumsq_kernel<float>, gr, dim3(256), 0, st, (const float*)g, per_img / 4, ss, plan.ws, plan.stride);
  else hipLaunchKernelGGL(sumsq_kernel<bf16_t>, gr, dim3(256), 0, st, (const bf16_t*)g, per_img / 8, ss, plan.ws, plan.stride);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return dg_det_reduce(plan, 0, ss, B, st);
}

// n_b = sqrt(ss_b + 1e-12); gp_ret = lambda * mean_b (n_b-1)^2 ; coef_b = weight*lambda*(2/Bglobal)*(n_b-1)/n_b
__global__ void gp_finish_kernel(const float* ss, int B, int Bglobal, float lambda, float weight, float* coef, float* scalars) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const float n = sqrtf(ss[b] + 1e-12f);
    const float d = n - 1.f;
    s += d * d;
    coef[b] = weight * lambda * (2.f / (float)Bglobal) * d / n;
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) scalars[0] = lambda * s / (float)Bglobal;
}
extern "C" int dg_gp_finish(const float* ss, int B, int B_global, float gp_lambda, float weight, float* coef, float* scalars,
                            void* stream) {
  if (!ss || !coef || !scalars || B <= 0 || B_global < B) return DG_ERR_BAD_ARG;
  hipLaunchKernelGGL(gp_finish_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ss, B, B_global, gp_lambda, weight, coef, scalars);
  return dg_check_launch();
}

// ------------------------------------------------------------------ L1 content loss + gradient
template <typename T, bool SQ>   // SQ: accumulate (a-b)^2 instead of |a-b| (MSE metric); no gradient in that mode
__global__ void l1_kernel(const T* a, long long lda, const T* b, long long ldb, long long rows, int cchunks, float* acc,
                          T* grad, long long ldg, float gscale, const T* addend, long long ldadd, float* det_ws) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float part[4];
  const long long total = rows * cchunks;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cchunks; const int c = (int)(i % cchunks) * EPC;
    float av[EPC], bv[EPC];
    ldc(a + r * lda + c, av); ldc(b + r * ldb + c, bv);
    float gv[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float d = av[e] - bv[e];
      s += SQ ? d * d : fabsf(d);
      gv[e] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    }
    if (grad) {
      if (addend) {
        float ad[EPC];
        ldc(addend + r * ldadd + c, ad);
#pragma unroll
        for (int e = 0; e < EPC; ++e) gv[e] += ad[e];
      }
      stc(grad + r * ldg + c, gv);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(det_ws ? det_ws + blockIdx.x : acc, part[0] + part[1] + part[2] + part[3]);
}
extern "C" int dg_l1(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int C, float* acc,
                     void* grad, int64_t ldg, float grad_scale, const void* addend, int64_t ldadd, void* stream) {
  if (!a || !b || !acc || rows <= 0 || C <= 0 || C % 8 || lda % 8 || ldb % 8) return DG_ERR_BAD_SHAPE;
  if (grad && ldg % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int epc = dtype == DG_F32 ? 4 : 8;
  long long nb = (rows * (C / epc) + 256 * 8 - 1) / (256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  if (dtype != DG_F32 && dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  DetPlan plan;
  if (dg_det_begin(1, (int)nb, st, &plan) != DG_OK) return DG_ERR_LAUNCH;
  nb = plan.copies;
  if (dtype == DG_F32) hipLaunchKernelGGL((l1_kernel<float, false>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)a, lda, (const float*)b, ldb, rows, C / 4, acc, (float*)grad, ldg, grad_scale, (const float*)addend, ldadd, plan.ws);
  else hipLaunchKernelGGL((l1_kernel<bf16_t, false>), dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, rows, C / 8, acc, (bf16_t*)grad, ldg, grad_scale, (const bf16_t*)addend, ldadd, plan.ws);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return dg_det_reduce(plan, 0, acc, 1, st);
}

extern "C" int dg_sqdiff(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int C, float* acc,
                         void* stream) {
  if (!a || !b || !acc || rows <= 0 || C <= 0 || C % 8 || lda % 8 || ldb % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int epc = dtype == DG_F32 ? 4 : 8;
  long long nb = (rows * (C / epc) + 256 * 8 - 1) / (256 * 8);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  if (dtype != DG_F32 && dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  DetPlan plan;
  if (dg_det_begin(1, (int)nb, st, &plan) != DG_OK) return DG_ERR_LAUNCH;
  nb = plan.copies;
  if (dtype == DG_F32) hipLaunchKernelGGL((l1_kernel<float, true>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)a, lda, (const float*)b, ldb, rows, C / 4, acc, (float*)nullptr, 0ll, 0.f, (const float*)nullptr, 0ll, plan.ws);
  else hipLaunchKernelGGL((l1_kernel<bf16_t, true>), dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, rows, C / 8, acc, (bf16_t*)nullptr, 0ll, 0.f, (const bf16_t*)nullptr, 0ll, plan.ws);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return dg_det_reduce(plan, 0, acc, 1, st);
}

// ------------------------------------------------------------------ column sum (bias gradients)
struct ColsumDst { float* p[8]; int seg; };     // column c accumulates into p[c / seg][c % seg] (one destination: seg = C)
template <typename T, int UNR>
__global__ __launch_bounds__(256) void colsum_kernel(const T* dy, long long ld_outer, int rows_inner, long long ld, int rows,
                                                      int cchunks, int rows_per_block, const ColsumDst dst, float* det_ws, long long det_stride) {
  // HBM-bound pass: every thread keeps UNR independent 16-byte loads in flight (a one-load-per-iteration loop with a
  // 64-bit division in it ran at 0.6 TB/s)
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[256 * 8];
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  const int nrl = 256 / cchunks;                       // row lanes per workgroup
  const int tx = threadIdx.x % cchunks, ty = threadIdx.x / cchunks;
  float s[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) s[e] = 0.f;
  if (ty < nrl) {
    const T* base = dy + tx * EPC;
    for (int r = r0 + ty; r < r1; r += nrl * UNR) {
      float v[UNR][EPC];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int rr = r + u * nrl;
        const int rc = rr < r1 ? rr : r0;                // clamped to a valid row; masked out below
        const int ro = (int)((unsigned)rc / (unsigned)rows_inner), ri = rc - ro * rows_inner;
        ldc(base + ro * ld_outer + ri * ld, v[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float m = (r + u * nrl) < r1 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[e] += m * v[u][e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) red[threadIdx.x * EPC + e] = s[e];
  __syncthreads();
  if (ty == 0) {
    for (int t = 1; t < nrl; ++t)
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] += red[(t * cchunks + tx) * EPC + e];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = tx * EPC + e;
      atomicAdd(det_ws ? det_ws + blockIdx.x * det_stride + c : dst.p[c / dst.seg] + c % dst.seg, s[e]);
    }
  }
}
static int colsum_launch(int dtype, const void* dy, int64_t rows_outer, int64_t ld_outer, int64_t rows_inner, int64_t ld, int C,
                         const ColsumDst& dst, long long max_blocks, hipStream_t st) {
  const long long rows = (long long)rows_outer * rows_inner;
  long long nb = rows / 256;
  if (nb < 1) nb = 1;
  if (nb > max_blocks) nb = max_blocks;
  DetPlan plan;
  if (dg_det_begin(C, (int)nb, st, &plan) != DG_OK) return DG_ERR_LAUNCH;
  nb = plan.copies;
  const long long rpb = (rows + nb - 1) / nb;
  nb = (rows + rpb - 1) / rpb;
  if (dtype == DG_F32) hipLaunchKernelGGL((colsum_kernel<float, 8>), dim3((unsigned)nb), dim3(256), 0, st, (const float*)dy, ld_outer, (int)rows_inner, ld, (int)rows, C / 4, (int)rpb, dst, plan.ws, plan.stride);
  else hipLaunchKernelGGL((colsum_kernel<bf16_t, 8>), dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)dy, ld_outer, (int)rows_inner, ld, (int)rows, C / 8, (int)rpb, dst, plan.ws, plan.stride);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  for (int k = 0; k * dst.seg < C; ++k) {
    const int rc = dg_det_reduce(plan, (long long)k * dst.seg, dst.p[k], dst.seg, st);
    if (rc != DG_OK) return rc;
  }
  return DG_OK;
}
extern "C" int dg_colsum(int dtype, const void* dy, int64_t rows_outer, int64_t ld_outer, int64_t rows_inner, int64_t ld,
                         int C, float* db, void* stream) {
  if (!dy || !db || rows_outer <= 0 || rows_inner <= 0 || C <= 0 || C % 8 || ld % 8 || ld_outer % 8) return DG_ERR_BAD_SHAPE;
  const long long rows = (long long)rows_outer * rows_inner;
  const int epc = dtype == DG_F32 ? 4 : 8;
  if (C / epc > 256 || rows >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  if (dtype != DG_F32 && dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  // few, long workgroups: the pass is bound by the final same-address atomics, not by HBM (134 MB bf16, measured:
  // 4096 blocks 404 us, 1024 214 us (old kernel), 512 116 us, 256 72 us, 128 62 us)
  ColsumDst dst{};
  dst.p[0] = db; dst.seg = C;
  return colsum_launch(dtype, dy, rows_outer, ld_outer, rows_inner, ld, C, dst, 128, reinterpret_cast<hipStream_t>(stream));
}

// Column sums of ONE pass over a wide tensor into nseg destinations: db[k][c] += sum over rows of dy[row, k * (C / nseg) + c] -- the five
// bias gradients of a dense block from its adjoint slab [N*H*W, 5 * 128] (generator.py:24-41), rows of `ld` elements.
extern "C" int dg_colsum_multi(int dtype, const void* dy, int64_t rows, int64_t ld, int C, int nseg, float* const* db, void* stream) {
  if (!dy || !db || rows <= 0 || C <= 0 || nseg < 1 || nseg > 8 || C % nseg || (C / nseg) % 8 || ld % 8 || rows >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  if (dtype != DG_F32 && dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  if (C / (dtype == DG_F32 ? 4 : 8) > 256) return DG_ERR_BAD_SHAPE;
  ColsumDst dst{};
  for (int k = 0; k < nseg; ++k) { if (!db[k]) return DG_ERR_BAD_ARG; dst.p[k] = db[k]; }
  dst.seg = C / nseg;
  // the atomics spread over nseg times the addresses of a 128-column pass: as many more workgroups
  return colsum_launch(dtype, dy, rows, ld, 1, ld, C, dst, 128ll * nseg, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------ small head helpers
template <typename TO>
__global__ void bias_act_kernel(const float* in, int ldi, const float* bias, TO* out, int ldo, int rows, int C, int has_act,
                                float slope, const TO* mask, int ldmask, float mslope) {
  const int total = rows * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / C, c = i % C;
    float v = in[(long long)r * ldi + c];
    if (bias) v += bias[c];
    if (has_act) v = leaky(v, slope);
    if (mask) v *= leaky_grad(ld_elem(mask + (long long)r * ldmask + c), mslope);
    st_elem(out + (long long)r * ldo + c, v);
  }
}
extern "C" int dg_bias_act(int out_dtype, const float* in, int ldi, const float* bias, void* out, int ldo, int rows, int C,
                           int has_act, float slope, const void* mask, int ldmask, float mask_slope, void* stream) {
  if (!in || !out || rows <= 0 || C <= 0) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = ew_blocks((long long)rows * C);
  if (out_dtype == DG_F32) hipLaunchKernelGGL(bias_act_kernel<float>, dim3(nb), dim3(256), 0, st, in, ldi, bias, (float*)out, ldo, rows, C, has_act, slope, (const float*)mask, ldmask, mask_slope);
  else if (out_dtype == DG_BF16) hipLaunchKernelGGL(bias_act_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, in, ldi, bias, (bf16_t*)out, ldo, rows, C, has_act, slope, (const bf16_t*)mask, ldmask, mask_slope);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

__global__ void sum_strided_kernel(const float* in, int n, int stride, float scale, float* out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += in[(long long)i * stride];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}
extern "C" int dg_sum_strided(const float* in, int n, int stride, float scale, float* out, void* stream) {
  if (!in || !out || n <= 0) return DG_ERR_BAD_ARG;
  hipLaunchKernelGGL(sum_strided_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), in, n, stride, scale, out);
  return dg_check_launch();
}

__global__ void fill_col_kernel(float* buf, int rows, int ld, int col, float value) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) buf[(long long)r * ld + col] = value;
}
extern "C" int dg_fill_col(float* buf, int rows, int ld, int col, float value, void* stream) {
  if (!buf || rows <= 0) return DG_ERR_BAD_ARG;
  hipLaunchKernelGGL(fill_col_kernel, dim3(ew_blocks(rows)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), buf, rows, ld, col, value);
  return dg_check_launch();
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam semantics, stage.py:63-64)
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, bf16_t* shadow, long long n4, float step_size,
                            float beta1, float beta2, float eps, float inv_sqrt_bc2, float gscale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
    float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = ga[e] * gscale;
      ma[e] = beta1 * ma[e] + (1.f - beta1) * gg;
      va[e] = beta2 * va[e] + (1.f - beta2) * gg * gg;
      const float denom = sqrtf(va[e]) * inv_sqrt_bc2 + eps;
      pa[e] -= step_size * (ma[e] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
    if (shadow) st4(shadow + 4 * i, pa);
  }
}
extern "C" int dg_adam(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr, float beta1,
                       float beta2, float eps, int step, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || n % 4 || step < 1) return DG_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                     (bf16_t*)shadow_bf16, (long long)(n / 4), (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  return dg_check_launch();
}

// ------------------------------------------------------------------ layout / precision converters
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* dst, int N, int C, int H, int W, int Cpad) {
  const long long npix = (long long)N * H * W, hw = (long long)H * W;
  const long long total = npix * Cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i / npix); const long long pix = i % npix;     // consecutive threads -> consecutive pixels
    const long long n = pix / hw, r = pix % hw;
    const float v = c < C ? src[(n * C + c) * hw + r] : 0.f;
    st_elem(dst + pix * Cpad + c, v);
  }
}
extern "C" int dg_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cpad, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = ew_blocks((long long)N * H * W * Cpad);
  if (dtype == DG_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(nb), dim3(256), 0, st, src, (float*)dst, N, C, H, W, Cpad);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, src, (bf16_t*)dst, N, C, H, W, Cpad);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, long long lds, float* dst, int N, int C, int H, int W) {
  const long long hw = (long long)H * W, total = (long long)N * C * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i % hw, t = i / hw;
    const int c = (int)(t % C); const long long n = t / C;
    dst[i] = ld_elem(src + (n * hw + r) * lds + c);
  }
}
extern "C" int dg_nhwc_to_nchw(int dtype, const void* src, int64_t lds, float* dst, int N, int C, int H, int W, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || lds < C) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = ew_blocks((long long)N * H * W * C);
  if (dtype == DG_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)src, lds, dst, N, C, H, W);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)src, lds, dst, N, C, H, W);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

template <typename T>
__global__ void cast_kernel(const float* src, T* dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) st_elem(dst + i, src[i]);
}
extern "C" int dg_cast(int dtype, const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return DG_ERR_BAD_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, src, (float*)dst, (long long)n);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, src, (bf16_t*)dst, (long long)n);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// master [CoutP][9][CinP] fp32 -> kind 0: same layout in dtype; kind 1: [CinP][9][CoutP] (data-gradient pack); kind 2: the
// same with the taps MIRRORED (tap t of dst = tap 8 - t of master): the weights with which the data gradient of a stride-1 conv
// is the plain FORWARD conv of dy (dx(p) = sum_t' dy(p + t') W[.][8 - t'][.]) -- used to run the data gradient of a layer with
// <= 2 real output channels (generator conv3.2) on the im2col forward kernel
template <typename T>
__global__ void repack_kernel(const float* master, T* dst, int CoutP, int CinP, int kind) {
  const long long total = (long long)CoutP * 9 * CinP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (kind == 0) { st_elem(dst + i, master[i]); continue; }
    const int co = (int)(i % CoutP); const long long t = i / CoutP;
    const int tap = (int)(t % 9), ci = (int)(t / 9);
    st_elem(dst + i, master[((long long)co * 9 + (kind == 2 ? 8 - tap : tap)) * CinP + ci]);
  }
}
// dw[co][t][ci] += tmp[ci][8 - t][co]: folds the result of a weight-gradient launch with swapped operand roles (see
// dg_repack_conv_weights kind 2 and engine.NativeGenerator.backward) back into the layer's own gradient layout
__global__ void wgrad_unswap_kernel(const float* tmp, float* dw, int CoutP, int CinP) {
  const int total = CoutP * 9 * CinP;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ci = i % CinP, t = (i / CinP) % 9, co = i / (CinP * 9);
    dw[i] += tmp[((long long)ci * 9 + (8 - t)) * CoutP + co];
  }
}
extern "C" int dg_wgrad_unswap(const float* tmp, float* dw, int CoutP, int CinP, void* stream) {
  if (!tmp || !dw || CoutP <= 0 || CinP <= 0) return DG_ERR_BAD_ARG;
  const int total = CoutP * 9 * CinP;
  hipLaunchKernelGGL(wgrad_unswap_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tmp, dw, CoutP, CinP);
  return dg_check_launch();
}
extern "C" int dg_repack_conv_weights(int dtype, int kind, const float* master, void* dst, int CoutP, int CinP, void* stream) {
  if (!master || !dst || CoutP <= 0 || CinP <= 0 || kind < 0 || kind > 2) return DG_ERR_BAD_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = ew_blocks((long long)CoutP * 9 * CinP);
  if (dtype == DG_F32) hipLaunchKernelGGL(repack_kernel<float>, dim3(nb), dim3(256), 0, st, master, (float*)dst, CoutP, CinP, kind);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(repack_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, master, (bf16_t*)dst, CoutP, CinP, kind);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// Data-gradient packs of a dense block's "virtual" convs (engine.NativeGenerator: the adjoint of slab slice j is one conv over the
// stacked adjoints of convs j+1..n).  masters[k-1] = conv k's fp32 weight [F][9][k*F]; dst = for j = 0..n-1 the kind-1 pack of
// the virtual conv V_j (F inputs, (n-j)*F outputs): dst_j[ci][tap][(k-j-1)*F + co] = W_k[co][tap][j*F + ci], k = j+1..n,
// the packs concatenated.  One launch per block instead of a gather + repack per slice.
struct DensePackArgs { const float* w[8]; };
template <typename T>
__global__ void dense_dgrad_pack_kernel(const DensePackArgs a, T* dst, int F, int n) {
  const long long per = 9ll * F * F;                   // elements per (slice, conv) pair
  const long long total = per * n * (n + 1) / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int j = 0;
    long long base = 0;
    while (i >= base + per * (n - j)) { base += per * (n - j); ++j; }
    const long long r = i - base;                      // index inside dst_j = [F][9][(n - j) * F]
    const int cov = (int)(r % ((long long)(n - j) * F));
    const long long t = r / ((long long)(n - j) * F);
    const int tap = (int)(t % 9), ci = (int)(t / 9);
    const int k = j + 1 + cov / F, co = cov % F;       // conv k (1-based), its output channel co
    st_elem(dst + i, a.w[k - 1][((long long)co * 9 + tap) * ((long long)k * F) + (long long)j * F + ci]);
  }
}
extern "C" int dg_repack_dense_dgrad(int dtype, const float* const* masters, int nconv, int F, void* dst, void* stream) {
  if (!masters || !dst || nconv < 1 || nconv > 8 || F <= 0 || F % 8) return DG_ERR_BAD_ARG;
  DensePackArgs a{};
  for (int k = 0; k < nconv; ++k) { if (!masters[k]) return DG_ERR_BAD_ARG; a.w[k] = masters[k]; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = ew_blocks(9ll * F * F * nconv * (nconv + 1) / 2);
  if (dtype == DG_F32) hipLaunchKernelGGL(dense_dgrad_pack_kernel<float>, dim3(nb), dim3(256), 0, st, a, (float*)dst, F, nconv);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(dense_dgrad_pack_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, a, (bf16_t*)dst, F, nconv);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

extern "C" const char* dg_version(void) { return "downgan_hip 0.1.0 (gfx950)"; }

// ------------------------------------------------------------------ resident-dataset minibatch gather (data feed)
// src: [n][HW][c_real] (compute dtype, real channels only, whole dataset resident in HBM); dst: [B][HW][c_pad] native
// activations (padding channels zero); one 16-byte destination chunk per thread.
template <typename T>
__global__ void gather_samples_kernel(const T* src, long long HW, int c_real, const long long* idx, int B, T* dst, int c_pad) {
  constexpr int EPC = DT<T>::EPC;
  const int cch = c_pad / EPC;
  const long long per = HW * cch, total = per * B;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const long long r = i - (long long)b * per;
    const long long p = r / cch;
    const int c0 = (int)(r - p * cch) * EPC;
    const T* s = src + (idx[b] * HW + p) * c_real;
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = (c0 + e < c_real) ? ld_elem(s + c0 + e) : 0.f;
    T* d = dst + ((long long)b * HW + p) * c_pad + c0;
    if constexpr (EPC == 8) { st4(d, v); st4(d + 4, v + 4); } else { st4(d, v); }
  }
}
extern "C" int dg_gather_samples(int dtype, const void* src, int64_t HW, int c_real, const int64_t* idx, int B, void* dst,
                                 int c_pad, void* stream) {
  if (!src || !idx || !dst || HW <= 0 || c_real <= 0 || B <= 0 || c_pad < c_real || c_pad % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int epc = dtype == DG_F32 ? 4 : 8;
  const unsigned nb = ew_blocks((long long)HW * (c_pad / epc) * B);
  if (dtype == DG_F32) hipLaunchKernelGGL(gather_samples_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)src, (long long)HW, c_real, (const long long*)idx, B, (float*)dst, c_pad);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(gather_samples_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)src, (long long)HW, c_real, (const long long*)idx, B, (bf16_t*)dst, c_pad);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
