// Linear family for the critic head (reference DoWnGAN/networks/critic.py:94-105).
//
// FC1 at cfg2 is [B=32] x [4 194 304 -> 100]: one streaming read of a 0.84 GB (bf16) weight per
// pass => HBM-bound.  Forward is a split-K MFMA GEMV-like kernel (both
// operands are K-contiguous, fragments come straight from global memory, no LDS: every weight byte
// is used by exactly one wave); dx and dw are VALU streaming kernels that read each weight /
// activation 16-B chunk once per small register tile.
#include "dg_internal.h"

template <typename T> struct MmaL;
template <> struct MmaL<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaL<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// y[b][o] += sum_k x[b][k] w[o][k].  A = w rows (o), B = x rows (b): D[row=o][col=b].
template <typename T, int NB, int NO>
__global__ __launch_bounds__(256) void lin_fwd_kernel(const T* __restrict__ x, long long ldx, const T* __restrict__ w,
                                                      long long ldw, float* y, int ldy, int B, int O, long long K,
                                                      long long kpw) {
  constexpr int EPC = DT<T>::EPC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const int l15 = lane & 15, g = lane >> 4;
  f32x4_t acc[NO][NB];
#pragma unroll
  for (int j = 0; j < NO; ++j)
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // K is dealt to the waves in INTERLEAVED blocks of UK * 32 (bf16) elements: at any moment the whole grid reads one
  // neighbourhood of each weight row (2048 waves x 256 B = 512 KB runs per row) instead of 2048 x 112 scattered streams
  // -- contiguous per-wave ranges kept the weight stream at 1.7 TB/s (DRAM page misses).  UK k-blocks' loads go out
  // before the first MFMA.
  constexpr int UK = NO * NB <= 14 ? 4 : 2;
  const long long blk = (long long)UK * 4 * EPC;
  for (long long kb = gw * blk; kb < K; kb += kpw * blk) {          // kpw = number of waves in the grid
    const long long k = kb + g * EPC;
    uint4 fb[UK][NB], fa[UK][NO];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const long long ku = k + u * 4 * EPC;
      const bool kin = ku < K;                        // K % (4 * EPC) == 0: a k-block is inside the range or past it
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int b = 16 * i + l15;
        const bool ok = b < B && kin;
        uint4 v = *reinterpret_cast<const uint4*>(x + (ok ? (long long)b * ldx + ku : 0ll));
        fb[u][i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
#pragma unroll
      for (int j = 0; j < NO; ++j) {
        uint4 v = *reinterpret_cast<const uint4*>(w + (long long)(16 * j + l15) * ldw + (kin ? ku : k));
        fa[u][j] = make_uint4(kin ? v.x : 0u, kin ? v.y : 0u, kin ? v.z : 0u, kin ? v.w : 0u);
      }
    }
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int i = 0; i < NB; ++i) MmaL<T>::run(fa[u][j], fb[u][i], acc[j][i]);
  }
#pragma unroll
  for (int j = 0; j < NO; ++j)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int b = 16 * i + l15;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = 16 * j + 4 * g + e;
        if (b < B && o < O) atomicAdd(y + (long long)b * ldy + o, acc[j][i][e]);
      }
    }
}

template <typename T, int NB, int NO>
static int lin_fwd_launch(const T* x, long long ldx, const T* w, long long ldw, float* y, int ldy, int B, int O,
                          long long K, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int UK = NO * NB <= 14 ? 4 : 2;
  const long long blk = (long long)UK * 4 * EPC;
  long long nw = (K + blk - 1) / blk;                  // one interleaved block stream per wave, at most 2048 waves
  if (nw > 2048) nw = 2048;
  nw = (nw + 3) / 4 * 4;
  const long long kpw = nw;
  const unsigned nb = (unsigned)(nw / 4);
  hipLaunchKernelGGL((lin_fwd_kernel<T, NB, NO>), dim3(nb), dim3(256), 0, st, x, ldx, w, ldw, y, ldy, B, O, K, kpw);
  return dg_check_launch();
}

template <typename T>
static int lin_fwd_dispatch(const void* xv, long long ldx, const void* wv, long long ldw, float* y, int ldy, int B,
                            int O, long long K, hipStream_t st) {
  const T* x = reinterpret_cast<const T*>(xv);
  const T* w = reinterpret_cast<const T*>(wv);
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int bb = B - b0 < 64 ? B - b0 : 64;
    const int nbf = (bb + 15) / 16;
    int o0 = 0;
    while (o0 < O) {
      int rc;
      const T* xb = x + (long long)b0 * ldx;
      const T* wo = w + (long long)o0 * ldw;
      float* yo = y + (long long)b0 * ldy + o0;
      if (O - o0 >= 112) {
        if (nbf == 1) rc = lin_fwd_launch<T, 1, 7>(xb, ldx, wo, ldw, yo, ldy, bb, 112, K, st);
        else if (nbf == 2) rc = lin_fwd_launch<T, 2, 7>(xb, ldx, wo, ldw, yo, ldy, bb, 112, K, st);
        else rc = lin_fwd_launch<T, 4, 7>(xb, ldx, wo, ldw, yo, ldy, bb, 112, K, st);
        o0 += 112;
      } else {
        if (nbf == 1) rc = lin_fwd_launch<T, 1, 1>(xb, ldx, wo, ldw, yo, ldy, bb, 16, K, st);
        else if (nbf == 2) rc = lin_fwd_launch<T, 2, 1>(xb, ldx, wo, ldw, yo, ldy, bb, 16, K, st);
        else rc = lin_fwd_launch<T, 4, 1>(xb, ldx, wo, ldw, yo, ldy, bb, 16, K, st);
        o0 += 16;
      }
      if (rc) return rc;
    }
  }
  return DG_OK;
}

extern "C" int dg_linear_fwd(int dtype, const void* x, int64_t ldx, const void* w, int64_t ldw, float* y, int ldy,
                             int B, int O, int64_t K, void* stream) {
  if (!x || !w || !y) return DG_ERR_BAD_ARG;
  if (B <= 0 || O <= 0 || O % 16 || K <= 0) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) {
    if (K % 16 || ldx % 4 || ldw % 4) return DG_ERR_BAD_SHAPE;
    return lin_fwd_dispatch<float>(x, ldx, w, ldw, y, ldy, B, O, K, st);
  } else if (dtype == DG_BF16) {
    if (K % 32 || ldx % 8 || ldw % 8) return DG_ERR_BAD_SHAPE;
    return lin_fwd_dispatch<bf16_t>(x, ldx, w, ldw, y, ldy, B, O, K, st);
  }
  return DG_ERR_BAD_DTYPE;
}

// dx[b][k] = (sum_o dy[b][o] w[o][k]) * leaky'(mask[b][k])
template <typename T, typename TO, int BG>
__global__ __launch_bounds__(256) void lin_dx_kernel(const float* __restrict__ dy, int ldo, const T* __restrict__ w,
                                                     long long ldw, TO* dx, long long lddx, const T* mask,
                                                     long long ldmask, float slope, int B, int O, long long K) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sdy[BG * 128];
  const int b0 = blockIdx.y * BG;
  for (int i = threadIdx.x; i < BG * O; i += 256) {
    const int b = i / O, o = i % O;
    sdy[b * 128 + o] = (b0 + b < B) ? dy[(long long)(b0 + b) * ldo + o] : 0.f;
  }
  __syncthreads();
  const long long k = ((long long)blockIdx.x * 256 + threadIdx.x) * EPC;
  if (k >= K) return;
  float acc[BG][EPC];
#pragma unroll
  for (int b = 0; b < BG; ++b)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[b][e] = 0.f;
  for (int o = 0; o < O; ++o) {
    float wv[EPC];
    ld4(w + (long long)o * ldw + k, wv);
    if constexpr (EPC == 8) ld4(w + (long long)o * ldw + k + 4, wv + 4);
#pragma unroll
    for (int b = 0; b < BG; ++b) {
      const float d = sdy[b * 128 + o];
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[b][e] = fmaf(d, wv[e], acc[b][e]);
    }
  }
#pragma unroll
  for (int b = 0; b < BG; ++b) {
    if (b0 + b >= B) continue;
    if (mask) {
      float mv[EPC];
      ld4(mask + (long long)(b0 + b) * ldmask + k, mv);
      if constexpr (EPC == 8) ld4(mask + (long long)(b0 + b) * ldmask + k + 4, mv + 4);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[b][e] *= leaky_grad(mv[e], slope);
    }
    st4(dx + (long long)(b0 + b) * lddx + k, acc[b]);
    if constexpr (EPC == 8) st4(dx + (long long)(b0 + b) * lddx + k + 4, acc[b] + 4);
  }
}

extern "C" int dg_linear_dx(int dtype, int out_dtype, const float* dy, int ldo, const void* w, int64_t ldw, void* dx,
                            int64_t lddx, const void* mask, int64_t ldmask, float mask_slope, int B, int O, int64_t K,
                            void* stream) {
  if (!dy || !w || !dx) return DG_ERR_BAD_ARG;
  if (B <= 0 || O <= 0 || O > 128 || K <= 0 || K % 8 || ldw % 8 || lddx % 8) return DG_ERR_BAD_SHAPE;
  if (mask && ldmask % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int BG = 16;             // batch rows per pass over W: a batch of 32 streams the matrix twice (was 4x with 8)
  const int epc = dtype == DG_F32 ? 4 : 8;
  dim3 grid((unsigned)((K / epc + 255) / 256), (unsigned)((B + BG - 1) / BG));
  if (dtype == DG_F32 && out_dtype == DG_F32)
    hipLaunchKernelGGL((lin_dx_kernel<float, float, BG>), grid, dim3(256), 0, st, dy, ldo, (const float*)w, ldw,
                       (float*)dx, lddx, (const float*)mask, ldmask, mask_slope, B, O, K);
  else if (dtype == DG_BF16 && out_dtype == DG_BF16)
    hipLaunchKernelGGL((lin_dx_kernel<bf16_t, bf16_t, BG>), grid, dim3(256), 0, st, dy, ldo, (const bf16_t*)w, ldw,
                       (bf16_t*)dx, lddx, (const bf16_t*)mask, ldmask, mask_slope, B, O, K);
  else if (dtype == DG_BF16 && out_dtype == DG_F32)
    hipLaunchKernelGGL((lin_dx_kernel<bf16_t, float, BG>), grid, dim3(256), 0, st, dy, ldo, (const bf16_t*)w, ldw,
                       (float*)dx, lddx, (const bf16_t*)mask, ldmask, mask_slope, B, O, K);
  else
    return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// dw[o][k] += sum_b dy[b][o] x[b][k]
template <typename T, int OT>
__global__ __launch_bounds__(256) void lin_dw_kernel(const float* __restrict__ dy, int ldo, const T* __restrict__ x,
                                                     long long ldx, float* dw, long long lddw, int B, int O,
                                                     long long K) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float sdy[64 * OT];
  const int o0 = blockIdx.y * OT;
  for (int i = threadIdx.x; i < B * OT; i += 256) {
    const int b = i / OT, o = i % OT;
    sdy[b * OT + o] = (o0 + o < O) ? dy[(long long)b * ldo + o0 + o] : 0.f;
  }
  __syncthreads();
  const long long k = ((long long)blockIdx.x * 256 + threadIdx.x) * EPC;
  if (k >= K) return;
  float acc[OT][EPC];
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[o][e] = 0.f;
  for (int b = 0; b < B; ++b) {
    float xv[EPC];
    ld4(x + (long long)b * ldx + k, xv);
    if constexpr (EPC == 8) ld4(x + (long long)b * ldx + k + 4, xv + 4);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const float d = sdy[b * OT + o];
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[o][e] = fmaf(d, xv[e], acc[o][e]);
    }
  }
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    if (o0 + o >= O) continue;
    float* p = dw + (long long)(o0 + o) * lddw + k;
#pragma unroll
    for (int e4 = 0; e4 < EPC; e4 += 4) {
      float4 t = *reinterpret_cast<float4*>(p + e4);
      t.x += acc[o][e4]; t.y += acc[o][e4 + 1]; t.z += acc[o][e4 + 2]; t.w += acc[o][e4 + 3];
      *reinterpret_cast<float4*>(p + e4) = t;
    }
  }
}

extern "C" int dg_linear_dw(int dtype, const float* dy, int ldo, const void* x, int64_t ldx, float* dw, int64_t lddw,
                            int B, int O, int64_t K, void* stream) {
  if (!dy || !x || !dw) return DG_ERR_BAD_ARG;
  if (B <= 0 || B > 64 || O <= 0 || K <= 0 || K % 8 || ldx % 8 || lddw % 4) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int OT = 16;
  const int epc = dtype == DG_F32 ? 4 : 8;
  dim3 grid((unsigned)((K / epc + 255) / 256), (unsigned)((O + OT - 1) / OT));
  if (dtype == DG_F32)
    hipLaunchKernelGGL((lin_dw_kernel<float, OT>), grid, dim3(256), 0, st, dy, ldo, (const float*)x, ldx, dw, lddw, B, O, K);
  else if (dtype == DG_BF16)
    hipLaunchKernelGGL((lin_dw_kernel<bf16_t, OT>), grid, dim3(256), 0, st, dy, ldo, (const bf16_t*)x, ldx, dw, lddw, B, O, K);
  else
    return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
