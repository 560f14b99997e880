// Linear family for the critic head (reference DoWnGAN/networks/critic.py:94-105).
//
// FC1 at cfg2 is [B=32] x [4 194 304 -> 100]: one streaming read of a 0.84 GB (bf16) weight per
// pass => HBM-bound.  Forward is a split-K MFMA GEMV-like kernel (both
// operands are K-contiguous, fragments come straight from global memory, no LDS: every weight byte
// is used by exactly one wave); dx and dw are VALU streaming kernels that read each weight /
// activation 16-B chunk once per small register tile.
#include "dg_internal.h"
#include <stdlib.h>

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

// ---- helpers of the packed-fp32 streaming kernels (lin_dx_wide_kernel, lin_dw_wide_kernel)
typedef __attribute__((ext_vector_type(2))) float dg_f32x2_t;
template <typename T> struct LdX4;
// one row's 4 elements per lane through a buffer descriptor based at the (workgroup-uniform) row: scalar base + 32-bit lane
// offset.  Written as pointer arithmetic the compiler hoists `x + k` into a 64-bit VGPR pair per lane and serialises the loads.
typedef __attribute__((ext_vector_type(2))) unsigned int dg_u32x2_t;
template <> struct LdX4<float> {
  typedef dg_u32x4_t raw;
  static __device__ __forceinline__ raw load(const float* row, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)row, 0, -1, 0x00020000), voff, 0, 0);
  }
  static __device__ __forceinline__ void cvt(const raw& t, dg_f32x2_t& a, dg_f32x2_t& b) {
    a = dg_f32x2_t{__uint_as_float(t[0]), __uint_as_float(t[1])}; b = dg_f32x2_t{__uint_as_float(t[2]), __uint_as_float(t[3])};
  }
};
template <> struct LdX4<bf16_t> {
  typedef dg_u32x2_t raw;
  static __device__ __forceinline__ raw load(const bf16_t* row, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b64(__builtin_amdgcn_make_buffer_rsrc((void*)row, 0, -1, 0x00020000), voff, 0, 0);
  }
  static __device__ __forceinline__ void cvt(const raw& t, dg_f32x2_t& a, dg_f32x2_t& b) {
    a = dg_f32x2_t{__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u)};
    b = dg_f32x2_t{__uint_as_float(t[1] << 16), __uint_as_float(t[1] & 0xffff0000u)};
  }
};

// y[b][o] += sum_k x[b][k] w[o][k].  A = w rows (o), B = x rows (b): D[row=o][col=b].
template <typename T, int NB, int NO>
__global__ __launch_bounds__(256) void lin_fwd_kernel(const T* __restrict__ x, long long ldx, const T* __restrict__ w,
                                                      long long ldw, float* y, int ldy, int B, int O, long long K,
                                                      long long kpw, float* det_ws, long long det_stride) {
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
  // deterministic mode (dg_internal.h DetPlan): wave gw accumulates into copy gw of y inside the workspace
  float* const yp = det_ws ? det_ws + gw * det_stride : y;
#pragma unroll
  for (int j = 0; j < NO; ++j)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int b = 16 * i + l15;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = 16 * j + 4 * g + e;
        if (b < B && o < O) atomicAdd(yp + (long long)b * ldy + o, acc[j][i][e]);
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
  // deterministic mode: one copy of the y rows [B][ldy] of this call per wave (the copies hold this call's rows only: y - based offsets)
  DetPlan plan;
  const long long region = (long long)(B - 1) * ldy + O;
  if (dg_det_begin((region + 3) / 4 * 4, (int)nw, st, &plan) != DG_OK) return DG_ERR_LAUNCH;
  if (dg_det_on()) {                       // (a block is four waves: the workspace must hold at least four copies of the rows)
    if (!plan.ws || plan.copies < 4) return DG_ERR_BAD_ARG;
    nw = plan.copies / 4 * 4;
  }
  const long long kpw = nw;
  const unsigned nb = (unsigned)(nw / 4);
  hipLaunchKernelGGL((lin_fwd_kernel<T, NB, NO>), dim3(nb), dim3(256), 0, st, x, ldx, w, ldw, y, ldy, B, O, K, kpw, plan.ws, plan.stride);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  plan.copies = (int)nw;
  return dg_det_reduce(plan, 0, y, region, st);
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

// The same product with 32 batch rows per pass over W (the 16-row kernel above streams the 0.94 GB FC1 matrix twice for a batch
// of 32) and half the VALU slots: a lane owns 4 consecutive k of all 32 rows = 64 float2 accumulators fed by v_pk_fma_f32; the
// adjoint values sit TRANSPOSED in LDS ([o][32 rows]: eight broadcast ds_read_b128 per weight row against 64 packed FMAs, read
// one weight row ahead), the weight rows arrive through a row-based buffer descriptor six rows ahead of their arithmetic.
template <typename T, typename TO>
__global__ __launch_bounds__(256, 2) void lin_dx_wide_kernel(const float* __restrict__ dy, int ldo, const T* __restrict__ w,
                                                             long long ldw, TO* dx, long long lddx, const T* mask,
                                                             long long ldmask, float slope, int B, int O, long long K) {
  constexpr int BG = 32;
  __shared__ __attribute__((aligned(16))) float sdyT[129 * BG];          // [o][b] (+ one zero row read past the end)
  const int b0 = blockIdx.y * BG;
  for (int i = threadIdx.x; i < (O + 1) * BG; i += 256) {
    const int o = i / BG, b = i % BG;
    sdyT[i] = (o < O && b0 + b < B) ? dy[(long long)(b0 + b) * ldo + o] : 0.f;
  }
  __syncthreads();
  const long long k = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (k >= K) return;
  const unsigned voff = (unsigned)k * (unsigned)sizeof(T);
  dg_f32x2_t acc[BG][2];
#pragma unroll
  for (int b = 0; b < BG; ++b) acc[b][0] = acc[b][1] = dg_f32x2_t{0.f, 0.f};
  // The 32 adjoint values of a weight row: eight broadcast ds_read_b128, issued one row AHEAD as inline asm (written as plain
  // loads the compiler splits them into ds_read_b32 and sinks each in front of its FMA pair with a wait).
  typedef __attribute__((ext_vector_type(4))) float f4_t;
  f4_t d[2][BG / 4];
  const unsigned lds0 = (unsigned)(unsigned long long)((__attribute__((address_space(3))) float*)sdyT);
  auto read_row = [&](f4_t (&dst)[BG / 4], int o) {
    const unsigned adr = lds0 + (unsigned)o * (BG * 4);
#pragma unroll
    for (int q = 0; q < BG / 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst[q]) : "v"(adr), "n"(16 * q));
  };
  // (the hardware does not interlock a register that an LDS read is still filling: between `read_row` and `landed` the values
  // must not be touched.  The compiler sees them as defined by the read asm, so a copy it inserted in between would read early;
  // tests/test_kernels_gpu.py::test_linear_dx_wide_kernel_shapes compares the kernel against the oracle at every row count /
  // ragged shape and is the guard against such a codegen change)
  auto landed = [&](f4_t (&dst)[BG / 4]) {          // all but the eight reads issued after this row's (LDS returns in order)
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(dst[0]), "+v"(dst[1]), "+v"(dst[2]), "+v"(dst[3]), "+v"(dst[4]), "+v"(dst[5]), "+v"(dst[6]), "+v"(dst[7]));
  };
  auto row_fma = [&](const typename LdX4<T>::raw& wr, const f4_t (&dv)[BG / 4]) {
    dg_f32x2_t wa, wb;
    LdX4<T>::cvt(wr, wa, wb);
#pragma unroll
    for (int q = 0; q < BG / 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const dg_f32x2_t d2 = {dv[q][e], dv[q][e]};
        acc[4 * q + e][0] = __builtin_elementwise_fma(d2, wa, acc[4 * q + e][0]);
        acc[4 * q + e][1] = __builtin_elementwise_fma(d2, wb, acc[4 * q + e][1]);
      }
  };
  auto wrow = [&](int o) { return LdX4<T>::load(w + (long long)(o < O ? o : O - 1) * ldw, voff); };   // past the end: a harmless re-read
  // weight rows PD ahead of their arithmetic (two waves per SIMD at ~210 VGPRs: the loads in flight are what hides HBM latency)
  constexpr int PD = 6;
  typename LdX4<T>::raw wq[PD];
#pragma unroll
  for (int u = 0; u < PD; ++u) wq[u] = wrow(u);
  read_row(d[0], 0);
  for (int o = 0; o < O; o += PD) {                  // rows past O: the zero adjoint row times a re-read of the last weight row
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      read_row(d[(u + 1) & 1], o + u + 1 < O ? o + u + 1 : O);
      landed(d[u & 1]);
      row_fma(wq[u], d[u & 1]);
      wq[u] = wrow(o + u + PD);
    }
  }
#pragma unroll
  for (int b = 0; b < BG; ++b) {
    if (b0 + b >= B) break;
    float v[4] = {acc[b][0][0], acc[b][0][1], acc[b][1][0], acc[b][1][1]};
    if (mask) {
      float mv[4];
      ld4(mask + (long long)(b0 + b) * ldmask + k, mv);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(mv[e], slope);
    }
    st4(dx + (long long)(b0 + b) * lddx + k, v);
  }
}

extern "C" int dg_linear_dx(int dtype, int out_dtype, const float* dy, int ldo, const void* w, int64_t ldw, void* dx,
                            int64_t lddx, const void* mask, int64_t ldmask, float mask_slope, int B, int O, int64_t K,
                            void* stream) {
  if (!dy || !w || !dx) return DG_ERR_BAD_ARG;
  if (B <= 0 || O <= 0 || O > 128 || K <= 0 || K % 8 || ldw % 8 || lddx % 8) return DG_ERR_BAD_SHAPE;
  if (mask && ldmask % 8) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (K >= 4096 && (long long)K * 4 < (1ll << 32) && B >= 16) {        // the FC1-sized products: 32 rows per pass, packed FMAs
    dim3 gridw((unsigned)((K / 4 + 255) / 256), (unsigned)((B + 31) / 32));
    if (dtype == DG_F32 && out_dtype == DG_F32) {
      hipLaunchKernelGGL((lin_dx_wide_kernel<float, float>), gridw, dim3(256), 0, st, dy, ldo, (const float*)w, (long long)ldw, (float*)dx,
                         (long long)lddx, (const float*)mask, (long long)ldmask, mask_slope, B, O, (long long)K);
      return dg_check_launch();
    }
    if (dtype == DG_BF16 && out_dtype == DG_BF16) {
      hipLaunchKernelGGL((lin_dx_wide_kernel<bf16_t, bf16_t>), gridw, dim3(256), 0, st, dy, ldo, (const bf16_t*)w, (long long)ldw, (bf16_t*)dx,
                         (long long)lddx, (const bf16_t*)mask, (long long)ldmask, mask_slope, B, O, (long long)K);
      return dg_check_launch();
    }
  }
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

// ---- FC1 weight gradient, all passes of a critic iteration in ONE sweep over the 1.9 GB fp32 gradient -------------------------
// dw[o][k] (=|+=) sum_b dy[b][o] x[b][k] for B <= 128 rows (the real, fake and penalty-tangent rows of wasserstein.py:52
// concatenated), O <= 112.  lin_dw_kernel re-reads x once per 16-row tile of o and read-modify-writes dw once per call (three
// calls per critic iteration: 3.5 ms per step at cfg2; this sweep: 1.1 ms).  A workgroup of 8 waves owns 256 consecutive k of
// ALL 112 gradient rows: wave w holds rows [14w, 14w+14) x 4 k per lane = 56 accumulators as float2 pairs (v_pk_fma_f32: 2 FMAs
// per VALU slot; the sweep is ~45 G FMA at cfg2, comparable to its memory time); the adjoint values are wave-uniform and come
// by SCALAR loads (SGPR operands of the packed FMAs: no LDS, no barrier); x is loaded once per wave through a buffer descriptor
// based at the row (the waves of a workgroup hit L1/L2 for the same 1-KB runs), one register set ahead of its arithmetic.
// `accumulate == 0` writes the result: neither the gradient zero-fill nor the read of dw is needed.
template <typename T, bool ACC, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void lin_dw_wide_kernel(const float* __restrict__ dy, int ldo, const T* __restrict__ x,
                                                                  long long ldx, float* dw, long long lddw, int B, int O,
                                                                  long long K, int cpw) {
  constexpr int OW = 112 / NWV, UB = 4;              // gradient rows per wave (28 or 14), batch rows per register set
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* dyw = dy + wave * OW;                 // wave-uniform: the adjoint values arrive by scalar loads (ldo >= 112)
  for (int c = 0; c < cpw; ++c) {
    const long long k = (((long long)blockIdx.x * cpw + c) * 64 + lane) * 4;
    if (k >= K) break;                               // K % 4 == 0; past-the-end lanes of the last chunk only
    dg_f32x2_t acc[OW][2];
#pragma unroll
    for (int j = 0; j < OW; ++j) acc[j][0] = acc[j][1] = dg_f32x2_t{0.f, 0.f};
    const unsigned voff = (unsigned)k * (unsigned)sizeof(T);   // a row is < 4 GB (checked by the launcher)
    // two register sets of UB rows: the next set's loads are in flight while the current one feeds UB * OW * 2 packed FMAs
    // (with the loads at the head of each set's own arithmetic the memory round trip was in the open: 2.3 ms per sweep)
    typename LdX4<T>::raw xr[2][UB];
    auto fetch = [&](int set, int b0) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {                 // past the last row: a harmless re-read of row B-1
        const int b = b0 + u < B ? b0 + u : B - 1;
        xr[set][u] = LdX4<T>::load(x + (long long)b * ldx, voff);
      }
    };
    auto row_fma = [&](const typename LdX4<T>::raw& r, int b) {
      const float* dp = dyw + (long long)b * ldo;
      dg_f32x2_t xa, xb;
      LdX4<T>::cvt(r, xa, xb);
#pragma unroll
      for (int j = 0; j < OW; ++j) {
        const float dv = dp[j];
        const dg_f32x2_t d2 = {dv, dv};
        acc[j][0] = __builtin_elementwise_fma(d2, xa, acc[j][0]);
        acc[j][1] = __builtin_elementwise_fma(d2, xb, acc[j][1]);
      }
    };
    const int Bm = B / (2 * UB) * (2 * UB);
    fetch(0, 0);
    for (int b0 = 0; b0 < Bm; b0 += 2 * UB) {
      fetch(1, b0 + UB);
#pragma unroll
      for (int u = 0; u < UB; ++u) row_fma(xr[0][u], b0 + u);
      fetch(0, b0 + 2 * UB);
#pragma unroll
      for (int u = 0; u < UB; ++u) row_fma(xr[1][u], b0 + UB + u);
    }
    for (int b = Bm; b < B; ++b) row_fma(LdX4<T>::load(x + (long long)b * ldx, voff), b);     // < 8 leftover rows
#pragma unroll
    for (int j = 0; j < OW; ++j) {
      const int o = wave * OW + j;
      if (o >= O) break;
      float4* p = reinterpret_cast<float4*>(dw + (long long)o * lddw + k);
      float4 t = make_float4(acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]);
      if (ACC) {
        const float4 old = *p;
        t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
      }
      *p = t;
    }
  }
}

extern "C" int dg_linear_dw_wide(int dtype, const float* dy, int ldo, const void* x, int64_t ldx, float* dw, int64_t lddw,
                                 int B, int O, int64_t K, int accumulate, void* stream) {
  if (!dy || !x || !dw) return DG_ERR_BAD_ARG;
  if (B <= 0 || B > 128 || O <= 0 || O > 112 || ldo < 112 || K <= 0 || K >= (1ll << 30) || K % 4 || ldx % 4 || lddw % 4 || ldx < K || lddw < K)
    return DG_ERR_BAD_SHAPE;
  if (dtype != DG_F32 && dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long chunks = (K + 255) / 256;
  const int cpw = chunks >= 8 * 2048 ? 8 : chunks >= 2048 ? 2 : 1;      // >= 2048 workgroups (4 resident rounds of 2 per CU)
  const unsigned grid = (unsigned)((chunks + cpw - 1) / cpw);
  constexpr int NWV = 8;
#define DG_LDW(T, ACC)                                                                                                  \
  hipLaunchKernelGGL((lin_dw_wide_kernel<T, ACC, NWV>), dim3(grid), dim3(64 * NWV), 0, st, dy, ldo, (const T*)x, \
                     (long long)ldx, dw, (long long)lddw, B, O, (long long)K, cpw)
  if (dtype == DG_F32) { if (accumulate) DG_LDW(float, true); else DG_LDW(float, false); }
  else { if (accumulate) DG_LDW(bf16_t, true); else DG_LDW(bf16_t, false); }
#undef DG_LDW
  return dg_check_launch();
}
