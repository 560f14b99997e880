// Dataset preprocessing of the data feed: z-score standardisation of every field over the whole
// record, with the binary land-sea mask exempt (reference DoWnGAN/helpers/gen_experiment_datasets.py:195-233:
// xr_standardize_array = (da - da.mean(skipna)) / da.std(skipna), population std), and the [time, var, lat, lon] staging of
// DoWnGAN/GAN/stage.py:28-31 -- here straight into the HBM-resident [n][H*W][c] store of ResidentLoader.  HBM-bound, one
// read of every field per pass.
#include "dg_internal.h"

// acc[3] (double) += { sum x, sum x^2, count } over the non-NaN elements of x[n]
__global__ void __launch_bounds__(256) moments_kernel(const float* __restrict__ x, long long n, double* __restrict__ acc) {
  double s = 0.0, ss = 0.0, c = 0.0;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (e[k] == e[k]) { s += e[k]; ss += (double)e[k] * e[k]; c += 1.0; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float e = x[(n4 << 2) + threadIdx.x];
    if (e == e) { s += e; ss += (double)e * e; c += 1.0; }
  }
  __shared__ double sh[3][4];
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); c += __shfl_xor(c, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][wave] = s; sh[1][wave] = ss; sh[2][wave] = c; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double t = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
    atomicAdd(acc + threadIdx.x, t);
  }
}

extern "C" int dg_moments(const float* x, int64_t n, double* acc, void* stream) {
  if (!x || !acc || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return DG_ERR_BAD_ARG;
  long long nb = (n / 4 + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(moments_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long long)n, acc);
  return dg_check_launch();
}

// dst[p][k] = (plane_k[p] - mean_k) * inv_std_k for the c fields of a record chunk: planes are [npix] fp32 (npix = times x
// lat x lon of the chunk), dst is the resident store chunk [npix][c] in the compute dtype.  One pixel per thread: the c plane
// reads are coalesced across the wave, the c-element pixel is written contiguously.
template <typename T>
__global__ void __launch_bounds__(256) stage_fields_kernel(dg_field_planes f, long long npix, T* __restrict__ dst) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
    T* d = dst + p * f.c;
    for (int k = 0; k < f.c; ++k) st_elem(d + k, (f.plane[k][p] - f.mean[k]) * f.inv_std[k]);
  }
}

extern "C" int dg_stage_fields(int dtype, const dg_field_planes* f, int64_t npix, void* dst, void* stream) {
  if (!f || !dst || npix <= 0 || f->c < 1 || f->c > DG_MAX_FIELDS) return DG_ERR_BAD_SHAPE;
  for (int k = 0; k < f->c; ++k)
    if (!f->plane[k]) return DG_ERR_BAD_ARG;
  long long nb = (npix + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) hipLaunchKernelGGL(stage_fields_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, *f, (long long)npix, (float*)dst);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(stage_fields_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, *f, (long long)npix, (bf16_t*)dst);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
