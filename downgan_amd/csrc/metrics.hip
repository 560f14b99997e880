// MS-SSIM of the per-step metrics pass (DoWnGAN/GAN/losses.py:12-38 via mlflow_tools/mlflow_epoch.py:53-63).
// HBM/LDS-bound stencil + reduction kernels, fp32 throughout (the reference computes the metric in fp32).
//   min/max per channel -> normalise to planar fp32 -> per scale: fused 5-quantity separable Gaussian + SSIM/CS maps +
//   plane sums; 2x2 average pooling between scales -> weighted geometric combination.
#include "dg_internal.h"

// ------------------------------------------------------------------ per-channel min / max
template <typename T>
__global__ __launch_bounds__(256) void minmax_partial_kernel(const T* x, long long pixels, long long ld, int C, float* partial) {
  __shared__ float smin[4][DG_SSIM_MAX_CH], smax[4][DG_SSIM_MAX_CH];
  float mn[DG_SSIM_MAX_CH], mx[DG_SSIM_MAX_CH];
#pragma unroll
  for (int c = 0; c < DG_SSIM_MAX_CH; ++c) { mn[c] = INFINITY; mx[c] = -INFINITY; }
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
    const T* px = x + p * ld;
#pragma unroll
    for (int c = 0; c < DG_SSIM_MAX_CH; ++c)
      if (c < C) { const float v = ld_elem(px + c); mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
  }
#pragma unroll
  for (int c = 0; c < DG_SSIM_MAX_CH; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6][c] = mn[c]; smax[threadIdx.x >> 6][c] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    float a = smin[0][c], b = smax[0][c];
    for (int w = 1; w < 4; ++w) { a = fminf(a, smin[w][c]); b = fmaxf(b, smax[w][c]); }
    partial[((long long)blockIdx.x * C + c) * 2] = a;
    partial[((long long)blockIdx.x * C + c) * 2 + 1] = b;
  }
}
__global__ void minmax_finish_kernel(const float* partial, int C, float* minmax) {
  const int c = blockIdx.x;
  float a = INFINITY, b = -INFINITY;
  for (int i = threadIdx.x; i < DG_MINMAX_PARTS; i += 64) {
    a = fminf(a, partial[((long long)i * C + c) * 2]);
    b = fmaxf(b, partial[((long long)i * C + c) * 2 + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o, 64)); b = fmaxf(b, __shfl_xor(b, o, 64)); }
  if (threadIdx.x == 0) { minmax[2 * c] = a; minmax[2 * c + 1] = b; }
}
extern "C" int dg_minmax_partial(int dtype, const void* x, int64_t pixels, int64_t ld, int C, float* partial, void* stream) {
  if (!x || !partial || pixels <= 0 || C <= 0 || C > DG_SSIM_MAX_CH || ld < C) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DG_F32) hipLaunchKernelGGL(minmax_partial_kernel<float>, dim3(DG_MINMAX_PARTS), dim3(256), 0, st, (const float*)x, (long long)pixels, (long long)ld, C, partial);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(minmax_partial_kernel<bf16_t>, dim3(DG_MINMAX_PARTS), dim3(256), 0, st, (const bf16_t*)x, (long long)pixels, (long long)ld, C, partial);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
extern "C" int dg_minmax_finish(const float* partial, int C, float* minmax, void* stream) {
  if (!partial || !minmax || C <= 0 || C > DG_SSIM_MAX_CH) return DG_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(minmax_finish_kernel, dim3(C), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), partial, C, minmax);
  return dg_check_launch();
}

// ------------------------------------------------------------------ (x - min_c) / (max_c - min_c), NHWC -> planar fp32
template <typename T>
__global__ void normalise_planar_kernel(const T* x, int N, long long HW, long long ld, int C, const float* minmax, float* out) {
  const long long total = (long long)N * HW;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW, hw = p - n * HW;
    const T* px = x + p * ld;
    for (int c = 0; c < C; ++c) {
      const float mn = minmax[2 * c], mx = minmax[2 * c + 1];
      out[(n * C + c) * HW + hw] = (ld_elem(px + c) - mn) / (mx - mn);
    }
  }
}
extern "C" int dg_normalise_planar(int dtype, const void* x, int N, int H, int W, int64_t ld, int C, const float* minmax,
                                   float* out, void* stream) {
  if (!x || !minmax || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > DG_SSIM_MAX_CH || ld < C) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long HW = (long long)H * W, total = (long long)N * HW;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (dtype == DG_F32) hipLaunchKernelGGL(normalise_planar_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)x, N, HW, (long long)ld, C, minmax, out);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(normalise_planar_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)x, N, HW, (long long)ld, C, minmax, out);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}

// ------------------------------------------------------------------ one scale: SSIM / CS plane sums
// Workgroup = a 32x32 tile of the 'valid' map of one plane.  LDS: the (32+win-1)^2 input tiles of X and Y, then the five
// column-filtered quantities (x, y, x^2, y^2, xy: the Gaussian runs along H first, like the restated package), then each
// thread finishes 4 pixels along W and the workgroup adds two partial sums to the plane's accumulators.
constexpr int ST = 32;
__global__ __launch_bounds__(256) void ssim_level_kernel(const float* X, const float* Y, int H, int W, dg_ssim_params p, float* sums) {
  constexpr int MAXE = ST + DG_SSIM_MAX_WIN - 1;
  __shared__ float sx[MAXE][MAXE + 1], sy[MAXE][MAXE + 1];
  __shared__ float sv[5][ST][MAXE + 1];
  __shared__ float red[2][4];
  const int win = p.win, E = ST + win - 1;
  const int Hv = H - win + 1, Wv = W - win + 1;
  const int plane = blockIdx.z, ty0 = blockIdx.y * ST, tx0 = blockIdx.x * ST;
  const float* xp = X + (long long)plane * H * W;
  const float* yp = Y + (long long)plane * H * W;
  for (int i = threadIdx.x; i < E * E; i += 256) {
    const int r = i / E, c = i - r * E;
    const int gy = ty0 + r, gx = tx0 + c;
    const bool ok = gy < H && gx < W;
    sx[r][c] = ok ? xp[(long long)gy * W + gx] : 0.f;
    sy[r][c] = ok ? yp[(long long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ST * E; i += 256) {
    const int r = i / E, c = i - r * E;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
    for (int k = 0; k < win; ++k) {
      const float g = p.g[k], xv = sx[r + k][c], yv = sy[r + k][c];
      a0 = fmaf(g, xv, a0); a1 = fmaf(g, yv, a1);
      a2 = fmaf(g, xv * xv, a2); a3 = fmaf(g, yv * yv, a3); a4 = fmaf(g, xv * yv, a4);
    }
    sv[0][r][c] = a0; sv[1][r][c] = a1; sv[2][r][c] = a2; sv[3][r][c] = a3; sv[4][r][c] = a4;
  }
  __syncthreads();
  float s_ssim = 0.f, s_cs = 0.f;
  for (int i = threadIdx.x; i < ST * ST; i += 256) {
    const int r = i / ST, c = i - r * ST;
    if (ty0 + r < Hv && tx0 + c < Wv) {
      float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < win; ++k) {
        const float g = p.g[k];
#pragma unroll
        for (int q = 0; q < 5; ++q) m[q] = fmaf(g, sv[q][r][c + k], m[q]);
      }
      const float mu1 = m[0], mu2 = m[1];
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
      const float cs = (2.f * s12 + p.C2) / (s1 + s2 + p.C2);
      s_cs += cs;
      s_ssim += ((2.f * mu12 + p.C1) / (mu1_sq + mu2_sq + p.C1)) * cs;
    }
  }
  s_ssim = wave_sum(s_ssim); s_cs = wave_sum(s_cs);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_ssim; red[1][threadIdx.x >> 6] = s_cs; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    atomicAdd(sums + 2 * plane + threadIdx.x, t);
  }
}
extern "C" int dg_ssim_level(const float* X, const float* Y, int planes, int H, int W, const dg_ssim_params* p, float* sums,
                             void* stream) {
  if (!X || !Y || !p || !sums || planes <= 0 || planes > 65535) return DG_ERR_BAD_SHAPE;
  if (p->win < 1 || p->win > DG_SSIM_MAX_WIN || H < p->win || W < p->win) return DG_ERR_BAD_SHAPE;
  const int Hv = H - p->win + 1, Wv = W - p->win + 1;
  dim3 grid((Wv + ST - 1) / ST, (Hv + ST - 1) / ST, planes);
  hipLaunchKernelGGL(ssim_level_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), X, Y, H, W, *p, sums);
  return dg_check_launch();
}

// ------------------------------------------------------------------ avg_pool2d(kernel 2, stride 2, padding = size % 2, zeros counted)
__global__ void avgpool2_kernel(const float* in, float* out, int planes, int H, int W, int Ho, int Wo, int ph, int pw) {
  const long long total = (long long)planes * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    const long long t = i / Wo;
    const int ho = (int)(t % Ho);
    const long long pl = t / Ho;
    const float* ip = in + pl * H * W;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int y = 2 * ho - ph + a, x = 2 * wo - pw + b;
        if (y >= 0 && y < H && x >= 0 && x < W) s += ip[(long long)y * W + x];
      }
    out[i] = s * 0.25f;
  }
}
extern "C" int dg_avgpool2(const float* in, float* out, int planes, int H, int W, void* stream) {
  if (!in || !out || planes <= 0 || H < 2 || W < 2) return DG_ERR_BAD_SHAPE;
  const int ph = H & 1, pw = W & 1;
  const int Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
  const long long total = (long long)planes * Ho * Wo;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, planes, H, W, Ho, Wo, ph, pw);
  return dg_check_launch();
}

// ------------------------------------------------------------------ prod_l relu(term_l)^w_l per plane, summed over planes
__global__ void msssim_finish_kernel(const float* sums, int levels, int planes, dg_msssim_combine cmb, float* out) {
  float acc = 0.f;
  for (int pl = threadIdx.x; pl < planes; pl += 64) {
    float v = 1.f;
    for (int l = 0; l < levels; ++l) {
      const float* s = sums + ((long long)l * planes + pl) * 2;
      const float term = fmaxf((l == levels - 1 ? s[0] : s[1]) * cmb.inv_count[l], 0.f);   // ssim at the last scale, cs before
      v *= powf(term, cmb.weight[l]);
    }
    acc += v;
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[0] = acc;
}
extern "C" int dg_msssim_finish(const float* sums, int levels, int planes, const dg_msssim_combine* cmb, float* out, void* stream) {
  if (!sums || !cmb || !out || levels < 1 || levels > DG_SSIM_MAX_LEVELS || planes <= 0) return DG_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(msssim_finish_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sums, levels, planes, *cmb, out);
  return dg_check_launch();
}

// ------------------------------------------------------------------ divergence / vorticity metrics (losses.py:119-193)
// One pass over both NHWC tensors: forward differences of channel 0 along H and channel 1 along W on the [1:, 1:] window,
// div = dudy + dvdx, vort = dvdx - dudy, and for each of the two fields the five moments
// {sum r, sum r^2, sum f, sum f^2, sum r*f} (double) from which the host forms the std-normalised MSE.
template <typename T>
__global__ __launch_bounds__(256) void div_vort_kernel(const T* hr, long long ldr, const T* fk, long long ldf, int N, int H, int W, double* sums) {
  __shared__ double red[4][10];
  double s[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) s[q] = 0.0;
  const long long per = (long long)(H - 1) * (W - 1), total = (long long)N * per;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / per, r = i - n * per;
    const int y = (int)(r / (W - 1)) + 1, x = (int)(r % (W - 1)) + 1;
    const long long p = (n * H + y) * W + x;
    float d[2], v[2];
    {
      const T* a = hr + p * ldr;
      const float dudy = ld_elem(a) - ld_elem(a - (long long)W * ldr), dvdx = ld_elem(a + 1) - ld_elem(a - ldr + 1);
      d[0] = dudy + dvdx; v[0] = dvdx - dudy;
    }
    {
      const T* a = fk + p * ldf;
      const float dudy = ld_elem(a) - ld_elem(a - (long long)W * ldf), dvdx = ld_elem(a + 1) - ld_elem(a - ldf + 1);
      d[1] = dudy + dvdx; v[1] = dvdx - dudy;
    }
    s[0] += d[0]; s[1] += (double)d[0] * d[0]; s[2] += d[1]; s[3] += (double)d[1] * d[1]; s[4] += (double)d[0] * d[1];
    s[5] += v[0]; s[6] += (double)v[0] * v[0]; s[7] += v[1]; s[8] += (double)v[1] * v[1]; s[9] += (double)v[0] * v[1];
  }
#pragma unroll
  for (int q = 0; q < 10; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = s[q];
  }
  __syncthreads();
  if (threadIdx.x < 10) atomicAdd(sums + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
extern "C" int dg_div_vort_sums(int dtype, const void* hr, int64_t ldhr, const void* fake, int64_t ldfake, int N, int H, int W,
                                double* sums, void* stream) {
  if (!hr || !fake || !sums || N <= 0 || H < 2 || W < 2 || ldhr < 2 || ldfake < 2) return DG_ERR_BAD_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)N * (H - 1) * (W - 1);
  long long nb = (total + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (dtype == DG_F32) hipLaunchKernelGGL(div_vort_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)hr, (long long)ldhr, (const float*)fake, (long long)ldfake, N, H, W, sums);
  else if (dtype == DG_BF16) hipLaunchKernelGGL(div_vort_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)hr, (long long)ldhr, (const bf16_t*)fake, (long long)ldfake, N, H, W, sums);
  else return DG_ERR_BAD_DTYPE;
  return dg_check_launch();
}
