// Two opt-in debugging aids of the train step (both off by default; nothing here is on the benchmarked path).
//
// 1. Deterministic reductions.  The weight-gradient kernels split their pixel range over workgroups and accumulate the partial
//    tiles with fp32 atomics; the small reductions (column sums, per-image sums of squares, L1 / squared-difference sums, the
//    split-K Linear forward) end in one atomic per workgroup or wave.  The order of those additions differs from run to run, so
//    two runs agree only to fp32 re-association.  With a caller-owned workspace registered here, every such launch writes the
//    partial of workgroup / split / wave c into COPY c of its target region inside the workspace (a unique writer per element:
//    the kernels keep their atomicAdd, onto zero) and `det_reduce_kernel` adds the copies to the target in index order.  A launch
//    whose copies do not fit the workspace runs with as many splits as fit (one split needs no workspace: unique writers).
//
// 2. dg_count_nonfinite: one launch that counts the NaN / Inf elements of up to eight buffers -- the native stand-in for the
//    reference's torch.autograd.set_detect_anomaly(True) (DoWnGAN/GAN/wasserstein.py:13), which checks every op's output;
//    TrainEngine(check_finite=True) runs it once per iteration over the scalars, the gradient buffers and the generated batch.
#include "dg_internal.h"

static std::atomic<float*> g_det_ws{nullptr};
static std::atomic<long long> g_det_floats{0};

extern "C" int dg_set_deterministic_workspace(void* ws, int64_t bytes) {
  if (ws && (bytes < (1 << 20) || (reinterpret_cast<uintptr_t>(ws) & 15))) return DG_ERR_BAD_ARG;
  g_det_floats.store(ws ? bytes / 4 : 0, std::memory_order_release);
  g_det_ws.store(reinterpret_cast<float*>(ws), std::memory_order_release);
  return DG_OK;
}
extern "C" int dg_deterministic(void) { return g_det_ws.load(std::memory_order_acquire) != nullptr; }

bool dg_det_on() { return g_det_ws.load(std::memory_order_acquire) != nullptr; }

int dg_det_begin(long long stride, int want, hipStream_t st, DetPlan* p) {
  p->ws = nullptr; p->stride = stride; p->copies = want < 1 ? 1 : want;
  float* ws = g_det_ws.load(std::memory_order_acquire);
  if (!ws) return DG_OK;
  const long long fit = stride > 0 ? g_det_floats.load(std::memory_order_acquire) / stride : 0;
  if (fit < 2 || want < 2) { p->copies = 1; return DG_OK; }           // one copy: every element has one writer, straight into the target
  if (p->copies > fit) p->copies = (int)fit;
  p->ws = ws;
  if (hipMemsetAsync(ws, 0, (size_t)p->copies * (size_t)stride * 4, st) != hipSuccess) return DG_ERR_LAUNCH;
  return DG_OK;
}

__global__ void det_reduce_kernel(const float* ws, long long stride, int copies, float* target, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < copies; ++c) s += ws[(long long)c * stride + i];
    target[i] += s;
  }
}

int dg_det_reduce(const DetPlan& p, long long off, float* target, long long n, hipStream_t st) {
  if (!p.ws || n <= 0) return DG_OK;
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, p.ws + off, p.stride, p.copies, target, n);
  return dg_check_launch();
}

// ------------------------------------------------------------------ non-finite census
struct FiniteArgs { const void* ptr[DG_FINITE_MAX]; long long n[DG_FINITE_MAX]; int dtype[DG_FINITE_MAX]; };

__global__ __launch_bounds__(256) void count_nonfinite_kernel(const FiniteArgs a, unsigned* counts) {
  const int b = blockIdx.y;
  const long long n = a.n[b];
  unsigned c = 0;
  if (a.dtype[b] == DG_F32) {
    const unsigned* p = reinterpret_cast<const unsigned*>(a.ptr[b]);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      c += (p[i] & 0x7f800000u) == 0x7f800000u;
  } else {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(a.ptr[b]);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
      c += (p[i] & 0x7f80u) == 0x7f80u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(counts + b, c);
}

extern "C" int dg_count_nonfinite(const dg_finite_bufs* bufs, uint32_t* counts, void* stream) {
  if (!bufs || !counts || bufs->nbuf < 1 || bufs->nbuf > DG_FINITE_MAX) return DG_ERR_BAD_ARG;
  FiniteArgs a{};
  long long nmax = 0;
  for (int i = 0; i < bufs->nbuf; ++i) {
    if (!bufs->ptr[i] || bufs->n[i] < 0) return DG_ERR_BAD_ARG;
    if (bufs->dtype[i] != DG_F32 && bufs->dtype[i] != DG_BF16) return DG_ERR_BAD_DTYPE;
    a.ptr[i] = bufs->ptr[i]; a.n[i] = bufs->n[i]; a.dtype[i] = bufs->dtype[i];
    if (bufs->n[i] > nmax) nmax = bufs->n[i];
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, sizeof(uint32_t) * bufs->nbuf, st) != hipSuccess) return DG_ERR_LAUNCH;
  long long nb = (nmax + 256 * 16 - 1) / (256 * 16);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)nb, bufs->nbuf), dim3(256), 0, st, a, counts);
  return dg_check_launch();
}
