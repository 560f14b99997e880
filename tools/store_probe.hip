// Store-pattern probe: how fast can 16x16-pixel x 128-channel bf16 tiles be written with the accumulator layouts an MFMA
// epilogue can produce?  (A) 8 B per lane, 32-B runs per pixel (current halo epilogue), (B) 16 B per lane, 64-B... 128-B runs
// per pixel (permuted weight rows), (C) 16 B per lane, 1-KB runs per wave instruction (LDS-transposed).  Destination pixel
// stride 1 or 2 (stride-2 data-gradient classes).   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int PAT>
__global__ __launch_bounds__(256, 2) void probe(char* y, int W, int H, int tiles_x, int tiles_y, int pstride, unsigned nwg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  // same XCD-chunked tile order as the conv kernels
  unsigned bid = blockIdx.x, q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  unsigned tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tx0 = (tile % tiles_x) * 16; tile /= tiles_x;
  const int ty0 = (tile % tiles_y) * 16;
  const int img = tile / tiles_y;
  const long long Wd = (long long)W * pstride, Hd = (long long)H * pstride;
  char* base = y + (((long long)img * Hd + (long long)ty0 * pstride) * Wd + (long long)tx0 * pstride) * 256;
  const unsigned v = tid * 0x9e3779b9u + blockIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long row = (long long)(wave * 4 + i) * pstride * Wd;
    if (PAT == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        u2 t = {v + j, v ^ (unsigned)i};
        *reinterpret_cast<u2*>(base + (row + l15 * pstride) * 256 + (16 * j + 4 * g) * 2) = t;
      }
    } else if (PAT == 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          u4 t = {v + h, v ^ (unsigned)i, v + k, v};
          *reinterpret_cast<u4*>(base + (row + l15 * pstride) * 256 + (64 * h + 16 * g + 8 * k) * 2) = t;
        }
    } else {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        u4 t = {v + qq, v ^ (unsigned)i, v, v};
        *reinterpret_cast<u4*>(base + (row + (qq * 4 + g) * pstride) * 256 + l15 * 16) = t;
      }
    }
  }
}

int main() {
  const int N = 8;
  char* y;
  const size_t bytes = (size_t)N * 1024 * 1024 * 256;
  if (hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pstride = 1; pstride <= 2; ++pstride) {
    const int W = 1024 / pstride;
    const int tiles = W / 16;
    const unsigned nwg = (unsigned)(tiles * tiles * N);
    for (int pat = 0; pat < 3; ++pat) {
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        const int reps = pstride == 1 ? 1 : 1;
        if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), 0, 0, y, W, W, tiles, tiles, pstride, nwg);
        if (pat == 1) hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), 0, 0, y, W, W, tiles, tiles, pstride, nwg);
        if (pat == 2) hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(256), 0, 0, y, W, W, tiles, tiles, pstride, nwg);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
        (void)reps;
      }
      const double wb = (double)nwg * 65536.0;
      printf("pixel stride %d pattern %c: %.3f ms  %.2f TB/s written\n", pstride, "ABC"[pat], best, wb / best * 1e-9);
    }
  }
  return 0;
}
