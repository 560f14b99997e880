// Store-pattern probe, second pass (round 4): what bounds the epilogue of the merged stride-2 data gradient (conv_halo.hip, SEG)?
// A parity class writes 16x16 pixels x 128 channels of dx = 256-B pixels at a 512-B stride; the halo kernel has TWO 4-wave
// workgroups per CU (78 KB of LDS each), every lane issues 16 stores of 16 B, and in-kernel stamps put that epilogue at ~20k
// cycles (3 TB/s chip-wide) against the 6 TB/s a plain store kernel reaches.  Variables here:
//   occ      resident workgroups per CU, forced with dynamic LDS (the plain probe ran at 8)
//   pattern  P: one class per workgroup (256 B @ 512 B), Q: eight waves, both px classes of a row (512 contiguous bytes per pixel pair),
//            R: one class per workgroup but the four classes of a tile by consecutive workgroups (seg = 2 order)
//   loads    n 16-byte buffer loads per lane issued BEFORE the stores from a 2-GB region (stands for the partner's patch / weight traffic)
// hipcc --offload-arch=gfx950 -O3 tools/store_probe2.hip -o tools/store_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u4;

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// PAT 0 (P): 4 waves, class (cls) of tile; blockIdx -> (img, cls, ty, tx) via remap.  PAT 2 (R): (img, ty, cls, tx) without remap.
// PAT 1 (Q): 8 waves: waves 0-3 px = 0, waves 4-7 px = 1 of class row py; blockIdx -> (img, py, ty, tx)
template <int PAT>
__global__ void probe(char* y, const char* src, int tiles, unsigned nwg, int nloads, unsigned srcmask) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wq = wave & 3, wh = wave >> 2;
  unsigned t = PAT == 2 ? blockIdx.x : xcd_remap(blockIdx.x, nwg);
  int tx, ty, img, py, px;
  if (PAT == 0) { tx = t % tiles; t /= tiles; ty = t % tiles; t /= tiles; py = (t >> 1) & 1; px = t & 1; img = t >> 2; }
  else if (PAT == 2) { tx = t % tiles; t /= tiles; py = (t >> 1) & 1; px = t & 1; t >>= 2; ty = t % tiles; img = t / tiles; }
  else { tx = t % tiles; t /= tiles; ty = t % tiles; t /= tiles; py = t & 1; px = wh; img = t >> 1; }
  const long long Wd = 1024;
  char* base = y + (((long long)img * 1024 + ty * 32 + py) * Wd + tx * 32 + px) * 256;
  unsigned v = tid * 0x9e3779b9u + blockIdx.x;
  // loads first (they retire before the stores issue; a real epilogue's mask words / the partner's patch)
  u4 acc = {0, 0, 0, 0};
  const unsigned lbase = (blockIdx.x * 4096u + tid * 16u);
  for (int i = 0; i < nloads; ++i) {
    const u4 r = *reinterpret_cast<const u4*>(src + (((size_t)lbase + (size_t)i * 65536u * 64u) & srcmask));
    acc[0] ^= r[0]; acc[1] ^= r[1]; acc[2] ^= r[2]; acc[3] ^= r[3];
  }
  v ^= acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long row = (long long)(wq * 4 + i) * 2 * Wd;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        u4 tv = {v + h, v ^ (unsigned)i, v + k, v};
        *reinterpret_cast<u4*>(base + (row + l15 * 2) * 256 + (64 * h + 16 * g + 8 * k) * 2) = tv;
      }
  }
  if (lds[0] == 77 && tid == 9999) y[0] = 1;      // keep the dynamic LDS allocation
}

int main() {
  const int N = 8;
  char *y, *src;
  const size_t bytes = (size_t)N * 1024 * 1024 * 256, sbytes = (size_t)1 << 31;
  if (hipMalloc(&y, bytes) != hipSuccess || hipMalloc(&src, sbytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, sbytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = 32;
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int lds_for_occ[] = {0, 156 * 1024, 78 * 1024, 52 * 1024, 39 * 1024};       // index = workgroups per CU (0: unlimited)
  for (int nloads = 0; nloads <= 8; nloads += 8)
    for (int pat = 0; pat < 3; ++pat)
      for (int occ : {0, 4, 2, 1}) {
        if (pat == 1 && occ != 0 && occ != 1 && occ != 2) continue;
        const unsigned nwg = pat == 1 ? (unsigned)(tiles * tiles * N * 2) : (unsigned)(tiles * tiles * N * 4);
        const int nt = pat == 1 ? 512 : 256;
        const int ldsb = lds_for_occ[occ];
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
          hipEventRecord(e0);
          if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(nt), ldsb, 0, y, src, tiles, nwg, nloads, (unsigned)(sbytes - 1) & ~15u);
          if (pat == 1) hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(nt), ldsb, 0, y, src, tiles, nwg, nloads, (unsigned)(sbytes - 1) & ~15u);
          if (pat == 2) hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(nt), ldsb, 0, y, src, tiles, nwg, nloads, (unsigned)(sbytes - 1) & ~15u);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (it > 0 && ms < best) best = ms;
        }
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        const double wb = (double)N * 1024 * 1024 * 256;
        printf("loads/lane %d pattern %c  wg/CU %s%d: %.3f ms  %.2f TB/s written (+%.2f TB/s read)\n", nloads, "PQR"[pat], occ ? "" : "max=", occ ? occ : 8,
               best, wb / best * 1e-9, (double)nwg * nt * 16.0 * nloads / best * 1e-9);
      }
  return 0;
}
