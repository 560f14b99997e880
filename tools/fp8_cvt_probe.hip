// What v_cvt_pk_fp8_f32 does on gfx950 with values outside the E4M3 range, without an explicit clamp in front:
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_cvt_probe.hip -o tools/fp8_cvt_probe && tools/fp8_cvt_probe
// Decides whether pack_fp8x16 (csrc/dg_internal.h) needs its fmin/fmax clamp (which launders NaN into -448) or can leave
// saturation and NaN propagation to the converter.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void probe(const float* x, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[i], w, false);
  out[i] = (unsigned)w & 0xffffu;
}

int main() {
  const float h[] = {0.f, 1.f, 447.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1e9f, -1e9f, INFINITY, -INFINITY, NAN, -NAN, 1e-9f, 0.0019f, 0.001f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; unsigned* o; unsigned r[32];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(unsigned) * n);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o, n);
  hipMemcpy(r, o, sizeof(unsigned) * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%14g -> 0x%02x 0x%02x\n", h[i], r[i] & 0xff, (r[i] >> 8) & 0xff);
  return 0;
}
