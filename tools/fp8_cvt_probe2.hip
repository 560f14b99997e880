// v_cvt_pk_fp8_f32 on gfx950 with MODE.FP16_OVFL set (s_setreg hwreg(HW_REG_MODE, 23, 1)): does the converter SATURATE at +-448 then, so
// that pack_fp8x16 (csrc/dg_internal.h) can drop its v_med3 clamp (16 % of an fp8 epilogue's VALU)?  Also: what the bit does to the
// bf16 conversion and whether it survives to the end of the wave.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_cvt_probe2.hip -o tools/fp8_cvt_probe2 && tools/fp8_cvt_probe2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void probe(const float* x, unsigned* out, int n, int ovfl) {
  const int i = threadIdx.x;
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");
  if (i >= n) return;
  int w = 0;
  float v = x[i];
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %1" : "+v"(w) : "v"(v));
  unsigned mode = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 32)" : "=s"(mode));
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  unsigned b;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(b) : "v"(v));
  out[3 * i] = (unsigned)w & 0xffffu;
  out[3 * i + 1] = mode;
  out[3 * i + 2] = b & 0xffffu;
}

int main() {
  const float h[] = {0.f, 1.f, 447.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1e9f, -1e9f, INFINITY, -INFINITY, NAN, -NAN, 1e-9f, 0.0019f, 0.001f, -500.f, 3.4e38f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; unsigned* o; unsigned r[96];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(unsigned) * 3 * n);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    probe<<<1, 64>>>(d, o, n, ovfl);
    hipMemcpy(r, o, sizeof(unsigned) * 3 * n, hipMemcpyDeviceToHost);
    printf("FP16_OVFL = %d (MODE = 0x%08x)\n", ovfl, r[1]);
    for (int i = 0; i < n; ++i) printf("%14g -> fp8 0x%02x 0x%02x   bf16 0x%04x\n", h[i], r[3 * i] & 0xff, (r[3 * i] >> 8) & 0xff, r[3 * i + 2]);
  }
  return 0;
}
