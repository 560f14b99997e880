// What MODE.FP16_OVFL does to v_mfma_f32_16x16x32_bf16 (and the scaled fp8 MFMA): NaN / Inf in the accumulator input and in the operands.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_cvt_probe3.hip -o tools/fp8_cvt_probe3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__global__ void probe(float* out, int ovfl, float cval, float aval) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 1" ::: "memory");
  const int lane = threadIdx.x;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
  if (lane == 3) a[2] = (__bf16)aval;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (lane == 5) c[1] = cval;
  f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  i32x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = 0x38383838; fb[i] = 0x38383838; }      // E4M3 1.0
  f32x4 c2 = {0.f, 0.f, 0.f, 0.f};
  if (lane == 5) c2[1] = cval;
  f32x4 d2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, c2, 0, 0, 0, 127, 0, 127);
  for (int e = 0; e < 4; ++e) { out[lane * 8 + e] = d[e]; out[lane * 8 + 4 + e] = d2[e]; }
}

int main() {
  float* o; float r[64 * 8];
  hipMalloc(&o, sizeof(r));
  const float cvals[] = {NAN, INFINITY, 1e30f, 7.0f};
  const float avals[] = {1.0f, NAN, INFINITY};
  for (int ovfl = 0; ovfl < 2; ++ovfl)
    for (float cv : cvals)
      for (float av : avals) {
        probe<<<1, 64>>>(o, ovfl, cv, av);
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        // lane 5 element 1 holds the C probe (row 4*(5/16)+1 = 1, col 5); lane 3's A row 3 taints D row 3: lane 0..15 with group 0, element 3
        printf("ovfl %d  C=%-8g A=%-4g : bf16 mfma D[c-slot]=%-12g D[a-row]=%-12g D[plain]=%-8g | fp8 mfma D[c-slot]=%-12g D[plain]=%g\n", ovfl, cv, av,
               r[5 * 8 + 1], r[0 * 8 + 3], r[20 * 8 + 0], r[5 * 8 + 4 + 1], r[20 * 8 + 4]);
      }
  return 0;
}
