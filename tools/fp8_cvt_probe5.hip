// v_cvt_scalef32_pk_fp8_bf16, second part: byte order of the two results, the word select, and the scale field 0 (blocks of zeros).
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_cvt_probe5.hip -o tools/fp8_cvt_probe5
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) short s2;
__global__ void probe(unsigned* o) {
  float sc = 1.0f;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 1" : "+v"(sc));
  const unsigned pkA = 0x3f80u | (0x4000u << 16);            // lo = 1.0, hi = 2.0
  const unsigned pkB = 0x4040u | (0xc080u << 16);            // lo = 3.0, hi = -4.0
  s2 w = {0x1111, 0x2222};
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(bf2, pkA), sc, false);
  o[0] = __builtin_bit_cast(unsigned, w);
  w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, __builtin_bit_cast(bf2, pkB), sc, true);
  o[1] = __builtin_bit_cast(unsigned, w);
  // scale exponent field 0 (bits 0x00000000) and 1, on zeros and on a tiny value
  const float z0 = __uint_as_float(0u), z1 = __uint_as_float(1u << 23);
  const unsigned pkZ = 0x0000u | (0x0080u << 16);            // lo = 0, hi = 2^-126 (smallest normal bf16)
  s2 z = {0, 0};
  z = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(z, __builtin_bit_cast(bf2, pkZ), z0, false);
  o[2] = __builtin_bit_cast(unsigned, z);
  s2 y = {0, 0};
  y = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(y, __builtin_bit_cast(bf2, pkZ), z1, false);
  o[3] = __builtin_bit_cast(unsigned, y);
  // a non-power-of-two scale: is the mantissa ignored?
  float s3 = 3.0f;
  s2 t = {0, 0};
  t = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(t, __builtin_bit_cast(bf2, pkB), s3, false);
  o[4] = __builtin_bit_cast(unsigned, t);
  unsigned keep = o[4];
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0\n\ts_nop 1" : "+v"(keep));
}
int main() {
  unsigned *o, h[8];
  hipMalloc(&o, 32); hipMemset(o, 0, 32);
  probe<<<1, 1>>>(o);
  hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
  printf("(1.0, 2.0) -> low word, old = 0x2222_1111:  0x%08x   (E4M3: 1.0 = 0x38, 2.0 = 0x40)\n", h[0]);
  printf("(3.0, -4.0) -> high word:                    0x%08x   (3.0 = 0x44, -4.0 = 0xc8)\n", h[1]);
  printf("(0, 2^-126) / scale bits 0x00000000:         0x%08x\n", h[2]);
  printf("(0, 2^-126) / scale bits 0x00800000 (2^-126): 0x%08x   (2^-126 / 2^-126 = 1.0 = 0x38)\n", h[3]);
  printf("(3.0, -4.0) / 3.0f:                          0x%08x   (exact division: 1.0 = 0x38, -1.33 = 0xb3; exponent only (/2): 1.5 = 0x3c, -2 = 0xc0)\n", h[4]);
  return 0;
}
