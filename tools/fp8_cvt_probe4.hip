// v_cvt_scalef32_pk_fp8_bf16 (gfx950): two packed bf16 -> two E4M3 bytes "scaled by an fp32 scale".  Does it equal the epilogue's current
// path -- unpack to fp32, multiply by 2^(127 - e), v_cvt_pk_fp8_f32 under MODE.FP16_OVFL (saturating) -- for EVERY bf16 bit pattern, and
// is the scale operand a divisor 2^(e - 127) or a factor?  What happens beyond +-448, with Inf / NaN, with the mode bit off?
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_cvt_probe4.hip -o tools/fp8_cvt_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) short s2;

__global__ void probe(unsigned* mism, unsigned* first, int e, int use_factor, int ovfl_new) {
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;       // bf16 bit pattern 0..65535
  const unsigned pkw = b | (b << 16);
  const float x = __uint_as_float(b << 16);
  const float inv = __uint_as_float((unsigned)(254 - e) << 23);    // 2^(127 - e)
  const float scl = __uint_as_float((unsigned)e << 23);            // 2^(e - 127)
  int ref = 0;
  float y = x * inv;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 1" : "+v"(y));
  ref = __builtin_amdgcn_cvt_pk_fp8_f32(y, y, ref, false);
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0\n\ts_nop 1" : "+v"(ref));
  float sc = use_factor ? inv : scl;
  if (ovfl_new) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 1" : "+v"(sc));
  s2 old = {0, 0};
  s2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, __builtin_bit_cast(bf2, pkw), sc, false);
  unsigned got = __builtin_bit_cast(unsigned, r);
  if (ovfl_new) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0\n\ts_nop 1" : "+v"(got));
  if ((got & 0xffffu) != ((unsigned)ref & 0xffffu)) {
    const unsigned k = atomicAdd(mism, 1u);
    if (k < 8) { first[3 * k] = b; first[3 * k + 1] = (unsigned)ref & 0xffffu; first[3 * k + 2] = got & 0xffffu; }
  }
}

int main() {
  unsigned *m, *f, hm, hf[24];
  hipMalloc(&m, 4); hipMalloc(&f, sizeof(hf));
  for (int ovfl = 0; ovfl < 2; ++ovfl)
    for (int fac = 0; fac < 2; ++fac)
      for (int e : {127, 120, 135, 100, 1, 250}) {
        hipMemset(m, 0, 4); hipMemset(f, 0, sizeof(hf));
        probe<<<256, 256>>>(m, f, e, fac, ovfl);
        hipMemcpy(&hm, m, 4, hipMemcpyDeviceToHost); hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
        printf("FP16_OVFL %d  scale operand = %s  e = %3d: %5u of 65536 bf16 patterns differ", ovfl, fac ? "2^(127-e) (factor) " : "2^(e-127) (divisor)", e, hm);
        for (unsigned k = 0; k < (hm < 4 ? hm : 4); ++k) printf("  [bf16 0x%04x: ref 0x%04x got 0x%04x]", hf[3 * k], hf[3 * k + 1], hf[3 * k + 2]);
        printf("\n");
      }
  return 0;
}
