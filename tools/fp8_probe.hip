// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: operand lane maps and the scale-byte selection, with exact data.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_probe.hip -o tools/fp8_probe && tools/fp8_probe
// Register map checked here with random data (found with tools/fp8_probe2.hip; all lines must print OK -- csrc/quant.hip and
// gg_halo4w_f8_kernel rely on it):
//   A / B: lane l (row or column l & 15, group g = l >> 4) holds 32 operand bytes; byte j is K index
//          64 * (j / 16) + 16 * g + (j % 16)
//   D: lane l register e holds D[row = 4 * (l >> 4) + e][col = l & 15]
//   scale: the E8M0 scale (2^(s - 127)) of K block kb = K / 32 of row r is byte `opsel` of the scale VGPR of lane r + 16 * kb
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OA, int OB>
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint32_t* sa, const uint32_t* sb, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int q = 0; q < 8; ++q) {
    a[q] = reinterpret_cast<const int*>(A + ((l & 15) * 128 + 32 * (l >> 4)))[q];      // A stored [row][k]
    b[q] = reinterpret_cast<const int*>(B + ((l & 15) * 128 + 32 * (l >> 4)))[q];      // B stored [col][k]
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
  for (int e = 0; e < 4; ++e) D[(4 * (l >> 4) + e) * 16 + (l & 15)] = c[e];
}

static float e4m3(uint8_t v) {   // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
  if (e == 15 && m == 7) x = NAN;
  return s ? -x : x;
}

template <int OA, int OB>
static int run(const char* what) {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  std::vector<uint32_t> sa(64), sb(64);
  uint32_t seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (auto& v : A) { v = (uint8_t)(rnd() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; }
  for (auto& v : B) { v = (uint8_t)(rnd() & 0xff); if ((v & 0x7f) == 0x7f) v = 0x38; }
  for (int l = 0; l < 64; ++l) {   // four different bytes per lane: only byte `opsel` may matter
    sa[l] = (uint32_t)(120 + rnd() % 14) | (uint32_t)(120 + rnd() % 14) << 8 | (uint32_t)(120 + rnd() % 14) << 16 | (uint32_t)(120 + rnd() % 14) << 24;
    sb[l] = (uint32_t)(120 + rnd() % 14) | (uint32_t)(120 + rnd() % 14) << 8 | (uint32_t)(120 + rnd() % 14) << 16 | (uint32_t)(120 + rnd() % 14) << 24;
  }
  uint8_t *dA, *dB; uint32_t *dsa, *dsb; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((probe<OA, OB>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double ref = 0, mag = 0;
      for (int g = 0; g < 4; ++g)          // the kernel gave lane (r, g) the bytes [32g, 32g + 32) of row r
        for (int j = 0; j < 32; ++j) {
          const int kb = (64 * (j / 16) + 16 * g + (j % 16)) / 32;
          const double fa = std::ldexp(1.0, (int)((sa[r + 16 * kb] >> (8 * OA)) & 0xff) - 127);
          const double fb = std::ldexp(1.0, (int)((sb[c + 16 * kb] >> (8 * OB)) & 0xff) - 127);
          const double term = (double)e4m3(A[r * 128 + 32 * g + j]) * (double)e4m3(B[c * 128 + 32 * g + j]) * fa * fb;
          ref += term; mag += std::fabs(term);
        }
      const double err = std::fabs(D[r * 16 + c] - ref) / (mag + 1e-30);     // relative to the sum of |products|: cancellation does not count
      if (err > worst) worst = err;
    }
  printf("%s opsel_a=%d opsel_b=%d: max |D - ref| / sum|a b| %.3e  %s\n", what, OA, OB, worst, worst < 1e-3 ? "OK (register map)" : "MISMATCH");
  // (a wrong register / scale map gives errors of order 1..1e3; what remains with the right one, ~1e-4 of sum|a b| on random
  // bytes that span all 17 binades of E4M3, is the instruction's own accumulation: products are not summed in full fp32)
  return worst < 1e-3 ? 0 : 1;
}

int main() {
  int bad = 0;
  bad += run<0, 0>("mfma_scale_16x16x128 e4m3");
  bad += run<1, 2>("mfma_scale_16x16x128 e4m3");
  bad += run<3, 0>("mfma_scale_16x16x128 e4m3");
  return bad;
}
