// Diagnostic probe: which (lane, byte) of the scale VGPRs feeds which (row/col, K block), and the K map of the operand bytes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OA, int OB>
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint32_t* sa, const uint32_t* sb, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = reinterpret_cast<const int*>(A + l * 32)[q]; b[q] = reinterpret_cast<const int*>(B + l * 32)[q]; }   // raw: lane l's 32 bytes
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
  for (int e = 0; e < 4; ++e) D[l * 4 + e] = c[e];      // raw: lane l register e
}

static uint8_t *dA, *dB; static uint32_t *dsa, *dsb; static float* dD;
template <int OA, int OB>
static std::vector<float> run(const std::vector<uint8_t>& A, const std::vector<uint8_t>& B, const std::vector<uint32_t>& sa, const std::vector<uint32_t>& sb) {
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((probe<OA, OB>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  return D;
}

int main() {
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 1024);
  const uint8_t ONE = 0x38, TWO = 0x40;
  std::vector<uint8_t> A(2048, ONE), B(2048, ONE);
  std::vector<uint32_t> s1(64, 0x7f7f7f7fu);
  // 0. all ones, unit scales: every D element must be 128
  { auto D = run<0, 0>(A, B, s1, s1); printf("all-ones: D[0]=%g D[255]=%g\n", D[0], D[255]); }
  // 1. D layout: A lane La byte 0 = 2.0 -> which D elements change (by +1)?  B lane Lb byte 0 = 2.0 likewise.
  for (int side = 0; side < 2; ++side)
    for (int L : {0, 1, 5, 16, 17, 37, 63}) {
      auto A2 = A, B2 = B;
      (side ? B2 : A2)[L * 32] = TWO;
      auto D = run<0, 0>(A2, B2, s1, s1);
      printf("%s lane %2d byte0=2: changed (lane,reg):", side ? "B" : "A", L);
      int n = 0;
      for (int i = 0; i < 256; ++i) if (D[i] != 128.f) { if (n < 6) printf(" (%d,%d:%g)", i / 4, i % 4, D[i]); ++n; }
      printf("  [%d changed]\n", n);
    }
  // 2. K pairing: A lane La byte ja = 2, B lane Lb byte jb = 2: element where both meet gets +3 (2*2-1) instead of +1+1 when same k
  {
    printf("K pairing (A lane 0 byte j vs B lane 0/16/32/48 byte j'): D(lane0,reg0) expected 131 when k matches, 130 otherwise\n");
    for (int ja : {0, 5, 16, 31})
      for (int Lb : {0, 16, 32, 48})
        for (int jb : {0, 5, 16, 31}) {
          auto A2 = A, B2 = B;
          A2[0 * 32 + ja] = TWO; B2[Lb * 32 + jb] = TWO;
          auto D = run<0, 0>(A2, B2, s1, s1);
          if (D[0] == 131.f) printf("  A(l0,j%d) pairs with B(l%d,j%d)\n", ja, Lb, jb);
        }
    for (int La : {16, 32, 48})
      for (int Lb : {0, 16, 32, 48}) {
        auto A2 = A, B2 = B;
        A2[La * 32] = TWO; B2[Lb * 32] = TWO;
        auto D = run<0, 0>(A2, B2, s1, s1);
        if (D[0] == 131.f) printf("  A(l%d,j0) pairs with B(l%d,j0)\n", La, Lb);
      }
  }
  // 3. scales: sa lane L byte b = 128 (x2): which D change and by how much (a K block of 32 ones doubled: +32)
  auto scale_scan = [&](auto runf, int opsel, int side) {
    for (int L : {0, 1, 16, 17, 32, 48, 63})
      for (int b = 0; b < 4; ++b) {
        auto s2 = s1;
        s2[L] = (s2[L] & ~(0xffu << (8 * b))) | (0x80u << (8 * b));
        auto D = side ? runf(A, B, s1, s2) : runf(A, B, s2, s1);
        int n = 0; float val = 0; int first = -1;
        for (int i = 0; i < 256; ++i) if (D[i] != 128.f) { if (first < 0) { first = i; val = D[i]; } ++n; }
        if (n) printf("  %s opsel=%d lane %2d byte %d -> %d changed, first (lane %d reg %d) = %g\n", side ? "sb" : "sa", opsel, L, b, n, first / 4, first % 4, val);
      }
  };
  printf("scale scan opsel 0:\n"); scale_scan([&](auto& a, auto& b, auto& x, auto& y) { return run<0, 0>(a, b, x, y); }, 0, 0);
  scale_scan([&](auto& a, auto& b, auto& x, auto& y) { return run<0, 0>(a, b, x, y); }, 0, 1);
  printf("scale scan opsel_a 1 / opsel_b 2:\n"); scale_scan([&](auto& a, auto& b, auto& x, auto& y) { return run<1, 2>(a, b, x, y); }, 1, 0);
  scale_scan([&](auto& a, auto& b, auto& x, auto& y) { return run<1, 2>(a, b, x, y); }, 2, 1);
  printf("scale scan opsel_a 3:\n"); scale_scan([&](auto& a, auto& b, auto& x, auto& y) { return run<3, 0>(a, b, x, y); }, 3, 0);
  // 4. which K block does lane L's scale cover: A nonzero only in lane group g's bytes, sa lane L doubled
  printf("scale x block (A = 1 only in lanes of group ga, zero elsewhere; sa lane L byte0 doubled, opsel 0): D(lane0,reg0)\n");
  for (int ga = 0; ga < 4; ++ga) {
    std::vector<uint8_t> A3(2048, 0);
    for (int l = 16 * ga; l < 16 * ga + 16; ++l) for (int j = 0; j < 32; ++j) A3[l * 32 + j] = ONE;
    for (int L : {0, 16, 32, 48}) {
      auto s2 = s1; s2[L] = 0x7f7f7f80u;
      auto D = run<0, 0>(A3, B, s2, s1);
      printf("  ga=%d L=%2d: %g\n", ga, L, D[0]);
    }
  }
  return 0;
}
