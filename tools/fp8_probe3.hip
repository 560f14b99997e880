// Diagnostic probe for the fp8 weight-gradient kernel (round 4): (1) what ds_read_b64_tr_b8 returns -- for lane l reading the 8 bytes
// at its address, which (source lane, source byte) ends up in (lane, byte); (2) the register maps of v_mfma_scale_f32_32x32x64_f8f6f4:
// A / B bytes -> (row / column, k), D registers -> (row, column), scale bytes -> (row, K block).
// hipcc --offload-arch=gfx950 -O2 tools/fp8_probe3.hip -o tools/fp8_probe3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void tr8(const uint8_t* img, int* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = img[i];
  __syncthreads();
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + threadIdx.x * 8));
  out[threadIdx.x * 2] = v[0]; out[threadIdx.x * 2 + 1] = v[1];
}

template <int OA, int OB>
__global__ void mf(const uint8_t* A, const uint8_t* B, const uint32_t* sa, const uint32_t* sb, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = reinterpret_cast<const int*>(A + l * 32)[q]; b[q] = reinterpret_cast<const int*>(B + l * 32)[q]; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OA, (int)sa[l], OB, (int)sb[l]);
  for (int e = 0; e < 16; ++e) D[l * 16 + e] = c[e];
}

static uint8_t *dA, *dB; static uint32_t *dsa, *dsb; static float* dD;
template <int OA, int OB>
static std::vector<float> run(const std::vector<uint8_t>& A, const std::vector<uint8_t>& B, const std::vector<uint32_t>& sa, const std::vector<uint32_t>& sb) {
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((mf<OA, OB>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  std::vector<float> D(1024);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  return D;
}

int main() {
  // ---- (1) transposing byte read
  {
    uint8_t* dimg; int* dout;
    hipMalloc(&dimg, 512); hipMalloc(&dout, 512);
    std::vector<uint8_t> lo(512), hi(512);
    for (int i = 0; i < 512; ++i) { lo[i] = (uint8_t)(i & 255); hi[i] = (uint8_t)(i >> 8); }
    std::vector<uint8_t> rlo(512), rhi(512);
    hipMemcpy(dimg, lo.data(), 512, hipMemcpyHostToDevice); hipLaunchKernelGGL(tr8, dim3(1), dim3(64), 0, 0, dimg, dout); hipMemcpy(rlo.data(), dout, 512, hipMemcpyDeviceToHost);
    hipMemcpy(dimg, hi.data(), 512, hipMemcpyHostToDevice); hipLaunchKernelGGL(tr8, dim3(1), dim3(64), 0, 0, dimg, dout); hipMemcpy(rhi.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b8, lane l reads address 8*l: (lane, byte) <- (source lane, source byte)\n");
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 8; ++j) { const int pos = rlo[l * 8 + j] | (rhi[l * 8 + j] << 8); printf(" (%2d,%d)", pos / 8, pos % 8); }
      printf("\n");
    }
  }
  // ---- (2) MFMA maps
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 4096);
  const uint8_t ONE = 0x38, TWO = 0x40;
  std::vector<uint8_t> A(2048, ONE), B(2048, ONE);
  std::vector<uint32_t> s1(64, 0x7f7f7f7fu);
  { auto D = run<0, 0>(A, B, s1, s1); printf("all-ones 32x32x64: D[0]=%g D[1023]=%g (expect 64)\n", D[0], D[1023]); }
  // rows / columns: A lane L byte 0 = 2 -> the D elements of one ROW change; B likewise one COLUMN
  for (int side = 0; side < 2; ++side)
    for (int L : {0, 1, 5, 31, 32, 33, 63}) {
      auto A2 = A, B2 = B;
      (side ? B2 : A2)[L * 32] = TWO;
      auto D = run<0, 0>(A2, B2, s1, s1);
      printf("%s lane %2d byte0=2: changed (lane,reg):", side ? "B" : "A", L);
      int n = 0;
      for (int i = 0; i < 1024; ++i) if (D[i] != 64.f) { if (n < 5) printf(" (%d,%d)", i / 16, i % 16); ++n; }
      printf("  [%d changed]\n", n);
    }
  // D map: set A row r (all bytes of lanes r, r+32) = 2 and B column c = 2 -> the element (r, c) becomes 4*64 = 256
  printf("D(row r, col c) lives in (lane, reg):\n");
  for (int r : {0, 1, 3, 4, 7, 8, 15, 16, 31})
    for (int c : {0, 5}) {
      auto A2 = A, B2 = B;
      for (int j = 0; j < 32; ++j) { A2[r * 32 + j] = TWO; A2[(r + 32) * 32 + j] = TWO; B2[c * 32 + j] = TWO; B2[(c + 32) * 32 + j] = TWO; }
      auto D = run<0, 0>(A2, B2, s1, s1);
      for (int i = 0; i < 1024; ++i) if (D[i] == 256.f) printf("  (r %2d, c %2d) -> lane %2d reg %2d\n", r, c, i / 16, i % 16);
    }
  // K map: A(lane 0, byte ja) = 2 and B(lane Lb, byte jb) = 2: D(0,0) = 64 + 3 when the k indices coincide, 64 + 2 otherwise
  printf("K pairing: A(lane La, byte ja) meets B(lane Lb, byte jb) at the same k:\n");
  for (int La : {0, 32})
    for (int ja : {0, 1, 15, 16, 17, 31})
      for (int Lb : {0, 32})
        for (int jb = 0; jb < 32; ++jb) {
          auto A2 = A, B2 = B;
          A2[La * 32 + ja] = TWO; B2[Lb * 32 + jb] = TWO;
          auto D = run<0, 0>(A2, B2, s1, s1);
          if (D[0] == 67.f) printf("  A(l%d,j%d) <-> B(l%d,j%d)\n", La, ja, Lb, jb);
        }
  // which K block (of 32) do the bytes of lane 0 / lane 32 belong to: zero A except one byte; scale of one lane doubled
  printf("scale map (opsel 0): A = 1 only at (lane La, byte ja); sa lane L byte 0 doubled -> D(0,0) doubles when that scale covers it\n");
  for (int La : {0, 32})
    for (int ja : {0, 15, 16, 31}) {
      std::vector<uint8_t> A3(2048, 0);
      A3[La * 32 + ja] = ONE;
      printf("  A(l%d,j%d):", La, ja);
      for (int L : {0, 32}) {
        for (int b = 0; b < 4; ++b) {
          auto s2 = s1; s2[L] = (s2[L] & ~(0xffu << (8 * b))) | (0x80u << (8 * b));
          auto D0 = run<0, 0>(A3, B, s2, s1);
          if (D0[0] == 2.f) printf(" opsel0: sa lane %d byte %d;", L, b);
          auto D1 = run<1, 0>(A3, B, s2, s1);
          if (D1[0] == 2.f) printf(" opsel1: sa lane %d byte %d;", L, b);
        }
      }
      printf("\n");
    }
  printf("scale rows (opsel 0): sa lane L byte 0 doubled -> rows of D that change\n");
  for (int L : {0, 1, 31, 32, 33}) {
    auto s2 = s1; s2[L] = 0x7f7f7f80u;
    auto D = run<0, 0>(A, B, s2, s1);
    int n = 0; printf("  sa lane %2d:", L);
    for (int i = 0; i < 1024; ++i) if (D[i] != 64.f) { if (n < 3) printf(" (lane %d reg %d = %g)", i / 16, i % 16, D[i]); ++n; }
    printf(" [%d changed]\n", n);
  }
  return 0;
}
