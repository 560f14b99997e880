// Row-tiled gather-GEMM kernels: the general lowering (any channel count / tap list) and its fast path.
#include "gg_common.h"

template <typename T, int BP, int BC, int WP, int WC>
__global__ __launch_bounds__(256, 2) void gg_kernel(const GGArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int NPW = BP / WP;
  constexpr int FP = WP / 16, FC = WC / 16;
  constexpr int PR = BP / 32;
  constexpr int CR = (BC + 31) / 32;
  constexpr int ROWS = BP + BC;
  static_assert((BP / WP) * (BC / WC) == 4, "4 waves per workgroup");
  __shared__ uint4 smem[2 * ROWS * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct, tile_p = tile / a.nct;
  const int p0 = tile_p * BP, c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w);

  int sy0[PR], sx0[PR], img[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    int m = p0 + r0 + 32 * i;
    if (m < a.M) {
      int gx = m % a.Wg, t = m / a.Wg;
      int gy = t % a.Hg;
      img[i] = t / a.Hg;
      sy0[i] = gy * a.sy_mul;
      sx0[i] = gx * a.sx_mul;
    } else {
      img[i] = -1; sy0[i] = 0; sx0[i] = 0;
    }
  }
  int tap = cc / a.cch, c8 = cc % a.cch;

  uint4 ra[PR], rb[CR];

  auto gload = [&]() {
    const bool kok = tap < a.ntaps;
    unsigned code = 0;
    if (kok) code = tap < 8 ? (unsigned)((a.tap_lo >> (8 * tap)) & 0xffull) : (a.tap_hi & 0xffu);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1, ws = (int)(code >> 4);
    int q = 0, cq = c8;
    if (a.src_ps) { q = c8 / a.cps_src_chunks; cq = c8 - q * a.cps_src_chunks; }
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int sy = sy0[i] + dy, sx = sx0[i] + dx;
      const bool ok = kok && img[i] >= 0 && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      long long off;
      if (!a.src_ps) off = ((long long)(img[i] * a.Hs + sy) * a.Ws + sx) * a.ldx + c8 * EPC;
      else off = (((long long)(img[i] * 2 * a.Hs + 2 * sy + (q >> 1))) * (2 * a.Ws) + 2 * sx + (q & 1)) * a.ldx + cq * EPC;
      // always-valid address + value select: keeps the staging registers out of scratch
      uint4 v = *reinterpret_cast<const uint4*>(X + (ok ? off : 0ll));
      ra[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int row = r0 + 32 * i, n = c0 + (WC == 64 ? perm64(row) : row);
      const bool ok = kok && row < BC && n < a.Nout;
      const long long woff = (long long)n * a.ldw + (long long)ws * a.Cred + c8 * EPC;
      uint4 v = *reinterpret_cast<const uint4*>(Wt + (ok ? woff : 0ll));
      rb[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    }
    c8 += 8;
    while (c8 >= a.cch) { c8 -= a.cch; ++tap; }
  };
  auto lstore = [&](int buf) {
    uint4* s = smem + buf * ROWS * 8;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int row = r0 + 32 * i;
      s[row * 8 + (cc ^ ((row >> 1) & 7))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int row = r0 + 32 * i;
      if (row < BC) s[(BP + row) * 8 + (cc ^ ((row >> 1) & 7))] = rb[i];
    }
  };

  f32x4_t acc[FC][FP];
#pragma unroll
  for (int j = 0; j < FC; ++j)
#pragma unroll
    for (int i = 0; i < FP; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int wp = wave % NPW, wc = wave / NPW;
  const int l15 = lane & 15, g = lane >> 4;
  const int nk = (a.kchunks + 7) >> 3;

  gload();
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) gload();
    const uint4* s = smem + cur * ROWS * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 4 + g;
      uint4 fa[FC], fb[FP];
#pragma unroll
      for (int j = 0; j < FC; ++j) {
        const int row = wc * WC + 16 * j + l15;
        fa[j] = s[(BP + row) * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int i = 0; i < FP; ++i) {
        const int row = wp * WP + 16 * i + l15;
        fb[i] = s[row * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < FC; ++j)
#pragma unroll
        for (int i = 0; i < FP; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  gg_epilogue<T, BP, BC, WP, WC>(a, acc, p0, c0, wp, wc, l15, g);
}

// ---------------------------------------------------------------------------------------------
// Fast path: every K-step (8 chunks) lies inside ONE tap (chunks-per-tap % 8 == 0, true for every
// wide layer: Cred >= 64 bf16 / 32 fp32).  Then tap, channel offset and the source-pixel shift are
// workgroup-uniform per K-step and live in SGPRs; each thread keeps constant 32-bit byte offsets for
// its rows and a 9-bit tap-validity mask.  Operands are fetched with raw buffer loads whose
// descriptor base is re-pointed per K-step (scalar adds); a padded / out-of-tile row simply gets an
// out-of-range offset and the hardware returns zeros - no per-row address arithmetic in the loop.

template <typename T, int BP, int BC, int WP, int WC>
__global__ __launch_bounds__(256, 2) void gg_fast_kernel(const GGArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int NPW = BP / WP;
  constexpr int FP = WP / 16, FC = WC / 16;
  constexpr int PR = BP / 32;
  constexpr int CR = (BC + 31) / 32;
  constexpr int ROWS = BP + BC;
  static_assert((BP / WP) * (BC / WC) == 4, "4 waves per workgroup");
  __shared__ uint4 smem[2 * ROWS * 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct, tile_p = tile / a.nct;
  const int p0 = tile_p * BP, c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;
  const int Wsrc = a.src_ps ? 2 * a.Ws : a.Ws, Hsrc = a.src_ps ? 2 * a.Hs : a.Hs, psm = a.src_ps ? 2 : 1;

  // workgroup-uniform base pixel: first row of the tile minus a margin that covers every tap shift
  long long pbase;
  {
    const int gx = p0 % a.Wg, t = p0 / a.Wg;
    const int gy = t % a.Hg, im = t / a.Hg;
    pbase = ((long long)im * Hsrc + (long long)gy * a.sy_mul * psm) * Wsrc + (long long)gx * a.sx_mul * psm - (2 * Wsrc + 2);
    if (pbase < 0) pbase = 0;
  }
  unsigned rowoff[PR], vmask[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    const int m = p0 + r0 + 32 * i;
    rowoff[i] = 0; vmask[i] = 0;
    if (m < a.M) {
      const int gx = m % a.Wg, t = m / a.Wg;
      const int gy = t % a.Hg, im = t / a.Hg;
      const int sy0 = gy * a.sy_mul, sx0 = gx * a.sx_mul;
      const long long p = ((long long)im * Hsrc + (long long)sy0 * psm) * Wsrc + (long long)sx0 * psm;
      rowoff[i] = (unsigned)((p - pbase) * a.ldx * ES) + cc * 16;
      for (int t2 = 0; t2 < a.ntaps; ++t2) {
        const unsigned code = t2 < 8 ? (unsigned)((a.tap_lo >> (8 * t2)) & 0xffull) : (a.tap_hi & 0xffu);
        const int sy = sy0 + (int)(code & 3u) - 1, sx = sx0 + (int)((code >> 2) & 3u) - 1;
        if ((unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws) vmask[i] |= 1u << t2;
      }
    }
  }
  unsigned woff[CR];
#pragma unroll
  for (int i = 0; i < CR; ++i) {
    const int row = r0 + 32 * i, prow = WC == 64 ? perm64(row) : row;     // LDS row `row` holds output channel c0 + prow
    woff[i] = (row < BC && c0 + prow < a.Nout) ? (unsigned)((long long)prow * a.ldw * ES) + cc * 16 : DG_OOB_OFF;
  }
  const char* Xb = reinterpret_cast<const char*>(a.x) + pbase * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;

  u32x4_t ra[PR], rb[CR];
  int tap = 0, cbase = 0;          // workgroup-uniform K position (SGPRs)
  auto gload = [&]() {
    const unsigned code = tap < 8 ? (unsigned)((a.tap_lo >> (8 * tap)) & 0xffull) : (a.tap_hi & 0xffu);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1, ws = (int)(code >> 4);
    long long xo;
    if (!a.src_ps) xo = ((long long)dy * a.Ws + dx) * a.ldx + cbase * EPC;
    else {
      const int q = cbase / a.cps_src_chunks, cq = cbase - q * a.cps_src_chunks;
      xo = ((long long)(2 * dy + (q >> 1)) * Wsrc + (2 * dx + (q & 1))) * a.ldx + cq * EPC;
    }
    const long long wo = (long long)ws * a.Cred + cbase * EPC;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + xo * ES), 0, (int)DG_OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(Wb + wo * ES), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const unsigned vo = ((vmask[i] >> tap) & 1u) ? rowoff[i] : DG_OOB_OFF;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[i], 0, 0);
    cbase += 8;
    if (cbase >= a.cch) { cbase = 0; ++tap; }
  };
  auto lstore = [&](int buf) {
    uint4* s = smem + buf * ROWS * 8;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      const int row = r0 + 32 * i;
      s[row * 8 + (cc ^ ((row >> 1) & 7))] = __builtin_bit_cast(uint4, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < CR; ++i) {
      const int row = r0 + 32 * i;
      if (row < BC) s[(BP + row) * 8 + (cc ^ ((row >> 1) & 7))] = __builtin_bit_cast(uint4, rb[i]);
    }
  };

  f32x4_t acc[FC][FP];
#pragma unroll
  for (int j = 0; j < FC; ++j)
#pragma unroll
    for (int i = 0; i < FP; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int wp = wave % NPW, wc = wave / NPW;
  const int l15 = lane & 15, g = lane >> 4;
  const int nk = a.kchunks >> 3;

  // write-after-barrier pipeline: the registers filled during step ks-1 are stored at the START of step ks and
  // refilled at once with the loads of step ks+2, so a load has a whole step to arrive
  gload();
  lstore(0);
  if (nk > 1) gload();
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) lstore(cur ^ 1);
    if (ks + 2 < nk) gload();
    const uint4* s = smem + cur * ROWS * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk * 4 + g;
      uint4 fa[FC], fb[FP];
#pragma unroll
      for (int j = 0; j < FC; ++j) {
        const int row = wc * WC + 16 * j + l15;
        fa[j] = s[(BP + row) * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int i = 0; i < FP; ++i) {
        const int row = wp * WP + 16 * i + l15;
        fb[i] = s[row * 8 + (ch ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int j = 0; j < FC; ++j)
#pragma unroll
        for (int i = 0; i < FP; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
    }
    __syncthreads();
    cur ^= 1;
  }
  gg_epilogue<T, BP, BC, WP, WC>(a, acc, p0, c0, wp, wc, l15, g);
}

template <typename T, int BP, int BC, int WP, int WC>
static int gg_launch_t(GGArgs& a, hipStream_t st) {
  a.nct = (unsigned)((a.Nout + BC - 1) / BC);
  const unsigned npt = (unsigned)((a.M + BP - 1) / BP);
  a.nwg = a.nct * npt;
  const bool fast_ok = a.cch % 8 == 0 && (!a.src_ps || a.cps_src_chunks % 8 == 0);
  g_last_kinds |= fast_ok ? 2 : 1;
  if (fast_ok)
    hipLaunchKernelGGL((gg_fast_kernel<T, BP, BC, WP, WC>), dim3(a.nwg), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gg_kernel<T, BP, BC, WP, WC>), dim3(a.nwg), dim3(256), 0, st, a);
  return dg_check_launch();
}

int gg_launch_rows(GGArgs& a, int dtype, hipStream_t st) {
  if (dtype == DG_F32) {
    if (a.Nout > 64) return gg_launch_t<float, 128, 128, 64, 64>(a, st);
    if (a.Nout > 32) return gg_launch_t<float, 128, 64, 64, 32>(a, st);
    if (a.Nout > 16) return gg_launch_t<float, 128, 32, 32, 32>(a, st);
    return gg_launch_t<float, 128, 16, 32, 16>(a, st);
  }
  if (a.Nout > 64) return gg_launch_t<bf16_t, 128, 128, 64, 64>(a, st);
  if (a.Nout > 32) return gg_launch_t<bf16_t, 128, 64, 64, 32>(a, st);
  if (a.Nout > 16) return gg_launch_t<bf16_t, 128, 32, 32, 32>(a, st);
  return gg_launch_t<bf16_t, 128, 16, 32, 16>(a, st);
}
