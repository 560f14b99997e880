// HBM-bound edge layers: <= 16 output channels (gg_halo16_kernel) and <= 2 real input channels (the im2col kernels).
#include "gg_common.h"

typedef int i32x4_t __attribute__((ext_vector_type(4)));
// ---------------------------------------------------------------------------------------------
// Halo kernel for layers with <= 16 output channels (generator conv3.2: 128 -> 2 at 1024^2; the critic's first-layer data
// gradient in the penalty): HBM-bound, 1/8 of the MFMA work of a 128-wide tile.  The per-tap kernel re-reads every input
// pixel 9x through L2 (4.9 ms per pass against a 1.7 ms HBM floor); here a 16x16-pixel tile keeps the (16+2)^2 patch of one
// 64-channel block in LDS for all taps, like the wide halo kernel, and ALL nine 16x64 weight tiles of the block beside it
// (67 KB in total, so two workgroups share a CU and one's loads overlap the other's MFMAs: no software pipeline needed).
// 4 waves, wave = 4 tile rows x 16 channels; per block and tap two k-chunks of (1 weight + 4 patch fragment reads, 4 MFMAs).
template <typename T>
__global__ __launch_bounds__(256, 2) void gg_halo16_kernel(const GGArgs a, int tiles_x, int tiles_y) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int KC = 8;                                                  // 16-byte chunks per block row (64 bf16 / 32 fp32 channels)
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B: conflict-free fragment reads at any row offset
  constexpr int NPL = (PROWS * KC + 255) / 256;                          // 11 patch chunks per thread
  constexpr int WROWS = 9 * 16;
  constexpr int NWL = (WROWS * KC + 255) / 256;                          // 5 weight chunks per thread
  extern __shared__ __attribute__((aligned(16))) char dsm16[];
  char* const s_patch = dsm16;                    // [PROWS][PITCH]
  char* const s_w = dsm16 + PROWS * PITCH;        // [9 taps][16 channels][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned tile = xcd_remap(blockIdx.x, a.nwg);
  unsigned rest = tile;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  const int ty0 = (rest % tiles_y) * TH;
  const int img = rest / tiles_y;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0,32)
  const int sy_base = ty0 - 1 > 0 ? ty0 - 1 : 0;
  const char* Xb = reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w);
  const int l15 = lane & 15, g = lane >> 4;
  const int ncb = a.cch / KC, ntaps = a.ntaps;

  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // the next block's patch and weights are fetched into registers while the current block computes (the kernel is
  // HBM-latency-bound: with the loads issued at the top of their own block every block waited a full memory round trip)
  u32x4_t rp[NPL], rw[NWL];
  auto load_block = [&](int cb) {
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(Xb + (long long)cb * KC * EPC * ES), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + 32 * i;
      const int py = pr / PW, px = pr - py * PW;
      const int sy = ty0 - 1 + py, sx = tx0 - 1 + px;
      const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
      const unsigned off = ok ? (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16 : DG_OOB_OFF;
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
    __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc((void*)(Wb + (long long)cb * KC * EPC * ES), 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int wr = r0 + 32 * i;                 // row = tap * 16 + channel
      const int t = wr >> 4, n = wr & 15;
      unsigned off = DG_OOB_OFF;
      if (wr < WROWS && t < ntaps && n < a.Nout) {
        const unsigned code = t < 8 ? (unsigned)((a.tap_lo >> (8 * t)) & 0xffull) : (a.tap_hi & 0xffu);
        off = (unsigned)(((long long)n * a.ldw + (long long)(code >> 4) * a.Cred) * ES) + cc * 16;
      }
      rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rwd, off, 0, 0);
    }
  };
  load_block(0);
  for (int cb = 0; cb < ncb; ++cb) {
    __syncthreads();                                // everybody is done with the previous block's tiles
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + 32 * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(s_patch + pr * PITCH + cc * 16) = __builtin_bit_cast(uint4, rp[i]);
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int wr = r0 + 32 * i;
      if (wr < WROWS) *reinterpret_cast<uint4*>(s_w + wr * PITCH + cc * 16) = __builtin_bit_cast(uint4, rw[i]);
    }
    __syncthreads();
    if (cb + 1 < ncb) load_block(cb + 1);
    for (int t = 0; t < ntaps; ++t) {
      const unsigned code = t < 8 ? (unsigned)((a.tap_lo >> (8 * t)) & 0xffull) : (a.tap_hi & 0xffu);
      const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
      const char* pb = s_patch + ((wave * 4 + 1 + dy) * PW + 1 + dx + l15) * PITCH + g * 16;
      const char* pa = s_w + (t * 16 + l15) * PITCH + g * 16;
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        const uint4 fa = *reinterpret_cast<const uint4*>(pa + kk * 64);
        uint4 fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + kk * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(fa, fb[i], acc[i]);
      }
    }
  }

  // epilogue: lane = pixel (tile row wave*4 + i, column l15), channels 4g .. 4g+3
  typedef EpiIO<T> IO;
  const int psm = a.dy_mul, psx = a.dx_mul;
  const long long pbase = ((long long)img * a.Hd + (long long)ty0 * psm + a.dy_off) * a.Wd + (long long)tx0 * psx + a.dx_off;
  const int cj = 4 * g;
  const bool cok = cj < a.Nout && tx0 + l15 < a.Wg;
  const float4 bias = (a.bias && cj < a.Nout) ? *reinterpret_cast<const float4*>(a.bias + cj) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto rsrc = [&](const void* p, long long ld) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + pbase * ld * ES), 0, (int)DG_OOB_OFF, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rY = rsrc(a.y, a.ldy);
  const __amdgpu_buffer_rsrc_t r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1), r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2),
                               rm = rsrc(a.mask ? a.mask : a.y, a.ldmask);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool ok = cok && ty0 + wave * 4 + i < a.Hg;
    const int pix = (wave * 4 + i) * psm * a.Wd + l15 * psx;
    const unsigned oy = ok ? (unsigned)((pix * (int)a.ldy + cj) * ES) : DG_OOB_OFF;
    float v[4] = {acc[i][0] + bias.x, acc[i][1] + bias.y, acc[i][2] + bias.z, acc[i][3] + bias.w};
    float r[4];
    if (a.has_act) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = leaky(v[e], a.act_slope);
    }
    if (a.r1) {
      IO::unpack(IO::load(r1, ok ? (unsigned)((pix * (int)a.ldr1 + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * a.s1 + r[e];
    }
    if (a.r2) {
      IO::unpack(IO::load(r2, ok ? (unsigned)((pix * (int)a.ldr2 + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * a.s2 + r[e];
    }
    const bool mask_on = a.mask && a.mask_c0 == 0;          // (<= 16 output channels: mask_c0 is 0 or excludes them all)
    if (mask_on && !a.mask_last) {
      IO::unpack(IO::load(rm, ok ? (unsigned)((pix * (int)a.ldmask + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(r[e], a.mask_slope);
    }
    if (a.accumulate) {
      IO::unpack(IO::load(rY, oy), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    if (mask_on && a.mask_last) {
      IO::unpack(IO::load(rm, ok ? (unsigned)((pix * (int)a.ldmask + cj) * ES) : DG_OOB_OFF), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= leaky_grad(r[e], a.mask_slope);
    }
    IO::store(v, rY, oy);
  }
}

template <typename T>
static int gg_launch_halo16_t(GGArgs& a, int N, hipStream_t st) {
  constexpr int LDS_BYTES = (324 + 9 * 16) * 144;
  DG_SET_MAX_LDS_ONCE((&gg_halo16_kernel<T>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = 1;
  a.nwg = (unsigned)(tiles_x * tiles_y * N);
  g_last_kinds |= 8;
  hipLaunchKernelGGL((gg_halo16_kernel<T>), dim3(a.nwg), dim3(256), LDS_BYTES, st, a, tiles_x, tiles_y);
  return dg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// im2col path for stride-1 FORWARD layers with <= 2 real input channels (critic features.0 on the
// 1024^2 tiles, generator conv1 with 2 covariates).  K = 9 taps x 2 channels = 18, so
// the layer is bound by writing its output to HBM, not by MFMA: the 18 (padded to 32) K values of every
// pixel are gathered straight from the 2 real channels (4/8-byte loads) instead of walking 9 taps x 16
// padded channels, the weight tile is built once per workgroup, and each workgroup streams several
// 128-pixel tiles.
template <typename T, bool LEAN>
__global__ __launch_bounds__(256, LEAN ? 3 : 2) void gg_im2col_kernel(const GGArgs a, int tiles_per_block) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int KCH = 32 / EPC;         // 16-B chunks per 32-element K row
  constexpr int TPC = EPC / 2;          // taps per chunk (2 channels per tap)
  constexpr int NCH = 128 * KCH / 256;  // chunks per thread per tile
  __shared__ uint4 sW[128 * KCH], sX[128 * KCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.y * 128;
  const T* X = reinterpret_cast<const T*>(a.x);
  const T* Wt = reinterpret_cast<const T*>(a.w);
  auto swz = [](int row) { return KCH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  auto load_pair = [](const T* p, unsigned* w2) {
    if constexpr (sizeof(T) == 2) { w2[0] = *reinterpret_cast<const unsigned*>(p); }
    else { const uint2 v = *reinterpret_cast<const uint2*>(p); w2[0] = v.x; w2[1] = v.y; }
  };
  // weight tile: row = output channel, K element k = tap*2 + ci
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int e = tid + 256 * i, row = e / KCH, col = e % KCH;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    const int n = c0 + perm64(row);                 // 64-channel wave tiles: permuted channel order
    if (n < a.Nout) {
#pragma unroll
      for (int tt = 0; tt < TPC; ++tt) {
        const int tp = col * TPC + tt;
        if (tp < 9) load_pair(Wt + ((long long)n * 9 + tp) * a.Cred, w + tt * (4 / TPC));
      }
    }
    sW[row * KCH + (col ^ swz(row))] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  const int wp = wave & 1, wc = wave >> 1;
  const int l15 = lane & 15, g = lane >> 4;
  constexpr int ES = (int)sizeof(T);
  // Everything below is branch-free per tile (buffer loads / stores with out-of-range offsets for padding, tile ends and
  // channel tails): the compiler can then count the outstanding operations and the LDS write of the next tile's gathered
  // rows waits for the gather loads only (s_waitcnt vmcnt(#stores issued after them)), not for this tile's stores to
  // reach memory -- with per-load branches it fell back to vmcnt(0) at the loop head, i.e. one store round trip per tile.
  // forward, stride 1, plain destination: GEMM row m is both the source and the destination pixel index
  const int col = tid % KCH, row0 = tid / KCH;                   // this thread's K chunk; rows row0 + (256 / KCH) * i
  int tap_rel[TPC];                                              // source pixel shift of the chunk's taps
  int tap_dy[TPC], tap_dx[TPC];
#pragma unroll
  for (int tt = 0; tt < TPC; ++tt) {
    const int tp = col * TPC + tt;
    tap_dy[tt] = tp < 9 ? tp / 3 - 1 : 4;                        // 4: never inside the image
    tap_dx[tt] = tp < 9 ? tp % 3 - 1 : 0;
    tap_rel[tt] = tap_dy[tt] * a.Ws + tap_dx[tt];
  }
  const int cb16 = c0 + wc * 64 + 16 * g;
  const bool cok = cb16 < a.Nout;
  float bias[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b4 = (a.bias && cok) ? *reinterpret_cast<const float4*>(a.bias + cb16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias[4 * q] = b4.x; bias[4 * q + 1] = b4.y; bias[4 * q + 2] = b4.z; bias[4 * q + 3] = b4.w;
  }
  const int ldb = (a.Nout >> 6) * 4, bidx = ((c0 + wc * 64) >> 6) * 4 + g;
  // the im2col rows of the NEXT tile are gathered into registers while the current tile's MFMAs and (long) store epilogue
  // run: the 4/8-byte gathers are latency-bound and nothing else would hide them (one barrier pair per tile)
  unsigned gw[NCH][4];
  auto gather = [&](int p0, bool live) {
    long long pbase = (long long)p0 - a.Ws - 1;
    if (pbase < 0) pbase = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.x) + pbase * a.ldx * ES), 0,
                                                                      (int)DG_OOB_OFF, 0x00020000);
    const int mrel = (int)(p0 - pbase);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = row0 + (256 / KCH) * i;
      const unsigned m = (unsigned)(p0 + row);
      const unsigned tq = m / (unsigned)a.Wg;
      const int gx = (int)(m - tq * (unsigned)a.Wg), gy = (int)(tq % (unsigned)a.Hg);
      const bool mok = live & (m < (unsigned)a.M);
#pragma unroll
      for (int tt = 0; tt < TPC; ++tt) {
        const bool ok = (int)mok & (int)((unsigned)(gy + tap_dy[tt]) < (unsigned)a.Hs) & (int)((unsigned)(gx + tap_dx[tt]) < (unsigned)a.Ws);
        const unsigned off = ok ? (unsigned)((mrel + row + tap_rel[tt]) * (int)a.ldx * ES) : DG_OOB_OFF;
        if constexpr (sizeof(T) == 2) gw[i][tt] = __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0);
        else {
          const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
          gw[i][2 * tt] = v[0]; gw[i][2 * tt + 1] = v[1];
        }
      }
    }
  };
  const int tile0 = blockIdx.x * tiles_per_block;
  auto publish = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = row0 + (256 / KCH) * i;
      sX[row * KCH + (col ^ swz(row))] = make_uint4(gw[i][0], gw[i][1], gw[i][2], gw[i][3]);
    }
    __syncthreads();
  };
  gather(tile0 * 128, tile0 * 128 < a.M);
  publish();
  // per tile: mask words | gather of tile t+1 | MFMAs | stores | barrier | gathered rows -> LDS | barrier.  The LDS write
  // sits at the END of the body so that its wait is "all but the 12 stores issued after the gather" on every path
  for (int t = 0; t < tiles_per_block; ++t) {
    const int p0 = (tile0 + t) * 128;
    if (p0 >= a.M) break;
    // operand resources of this tile (destination pixel = GEMM row, everything based at the tile's first pixel) and its
    // mask words, ahead of the next tile's gather: the epilogue then waits for these loads only
    auto rsrc = [&](const void* p, long long ld, int es) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + (long long)p0 * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
    };
    EpiRes R;
    R.rY = rsrc(a.y, a.ldy, ES);
    R.r1 = rsrc(a.r1 ? a.r1 : a.y, a.ldr1, ES); R.r2 = rsrc(a.r2 ? a.r2 : a.y, a.ldr2, ES); R.rm = rsrc(a.mask ? a.mask : a.y, a.ldmask, ES);
    R.rbi = rsrc(a.mask_bits ? a.mask_bits : a.y, ldb, 2); R.rbo = rsrc(a.out_bits ? a.out_bits : a.y, ldb, 2);
    R.rq = rsrc(a.out_q ? a.out_q : a.y, a.ldy, 1); R.rqs = rsrc(a.out_qs ? a.out_qs : a.y, a.ldqs, 1);
    R.ldy = (int)a.ldy; R.ld1 = (int)a.ldr1; R.ld2 = (int)a.ldr2; R.ldm = (int)a.ldmask;
    unsigned mbv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rel = wp * 64 + 16 * i + l15;
      mbv[i] = epi64_bits<LEAN>(a, R, (cok && p0 + rel < a.M) ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF);
    }
    gather(p0 + 128, t + 1 < tiles_per_block);
    f32x4_t acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KCH / 4; ++kk) {
      const int ch = kk * 4 + g;
      uint4 fa[4], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int row = wc * 64 + 16 * j + l15; fa[j] = sW[row * KCH + (ch ^ swz(row))]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = wp * 64 + 16 * i + l15; fb[i] = sX[row * KCH + (ch ^ swz(row))]; }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rel = wp * 64 + 16 * i + l15;
      const bool ok = cok && p0 + rel < a.M;
      epi64_pixel<T, LEAN>(a, R, acc[0][i], acc[1][i], acc[2][i], acc[3][i], bias,
                           ok ? (unsigned)((rel * R.ldy + cb16) * ES) : DG_OOB_OFF, ok ? (unsigned)((rel * R.ld1 + cb16) * ES) : DG_OOB_OFF,
                           ok ? (unsigned)((rel * R.ld2 + cb16) * ES) : DG_OOB_OFF, ok ? (unsigned)((rel * R.ldm + cb16) * ES) : DG_OOB_OFF,
                           ok ? (unsigned)((rel * ldb + bidx) * 2) : DG_OOB_OFF, mbv[i], a.mask && cb16 >= a.mask_c0);
    }
    __syncthreads();
    publish();
  }
}

// ---------------------------------------------------------------------------------------------
// The same layer without LDS and without barriers (bf16, image width a multiple of 16, 128-channel tiles, bias / activation /
// bit-mask / MXFP8-copy epilogues).  gg_im2col_kernel moves every tile through gather -> LDS -> barrier -> MFMA -> stores ->
// barrier with 12 waves per CU; its 2.7-3.4 TB/s of stores is per-tile latency, not bandwidth.  Here a WAVE owns a group of 16
// consecutive pixels of one image row and all 128 channels: the MFMA's B fragment of pixel n is K elements 8g..8g+7 = taps
// 4g..4g+3 x 2 channels, i.e. four dwords of x (lane groups 0-1: taps 0-7, group 2: tap 8, group 3: nothing), the eight A fragments (18 x 128 weights) live in registers for the whole kernel, and the BIAS rides in the two
// spare K slots 18/19 of lane group 2 (weight = bias split into a bf16 high and low part, pixel value = 1.0 twice: exact to 2^-17
// of the bias), so the epilogue is activation + rounding + stores.  Nothing is shared between waves: 16 waves per CU, each with
// the next group's pixels in flight behind the current group's eight 16-byte stores per lane.  Waves sweep the pixel
// groups interleaved (group = iteration * waves + wave), so the chip writes one moving window of the output.
template <int F>
__global__ __launch_bounds__(256, 4) void gg_im2col_direct_kernel(const GGArgs a, int ngroups, bool tiled) {
  typedef bf16_t T;
  constexpr int ES = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int c0 = blockIdx.y * 128;
  const T* Wt = reinterpret_cast<const T*>(a.w);
  uint4 fa[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = c0 + (j >> 2) * 64 + perm64((j & 3) * 16 + l15);      // channel of A row l15 of fragment j (Nout % 128 == 0)
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (g < 2) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) w[tt] = *reinterpret_cast<const unsigned*>(Wt + ((long long)n * 9 + 4 * g + tt) * a.Cred);
    } else if (g == 2) {
      w[0] = *reinterpret_cast<const unsigned*>(Wt + ((long long)n * 9 + 8) * a.Cred);
      const float b = a.bias ? a.bias[n] : 0.f;
      const bf16_t hi = f32_to_bf16(b), lo = f32_to_bf16(b - bf16_to_f32(hi));
      w[1] = (unsigned)hi | ((unsigned)lo << 16);
    }
    fa[j] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // The 16 pixels of a group read 3 rows x 18 columns of x = 54 dwords (2 channels each): ONE load per lane brings them in
  // (lane i < 54: row i / 18 - 1, column i % 18 - 1 relative to the group's first pixel; out-of-image -> 0) and four
  // ds_bpermute_b32 hand every lane its taps (lane constants; K slots past the 9 taps point at lane 63, which always holds 0).
  // Four gathers per lane straight from x touched ~48 cache lines per group in the 16-channel-padded layout (15 now).
  const int ld_r = lane / 18 - 1, ld_c = lane % 18 - 1;
  int perm_src[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int tp = 4 * g + tt;
    perm_src[tt] = tp < 9 ? ((tp / 3) * 18 + l15 + tp % 3) * 4 : 63 * 4;
  }
  const unsigned ones = g == 2 ? 0x3f803f80u : 0u;                 // K slots 18 / 19 of the pixel operand: bf16 1.0 twice
  const int ldb = (a.Nout >> 6) * 4;
  unsigned offy[2], boff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    offy[h] = (unsigned)((l15 * (int)a.ldy + c0 + 64 * h + 16 * g) * ES);
    boff[h] = (unsigned)((l15 * ldb + ((c0 + 64 * h) >> 6) * 4 + g) * 2);
  }
  // out_bits: a pixel's record is 8 words (128 channels); lane group g gets words g (first half) and 4 + g (second half) out of
  // the two epilogue calls.  Stored as ONE dword per lane -- group g writes words 2g, 2g+1, fetched from groups 2(g&1), 2(g&1)+1
  // by two ds_bpermute -- so a group's 256 bytes of mask words leave as one contiguous wave store instead of two scattered 2-byte
  // ones (the two 2-byte stores cost 8-14 % of the launch).
  const int ob_src = (l15 + 32 * (g & 1)) * 4, ob_sh = 16 * (g >> 1);
  const unsigned ob_off = (unsigned)((l15 * ldb + (c0 >> 6) * 4 + 2 * g) * 2);
  const float zero16[16] = {};
  float inv_u[2] = {0.f, 0.f};              // uniform-scale copy (dg_epilogue.out_u): 2^(127 - exponent of this lane's 32-channel block)
  if (F & 2048) {
#pragma unroll
    for (int h = 0; h < 2; ++h) inv_u[h] = mx_inv_scale((int)a.out_ue[(c0 + 64 * h + 16 * g) >> 5]);
  }
  auto gather = [&](int grp) -> unsigned {
    const bool live = grp >= 0;
    const unsigned m0 = live ? (unsigned)grp * 16u : 0u;
    const unsigned tq = m0 / (unsigned)a.Wg;
    const int sx = (int)(m0 - tq * (unsigned)a.Wg) + ld_c, sy = (int)(tq % (unsigned)a.Hg) + ld_r;
    long long pbase = (long long)m0 - a.Ws - 1;
    if (pbase < 0) pbase = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.x) + pbase * a.ldx * ES), 0,
                                                                      (int)DG_OOB_OFF, 0x00020000);
    const int mrel = (int)((long long)m0 - pbase) + ld_r * a.Ws + ld_c;
    const bool ok = (int)live & (int)(lane < 54) & (int)((unsigned)sy < (unsigned)a.Hs) & (int)((unsigned)sx < (unsigned)a.Ws);
    return __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (unsigned)(mrel * (int)a.ldx * ES) : DG_OOB_OFF, 0, 0);
  };
  // group order.  tiled (image height a multiple of 16): a workgroup walks 16x16-pixel tiles, wave w rows 4w..4w+3 of each (the
  // write pattern of the tiled conv kernels: 3.6 against 3.3 TB/s for the linear order, whose 4096 waves write one 16-MB window);
  // otherwise wave k of the grid takes groups k, k + waves, ...
  const int nwaves = (int)gridDim.x * 4;
  const int tiles_x = a.Wg / 16, tiles_y = a.Hg / 16, ntiles = tiled ? tiles_x * tiles_y * (a.M / (a.Hg * a.Wg)) : 0;
  auto grp_of = [&](int it) -> int {                 // < 0: past this wave's last group
    if (!tiled) { const int gq = (int)blockIdx.x * 4 + wave + it * nwaves; return gq < ngroups ? gq : -1; }
    unsigned tile = blockIdx.x + (unsigned)(it >> 2) * gridDim.x;
    if (tile >= (unsigned)ntiles) return -1;
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y, img = tile / tiles_y;
    return ((img * a.Hg + ty * 16 + 4 * wave + (it & 3)) * a.Wg + tx * 16) >> 4;
  };
  // a group's pixels are fetched two groups ahead (memory operations retire in order: a load comes back only after the stores
  // issued before it have been acknowledged)
  unsigned ab_run[2] = {0u, 0u};         // dg_epilogue.out_amax: largest block magnitude (bit pattern) this lane stored, per 64-channel half
  unsigned x0 = gather(grp_of(0)), x1 = gather(grp_of(1));
  for (int it = 0;; ++it) {
    const int grp = grp_of(it);
    if (grp < 0) break;
    const uint4 fb = make_uint4((unsigned)__builtin_amdgcn_ds_bpermute(perm_src[0], (int)x0),
                                (unsigned)__builtin_amdgcn_ds_bpermute(perm_src[1], (int)x0) | ones,
                                (unsigned)__builtin_amdgcn_ds_bpermute(perm_src[2], (int)x0),
                                (unsigned)__builtin_amdgcn_ds_bpermute(perm_src[3], (int)x0));
    const long long m0 = (long long)grp * 16;
    auto rsrc = [&](const void* p, long long ld, int es) {
      return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p) + m0 * ld * es), 0, (int)DG_OOB_OFF, 0x00020000);
    };
    EpiRes R;
    R.rY = rsrc(a.y, a.ldy, ES);
    R.r1 = R.r2 = R.rm = R.rY;
    R.rbi = (F & 2) ? rsrc(a.mask_bits, ldb, 2) : R.rY; R.rbo = (F & 4) ? rsrc(a.out_bits, ldb, 2) : R.rY;
    R.rq = (F & 256) ? rsrc(a.out_q, a.ldy, 1) : R.rY; R.rqs = (F & 256) ? rsrc(a.out_qs, a.ldqs, 1) : R.rY;
    R.ru = (F & 2048) ? rsrc(a.out_u, a.ldy, 1) : R.rY;
    R.ldy = (int)a.ldy; R.ld1 = R.ld2 = R.ldm = 0;
    unsigned mb[2] = {0u, 0u};
    if (F & 2) {                                                   // both mask words before the first store
      mb[0] = __builtin_amdgcn_raw_buffer_load_b16(R.rbi, boff[0], 0, 0);
      mb[1] = __builtin_amdgcn_raw_buffer_load_b16(R.rbi, boff[1], 0, 0);
    }
    const unsigned x2 = gather(grp_of(it + 2));
    unsigned ob[2] = {0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4_t acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        Mma<T>::run(fa[4 * h + j], fb, acc[j]);
      }
      epi64_pixel<T, true, F | 512 | ((F & 4) ? 1024 : 0)>(a, R, acc[0], acc[1], acc[2], acc[3], zero16, offy[h], 0u, 0u, 0u, boff[h], mb[h],
                                                            false, &ob[h], inv_u[h], (F & 2304) ? &ab_run[h] : nullptr);
    }
    if (F & 4) {
      const int both = (int)(ob[0] | (ob[1] << 16));
      const unsigned lo = ((unsigned)__builtin_amdgcn_ds_bpermute(ob_src, both) >> ob_sh) & 0xffffu;
      const unsigned hi = ((unsigned)__builtin_amdgcn_ds_bpermute(ob_src + 64, both) >> ob_sh) & 0xffffu;
      __builtin_amdgcn_raw_buffer_store_b32(lo | (hi << 16), R.rbo, ob_off, 0, 0);
    }
    x0 = x1; x1 = x2;
  }
  if constexpr ((F & 2304) != 0) {       // one atomic per (wave, 32-channel block): the lane pair (g, g ^ 1) already holds its block's maximum
    if (a.out_amax) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned m = ab_run[h];
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) { const unsigned o2 = (unsigned)__shfl_xor((int)m, sft, 64); m = m > o2 ? m : o2; }
        if (l15 == 0 && (g & 1) == 0) atomicMax(a.out_amax + ((c0 + 64 * h + 16 * g) >> 5), m);
      }
    }
  }
}

template <typename T>
static int gg_launch_im2col_t(GGArgs& a, hipStream_t st) {
  const int tiles = (a.M + 127) / 128;
  int tpb = tiles / 2048;            // a few tiles per workgroup so the weight tile is built rarely
  if (tpb < 1) tpb = 1;
  if (tpb > 16) tpb = 16;
  // big launches: ONE round of the resident workgroups, every workgroup the same number of tiles (16 tiles per workgroup
  // left 16384 workgroups on 768 slots: 21.3 rounds, the last a third full; +2 % at 1024^2)
  const bool lean = !a.r1 && !a.r2 && !a.mask && !a.accumulate;
  if constexpr (sizeof(T) == 2) {
    constexpr bool no_direct = false;
    const int F = (a.has_act ? 1 : 0) | (a.mask_bits ? 2 : 0) | (a.out_bits ? 4 : 0) | (a.out_q ? 256 : 0) | (a.out_u ? 2048 : 0) | (a.no_y ? 4096 : 0);
    const bool f_ok = F == 0 || F == 1 || F == 2 || F == 5 || F == 258 || F == 261 || F == 2306 || F == 2309 || F == 4354 || F == 4357 || F == 6402 || F == 6405 ||
                      F == 6146 || F == 6149;      // (the uniform-scale copy ALONE: mask bits / activation + bits, no bf16 store)
    if ((a.out_u || a.no_y || a.out_amax) && !(f_ok && !no_direct && lean && a.Wg % 16 == 0 && a.Nout % 128 == 0 && a.Hs == a.Hg && a.Ws == a.Wg)) return DG_ERR_BAD_SHAPE;
    if (!no_direct && lean && a.Wg % 16 == 0 && a.Nout % 128 == 0 && a.Hs == a.Hg && a.Ws == a.Wg &&
        (long long)a.M * a.ldy * 2 < (1ll << 46) && f_ok) {
      const int ngroups = a.M / 16;
      int nb = a.Hg % 16 == 0 ? ngroups / 16 : (ngroups + 3) / 4;   // tiles of 16 groups / workgroups of 4 groups
      if (nb > 1024) nb = 1024;                                    // 4 workgroups per CU resident, one round
      dim3 grid(nb, a.Nout / 128);
      g_last_kinds |= 16;
      const bool tiled = a.Hg % 16 == 0;
      switch (F) {
        case 0: hipLaunchKernelGGL((gg_im2col_direct_kernel<0>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 1: hipLaunchKernelGGL((gg_im2col_direct_kernel<1>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 2: hipLaunchKernelGGL((gg_im2col_direct_kernel<2>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 5: hipLaunchKernelGGL((gg_im2col_direct_kernel<5>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 258: hipLaunchKernelGGL((gg_im2col_direct_kernel<258>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 261: hipLaunchKernelGGL((gg_im2col_direct_kernel<261>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 2306: hipLaunchKernelGGL((gg_im2col_direct_kernel<2306>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 2309: hipLaunchKernelGGL((gg_im2col_direct_kernel<2309>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 4354: hipLaunchKernelGGL((gg_im2col_direct_kernel<4354>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 4357: hipLaunchKernelGGL((gg_im2col_direct_kernel<4357>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 6402: hipLaunchKernelGGL((gg_im2col_direct_kernel<6402>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 6146: hipLaunchKernelGGL((gg_im2col_direct_kernel<6146>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        case 6149: hipLaunchKernelGGL((gg_im2col_direct_kernel<6149>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
        default: hipLaunchKernelGGL((gg_im2col_direct_kernel<6405>), grid, dim3(256), 0, st, a, ngroups, tiled); break;
      }
      return dg_check_launch();
    }
  }
  static std::atomic<int> occ_cache[2] = {{0}, {0}};
  int occ = occ_cache[lean].load(std::memory_order_relaxed);
  if (!occ) {
    int n = 0;
    const hipError_t e = lean ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gg_im2col_kernel<T, true>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gg_im2col_kernel<T, false>, 256, 0);
    occ = (e == hipSuccess && n > 0) ? n : 2;
    occ_cache[lean].store(occ, std::memory_order_relaxed);
  }
  const int slots = 256 * occ;
  if (tiles >= 8 * slots) tpb = (tiles + slots - 1) / slots;
  dim3 grid((tiles + tpb - 1) / tpb, (a.Nout + 127) / 128);
  g_last_kinds |= 16;
  if (lean) hipLaunchKernelGGL((gg_im2col_kernel<T, true>), grid, dim3(256), 0, st, a, tpb);
  else hipLaunchKernelGGL((gg_im2col_kernel<T, false>), grid, dim3(256), 0, st, a, tpb);
  return dg_check_launch();
}

int gg_launch_halo16(GGArgs& a, int dtype, int N, hipStream_t st) {
  return dtype == DG_F32 ? gg_launch_halo16_t<float>(a, N, st) : gg_launch_halo16_t<bf16_t>(a, N, st);
}
int gg_launch_im2col(GGArgs& a, int dtype, hipStream_t st) {
  return dtype == DG_F32 ? gg_launch_im2col_t<float>(a, st) : gg_launch_im2col_t<bf16_t>(a, st);
}
