// The halo-patch conv kernel (forward of both strides, every data gradient of the wide layers): the dominant kernel of the step.
#include "gg_common.h"

// ---------------------------------------------------------------------------------------------
// Halo path (unit-stride gathers: stride-1 forward, every data gradient; stride-2 forward through parity planes).  One
// workgroup = a 16x16 tile of the GEMM-row grid of ONE image x 128 output channels.  The 9 taps of a 3x3 stencil read
// overlapping source pixels, so instead of staging a [pixels][K-slice] operand per tap (9x the bytes) the workgroup keeps
// the (16+2)x(16+2) source PATCH of the current channel block in LDS and every tap reads its fragments from the patch at a
// shifted row; only the weights stream per tap-step.
#ifdef DG_STAMP
// diagnostic build only (make stamp): per-wave cycle sums of the segments of a tap-step, blocks 0/1
__device__ unsigned long long g_stamps[2 * 8 * 12];
extern "C" int dg_debug_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 1;
}
// ... and the life of every workgroup of the last launch (up to DG_WGLOG_MAX): {HW_ID | XCC_ID << 32, start, end} of wave 0, to
// reconstruct per-CU residency (tools/stamp_probe.py --residency)
#define DG_WGLOG_MAX 16384
__device__ unsigned long long g_wglog[DG_WGLOG_MAX * 3];
extern "C" int dg_debug_wglog(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wglog), sizeof(unsigned long long) * 3 * (n < DG_WGLOG_MAX ? n : DG_WGLOG_MAX)) == hipSuccess ? 0 : 1;
}
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(v) do { } while (0)
#endif
// S2 = stride-2 forward on the same machinery.  out(y,x) = sum_{r,s} in(2y+r-1, 2x+s-1) w(r,s): split the input into its four
// parity planes (py,px) = (row & 1, col & 1) of 2x2 blocks; tap r reads plane py = (r != 1) at block offset (r == 0 ? -1 : 0),
// so every tap is a UNIT-stride shift in block coordinates.  The K loop runs over (plane, channel block) pairs; the patch
// of a pair is the 17x17 blocks of that plane (gathered with stride 2 straight from the NHWC tensor), and
// the pair's taps are the 1 / 2 / 2 / 4 taps that read the plane (a.tap_* are grouped by plane by gg_regroup_taps_by_plane).
bool gg_regroup_taps_by_plane(GGArgs& a) {
  if (a.ntaps != 9) return false;
  unsigned codes[9], out[9];
  bool seen[3][3] = {};
  for (int t = 0; t < 9; ++t) {
    codes[t] = t < 8 ? (unsigned)((a.tap_lo >> (8 * t)) & 0xffull) : (a.tap_hi & 0xffu);
    const int dy = (int)(codes[t] & 3u) - 1, dx = (int)((codes[t] >> 2) & 3u) - 1;
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || seen[dy + 1][dx + 1]) return false;
    seen[dy + 1][dx + 1] = true;
  }
  int n = 0;
  for (int plane = 0; plane < 4; ++plane)
    for (int t = 0; t < 9; ++t) {
      const int dy = (int)(codes[t] & 3u) - 1, dx = (int)((codes[t] >> 2) & 3u) - 1;
      if ((dy != 0) * 2 + (dx != 0) != plane) continue;
      const unsigned by = dy == -1 ? 0u : 1u, bx = dx == -1 ? 0u : 1u;      // block offset + 1
      out[n++] = by | (bx << 2) | (codes[t] & 0xf0u);
    }
  a.tap_lo = 0; a.tap_hi = 0;
  for (int t = 0; t < 9; ++t) {
    if (t < 8) a.tap_lo |= (unsigned long long)out[t] << (8 * t); else a.tap_hi = out[t];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Four-wave halo kernel, TWO workgroups per CU.  Tile: 16x16 pixels x 128 output channels, 64-channel K-steps, so patch (324 x 144 B) + two 16-KB weight slots = 78 KB and a second, independent workgroup shares
// the CU: its tap-step loop runs while this one sits in its prologue (exposed patch latency) or in its store-bound
// epilogue (then ~14k cycles per tile; 7-11k since the 16-byte, flag-specialised epilogue), which is where 9..18-step tiles
// lose 30-50 % of their time; and
// the two waves of a SIMD now belong to different workgroups (no shared barrier, no lock-step).  Each wave owns 4 tile
// rows x all 128 channels (8 x 4 accumulator fragments, 3 LDS fragment reads per 8 MFMAs instead of 4).
// One barrier per step, at its top:  BARRIER | DMA W[s+2] -> slot s&1 | mma(k0) | read k0 of s+1 | mma(k1) | read k1 of s+1.
// S2: stride-2 forward over the four parity planes of the input (above): K loop over (plane, channel block)
// pairs with 1 / 2 / 2 / 4 taps, the plane's patch gathered with stride 2.
// PS: pixel-shuffled source (data gradient of an up-sampling conv): virtual pixel (y, x), channel quarter q = stored pixel
// (2y + (q >> 1), 2x + (q & 1)); a 64-channel block lies inside one quarter, so its patch is a stride-2 gather like S2's.
// NW = 8: the same kernel with EIGHT waves and a 16x16-pixel x 256-channel tile (waves 0-3 the first 128 channels, waves 4-7
// the second), one workgroup per CU.  Both channel halves read ONE patch, so the patch bytes per flop halve -- which is what
// bounds the stride-2 forward (a stride-2 tile reads 4x the input pixels of a stride-1 tile: 227 flop per patch byte at 128
// channels, and the per-CU global->LDS path sustains only ~10-12 B/clk) -- at the price of the second, independent workgroup.
// SEG: the data gradient of a stride-2 layer as ONE launch.  Its four output-parity classes (1 / 2 / 2 / 4 taps, conv_plan.hip) all
// read dy and write the four interleaved pixel sets of dx.  As four launches each re-reads dy from HBM (features.2 at batch 32:
// 2.1 GB, far beyond the 256 MB Infinity Cache; 17.7 GB moved for 11.3 GB algorithmic).  Here a workgroup still computes ONE class of
// one tile, but the launch covers (image, class, tile) in that order and workgroups keep their launch order (no XCD-contiguous
// remap): the chip works through one image's four classes back to back, so classes 2-4 find that image's dy (67 MB) in the
// Infinity Cache, and the three inter-launch tails go away.  a.tap_* hold the classes' taps concatenated (9 in all).
template <typename T, bool S2, bool PS = false, int NW = 4, bool SEG = false>
__global__ __launch_bounds__(64 * NW, 2) void gg_halo4w_kernel(const GGArgs a, int tiles_x, int tiles_y) {
  static_assert(!SEG || (!S2 && !PS && NW == 4), "SEG is a mode of the plain four-wave kernel");
  constexpr int NT = 64 * NW, RPP = NT / 8;                              // threads; patch rows staged per pass
  constexpr int EPC = DT<T>::EPC;
  constexpr int ES = (int)sizeof(T);
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int BC = 32 * NW, KC = 8;                                    // 8 chunks per row: 64 bf16 / 32 fp32 channels
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B patch rows
  constexpr int WROW = KC * 16;                                          // 128 B weight rows, chunk c of row r at c ^ ((r >> 1) & 7)
  constexpr int NPL = (PROWS * KC + NT - 1) / NT;                        // 11 (6) patch chunks per thread
  constexpr int NWL = BC * KC / NT;                                      // 4 weight pieces per wave and step
  extern __shared__ __attribute__((aligned(16))) char dsm4w[];
  char* const s_patch = dsm4w;                    // [PROWS][PITCH]
  char* const s_w = dsm4w + PROWS * PITCH;        // [2][BC][WROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tile = SEG ? blockIdx.x : xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct;
  unsigned rest = tile / a.nct;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  // SEG order: a.seg == 1: (image, class, tile row, tile column); a.seg == 2: (image, tile row, class, tile column) -- the four
  // classes of one row of tiles back to back, so store-bound 1-tap and MFMA-bound 4-tap workgroups share the chip at any time
  int cls = 0;
  if (SEG && a.seg == 2) { cls = (int)(rest & 3u); rest >>= 2; }
  const int ty0 = (rest % tiles_y) * TH; rest /= tiles_y;
  if (SEG && a.seg != 2) { cls = (int)(rest & 3u); rest >>= 2; }    // parity class (py, px) = (cls >> 1, cls & 1)
  const int img = (int)rest;
  const int ntaps_l = SEG ? (0x4221 >> (4 * cls)) & 15 : a.ntaps;        // this workgroup's taps: [tap0_l, tap0_l + ntaps_l) of a.tap_*
  const int tap0_l = SEG ? (0x5310 >> (4 * cls)) & 15 : 0;
  const int c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0, RPP)
  const int wq = wave & 3, wh = wave >> 2;        // tile rows 4*wq.., channel half wh (0 unless NW = 8)
  const int sy_base = S2 ? (ty0 > 0 ? 2 * (ty0 - 1) : 0) : (ty0 - 1 > 0 ? ty0 - 1 : 0);
  const char* Xb = PS ? reinterpret_cast<const char*>(a.x) + ((long long)img * 2 * a.Hs + 2 * sy_base) * (2 * a.Ws) * a.ldx * ES
                      : reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncbr = a.cch / KC;                    // real channel blocks
  const int ncb = S2 ? 4 * ncbr : ncbr;           // (plane, channel block) pairs, plane-major
  const int nsteps = ncbr * ntaps_l;
  auto plane_of = [&](int vcb) { return S2 ? (int)(vcb >= ncbr) + (int)(vcb >= 2 * ncbr) + (int)(vcb >= 3 * ncbr) : 0; };
  auto ntaps_of = [&](int vcb) { return S2 ? (0x4221 >> (4 * plane_of(vcb))) & 15 : ntaps_l; };

  unsigned woff[NWL];                             // DMA piece i of this wave: rows wave*32 + 8i .. +7, lane = (row, physical chunk)
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    int row = wave * 32 + i * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    row = perm64(row);                              // LDS row holds output channel c0 + perm64(row)
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;
    woff[i] = (unsigned)((long long)row * a.ldw * ES) + logical * 16;
  }
  auto tap_code = [&](int vcb, int tap) {
    const int gt = S2 ? ((0x5310 >> (4 * plane_of(vcb))) & 15) + tap : tap0_l + tap;
    return gt < 8 ? (unsigned)((a.tap_lo >> (8 * gt)) & 0xffull) : (a.tap_hi & 0xffu);
  };
  // Patch addressing.  Thread (r0, cc) stages chunk cc of patch pixels r0 + RPP*i, i < NPL; pixel (py, px) of the patch lies at
  // O + py*rstep + px*cstep from the block's base, and RPP = K1*PW + K2, so chunk i sits at off0 + i*A + w_i*Bd with
  // w_i = (px0 + i*K2) / PW wraps of the column.  off0, the w_i (4 bits each) and the "outside the image / patch" bits are
  // computed ONCE; a patch load then costs five VALU operations per chunk instead of re-deriving (py, px), the bounds and the
  // offset (~15, on a staging path that is issue-bound: -3...-9 % per launch).  The parity plane (S2) / channel quarter (PS) of a
  // block only moves the uniform base.
  constexpr int K1 = RPP / PW, K2 = RPP % PW;
  const int p_rstep = (S2 ? 2 : PS ? 4 : 1) * a.Ws * a.ldx * ES, p_cstep = ((S2 || PS) ? 2 : 1) * a.ldx * ES;
  const int p_A = K1 * p_rstep + K2 * p_cstep, p_Bd = p_rstep - PW * p_cstep;
  unsigned p_off0, p_wlo = 0, p_whb = 0;           // p_whb: w_8.. in bits 0-15, invalid bits of chunk i at bit 16 + i
  {
    const int py0 = r0 / PW, px0 = r0 - py0 * PW;
    const int o_rows = S2 ? 2 * (ty0 - 1) - sy_base : PS ? 4 * (ty0 - 1 - sy_base) : ty0 - 1 - sy_base;      // in rows of Ws pixels
    const int o_cols = (S2 || PS) ? 2 * (tx0 - 1) : tx0 - 1;
    p_off0 = (unsigned)((o_rows * a.Ws + o_cols) * a.ldx * ES + py0 * p_rstep + px0 * p_cstep) + cc * 16;
    const unsigned hlim = S2 ? (unsigned)a.Hs >> 1 : (unsigned)a.Hs, wlim = S2 ? (unsigned)a.Ws >> 1 : (unsigned)a.Ws;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int w = (px0 + i * K2) / PW;
      const int py = py0 + i * K1 + w, px = px0 + i * K2 - w * PW;
      const bool ok = r0 + RPP * i < PROWS && (unsigned)(ty0 - 1 + py) < hlim && (unsigned)(tx0 - 1 + px) < wlim;
      if (i < 8) p_wlo |= (unsigned)w << (4 * i); else p_whb |= (unsigned)w << (4 * (i - 8));
      if (!ok) p_whb |= 1u << (16 + i);
    }
  }
  asm volatile("" : "+v"(p_off0), "+v"(p_wlo), "+v"(p_whb));
  u32x4_t rp[NPL];
  auto load_patch = [&](int vcb) {
    const int plane = plane_of(vcb), cb = vcb - plane * ncbr;
    const int ppy = plane >> 1, ppx = plane & 1;
    const int psq = PS ? (cb * KC) / a.cps_src_chunks : 0;                 // channel quarter of this block
    const int psy = psq >> 1, psx = psq & 1;
    long long cboff = PS ? (long long)(cb * KC - psq * a.cps_src_chunks) * EPC * ES : (long long)cb * KC * EPC * ES;
    if (S2) cboff += (long long)(ppy * a.Ws + ppx) * a.ldx * ES;
    if (PS) cboff += (long long)(psy * 2 * a.Ws + psx) * a.ldx * ES;
    // the descriptor base is workgroup-uniform, but derived from the (plane, channel block) counter the compiler does not prove
    // uniform: without the readfirstlane pair every one of the NPL loads below sits in its own waterfall loop (S2 / PS instances)
    const unsigned long long xbase = (unsigned long long)(Xb + cboff);
    const unsigned long long xuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xbase);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xuni, 0, (int)DG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int w = (int)__builtin_amdgcn_ubfe(i < 8 ? p_wlo : p_whb, i < 8 ? 4 * i : 4 * (i - 8), 4);
      const unsigned bad = (unsigned)__builtin_amdgcn_sbfe((int)p_whb, 16 + i, 1);          // all ones when the chunk is outside
      const unsigned off = ((unsigned)__mul24(w, p_Bd) + p_off0 + (unsigned)(i * p_A)) | bad;  // >= DG_OOB_OFF: the load returns 0
      rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  char* const st_base = s_patch + r0 * PITCH + cc * 16;
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + RPP * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(st_base + i * RPP * PITCH) = __builtin_bit_cast(uint4, rp[i]);
    }
  };
  typedef int i32x4h_t __attribute__((ext_vector_type(4)));
  i32x4h_t w_rs;
  int w_dst0 = 0;
  auto dma_setup = [&](int vcb, int tap, int slot) {
    const unsigned code = tap_code(vcb, tap);
    const long long wo = (long long)(code >> 4) * a.Cred + (vcb - plane_of(vcb) * ncbr) * KC * EPC;
    const unsigned long long wbase = (unsigned long long)(Wb + wo * ES);
    w_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wbase);
    w_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32) & 0xffff);
    w_rs[2] = (int)DG_OOB_OFF;
    w_rs[3] = 0x00020000;
    w_dst0 = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_w + slot * (BC * WROW) + (wave * 32) * WROW)));
  };
  auto dma_piece = [&](int i) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(w_dst0 + i * 8 * WROW), "v"(woff[i]), "s"(w_rs) : "memory");
  };
  auto dma_w = [&](int cb, int tap, int slot) {
    dma_setup(cb, tap, slot);
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma_piece(i);
  };
  auto barrier_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };
  auto barrier_keep_patch = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPL) : "memory"); __syncthreads(); };

  constexpr bool PREB = sizeof(T) == 2;      // (the fp32 parity instances keep the bias in the epilogue: 28 -> 192 B of scratch otherwise)
  // accumulators start at the bias (acc[4h + q][.][e] = channel c0 + wh*128 + 64h + 16g + 4q + e, the epilogue's perm64 order):
  // its loads ride behind the first patch instead of costing every tile's epilogue a global round trip per 64-channel half
  f32x4_t acc[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4_t b = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int cbq = c0 + (tid >> 8) * 128 + (j >> 2) * 64 + 16 * ((tid & 63) >> 4) + 4 * (j & 3);
    if (PREB && a.bias && cbq < a.Nout) { const float4 b4 = *reinterpret_cast<const float4*>(a.bias + cbq); b = f32x4_t{b4.x, b4.y, b4.z, b4.w}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = b;
  }

  const char* fa_k[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fa_k[kk] = s_w + (wh * 128 + l15) * WROW + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) * 16);
  const char* const fb_lane = s_patch + l15 * PITCH + g * 16;
  auto read_frags = [&](uint4 (&fa)[8], uint4 (&fb)[4], int pa, const char* pb, int kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = *reinterpret_cast<const uint4*>(fa_k[kk] + pa + j * 16 * WROW);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + kk * 64);
  };
  auto mma_rows = [&](const uint4 (&fa)[8], const uint4 (&fb)[4], int j0) {     // two weight fragments x four pixel rows
#pragma unroll
    for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) Mma<T>::run(fa[j], fb[i], acc[j][i]);
  };
  auto adv = [&](int& c_, int& t_) { if (++t_ == ntaps_of(c_)) { t_ = 0; ++c_; } };
  auto patch_ptr = [&](int vcb_, int tap_) -> const char* {
    const unsigned code = tap_code(vcb_, tap_);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
    return fb_lane + ((wq * 4 + 1 + dy) * PW + 1 + dx) * PITCH;
  };

#ifdef DG_STAMP
  unsigned long long tK0 = 0, tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sAB = 0, sBC = 0, sCD = 0, sDE = 0, tL0 = 0;
  STAMP(tK0);
#endif
  load_patch(0);
  int cb = 0, tap = 0, cbw = 0, tapw = 0;
  dma_w(0, 0, 0);
  adv(cbw, tapw);
  if (nsteps > 1) dma_w(cbw, tapw, 1);
  adv(cbw, tapw);                    // -> W[2]
  store_patch();
  barrier_all();
  // ONE fragment set (accumulators 128 + fragments 48 + patch staging 44 registers): a wave waits for its LDS reads in
  // the open, which is what the second workgroup on the CU is there to cover.
  uint4 fa[8], fb[4];
  const char* pb = patch_ptr(0, 0);
  int pa = 0;
  read_frags(fa, fb, pa, pb, 0);
  STAMP(tL0);
  for (int s = 0; s < nsteps; ++s) {
    STAMP(tA);
    const bool more = s + 1 < nsteps;
    int ntap = tap + 1, ncbn = cb;
    const int ntaps_cb = ntaps_of(cb);
    if (ntap == ntaps_cb) { ntap = 0; ncbn = cb + 1; }
    const bool swap = ntap == 0 && more;
    const bool patch_now = cb + 1 < ncb && (ntaps_cb == 1 || tap == 0);     // fetch the next block's patch during its predecessor's first step
    const bool fetch = s + 2 < nsteps;
    // k-block 0 (fragments read at the end of the previous step), then k-block 1: after it every wave has read all it
    // needs of this step, so the barrier below frees slot s&1 (and, at a block end, the patch); W[s+1] has landed by then
    // k-block 0, and the weight fragments of k-block 1 re-read row pair by row pair right behind the MFMAs that consumed their
    // k-block-0 contents (that latency hides under the remaining rows; only the four pixel fragments are read in the open at
    // the end): +0.5-1.5 % per launch against reading all twelve fragments after the block (the chip gives about half of a
    // cycle saving back as clock).  The asm fences keep each read in its slot and each MFMA pair in front of it.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mma_rows(fa, fb, 2 * q);
      asm volatile("" : "+v"(acc[2 * q][0]), "+v"(acc[2 * q][1]), "+v"(acc[2 * q][2]), "+v"(acc[2 * q][3]),
                        "+v"(acc[2 * q + 1][0]), "+v"(acc[2 * q + 1][1]), "+v"(acc[2 * q + 1][2]), "+v"(acc[2 * q + 1][3]) :: "memory");
      __builtin_amdgcn_sched_barrier(0);
      fa[2 * q] = *reinterpret_cast<const uint4*>(fa_k[1] + pa + (2 * q) * 16 * WROW);
      fa[2 * q + 1] = *reinterpret_cast<const uint4*>(fa_k[1] + pa + (2 * q + 1) * 16 * WROW);
      asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const uint4*>(pb + i * PW * PITCH + 64);
    __builtin_amdgcn_sched_barrier(0);
    if (patch_now) load_patch(cb + 1);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(tB);
#pragma unroll
    for (int q = 0; q < 4; ++q) mma_rows(fa, fb, 2 * q);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(tC);
    if (patch_now && !swap) barrier_keep_patch(); else barrier_all();
    STAMP(tD);
    pa = ((s + 1) & 1) * (BC * WROW);
    pb = patch_ptr(ncbn < ncb ? ncbn : 0, ntap);
    if (swap) {                      // channel-block boundary: the single-buffered patch is rewritten, then published
      store_patch();
      barrier_all();
    }
    // next step's first fragments go out before the DMA pieces: the ~700 cycles a wave spends issuing those then cover
    // the LDS read latency instead of preceding it
    if (more) read_frags(fa, fb, pa, pb, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (fetch) { dma_setup(cbw, tapw, s & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_piece(q);
    }
    adv(cbw, tapw);
    __builtin_amdgcn_sched_barrier(0);
    tap = ntap; cb = ncbn;
    STAMP(tE);
#ifdef DG_STAMP
    sAB += tB - tA; sBC += tC - tB; sCD += tD - tC; sDE += tE - tD;
#endif
  }
#ifdef DG_STAMP
  unsigned long long tL1, tX;
  STAMP(tL1);
#endif
#ifdef DG_STAMP
  unsigned long long est[4] = {0, 0, 0, 0};
  halo_epilogue<T, 2, PREB>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g, SEG ? cls >> 1 : -1, SEG ? cls & 1 : -1, est);
#else
  if constexpr (SEG) halo_epilogue<T, 2, PREB>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g, cls >> 1, cls & 1);
  else halo_epilogue<T, 2, PREB>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g);
#endif
#ifdef DG_STAMP
  STAMP(tX);
  if (blockIdx.x < DG_WGLOG_MAX && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = g_wglog + (unsigned long long)blockIdx.x * 3;
    o[0] = (unsigned long long)hw | ((unsigned long long)xcc << 32); o[1] = tK0; o[2] = tX;
  }
  // two workgroups from the MIDDLE of the launch (steady state: the first round starts in lock-step and runs ~25 % faster per step)
  const unsigned sb = blockIdx.x - a.nwg / 2;
  if (sb < 2u && lane == 0) {
    unsigned long long* o = g_stamps + (sb * 8 + wave) * 12;
    o[0] = sAB; o[1] = sBC; o[2] = sCD; o[3] = sDE; o[4] = (unsigned long long)nsteps; o[5] = tL1 - tL0; o[6] = tX - tL1; o[7] = tL0 - tK0;
    o[8] = est[0] - tL1; o[9] = est[1] - est[0]; o[10] = est[2] - est[1]; o[11] = est[3] - est[2];
  }
#endif
}

template <typename T, bool S2, bool PS = false, int NW = 4, bool SEG = false>
static int gg_launch_halo4w(GGArgs& a, int N, hipStream_t st) {
  constexpr int BC = 32 * NW;
  constexpr int LDS_BYTES = 324 * 144 + 2 * BC * 128;
  DG_SET_MAX_LDS_ONCE((&gg_halo4w_kernel<T, S2, PS, NW, SEG>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = (unsigned)((a.Nout + BC - 1) / BC);
  a.nwg = a.nct * (unsigned)(tiles_x * tiles_y * N) * (SEG ? 4u : 1u);
  g_last_kinds |= 8;
#ifdef DG_STAMP
  static const int abl = getenv("DG_ABL") ? atoi(getenv("DG_ABL")) : 0;
  a.dbg = abl;
  if (abl & 2) {        // one workgroup per CU: no partner on the CU's vector-memory pipe / matrix pipe
    hipFuncSetAttribute((const void*)&gg_halo4w_kernel<T, S2, PS, NW, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((gg_halo4w_kernel<T, S2, PS, NW, SEG>), dim3(a.nwg), dim3(64 * NW), 100 * 1024, st, a, tiles_x, tiles_y);
    return dg_check_launch();
  }
#endif
  hipLaunchKernelGGL((gg_halo4w_kernel<T, S2, PS, NW, SEG>), dim3(a.nwg), dim3(64 * NW), LDS_BYTES, st, a, tiles_x, tiles_y);
  return dg_check_launch();
}

int gg_launch_halo(GGArgs& a, int dtype, int N, bool s2, bool ps, int nw, hipStream_t st) {
  // the kernel forms patch offsets with 24-bit multiplies of the row step (bytes between patch rows); conv_plan.hip routes
  // oversized rows to the row-tiled kernel BEFORE it regroups taps / merges classes, so this is a guard, not a fallback
  if (!gg_halo_row_step_ok(a.Ws, a.ldx, dtype, ps ? 4 : s2 ? 2 : 1)) return DG_ERR_BAD_SHAPE;
  if (a.seg) return dtype == DG_BF16 ? gg_launch_halo4w<bf16_t, false, false, 4, true>(a, N, st) : gg_launch_halo4w<float, false, false, 4, true>(a, N, st);
  if (dtype == DG_F32) {
    if (ps) return gg_launch_halo4w<float, false, true>(a, N, st);
    return s2 ? gg_launch_halo4w<float, true>(a, N, st) : gg_launch_halo4w<float, false>(a, N, st);
  }
  if (ps) return gg_launch_halo4w<bf16_t, false, true>(a, N, st);
  if (s2) return nw == 8 ? gg_launch_halo4w<bf16_t, true, false, 8>(a, N, st) : gg_launch_halo4w<bf16_t, true>(a, N, st);
  return gg_launch_halo4w<bf16_t, false>(a, N, st);
}
