#include "gg_common.h"

typedef int i32x4_t __attribute__((ext_vector_type(4)));
// ---------------------------------------------------------------------------------------------
// MXFP8 version of the four-wave halo kernel (BASELINE configs[4]: the critic's wide layers, critic.py:25-88): operands are
// OCP E4M3 bytes with one E8M0 scale per block of 32 consecutive channels (csrc/quant.hip, which also records how the
// instruction maps operand bytes and scale lanes to K -- measured with tools/fp8_probe2.hip), the MFMA is
// v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulate, 2x the bf16 rate).  A K-step is still 8 chunks = 128 bytes per
// patch / weight row, now 128 channels, and the LDS images, swizzles, DMA pieces and patch pipeline are those of
// gg_halo4w_kernel: a lane's 32-byte operand is chunk g of the row's first and of its second 64-byte half -- exactly the two
// 16-byte fragments the bf16 kernel feeds to two MFMAs -- so one scaled MFMA replaces two bf16 ones at the same LDS bytes
// and the same matrix-pipe cycles for twice the channels.  Beside the operands, each K-step needs 4 scale bytes per patch
// pixel (fetched with the patch into s_ps) and per weight row (a fifth LDS-DMA piece of waves 0/1 into s_ws); lane group g
// reads byte g.  Output: bf16 through the common epilogue (activations, masks, bit masks as in the bf16 kernel).
typedef int i32x8_t __attribute__((ext_vector_type(8)));

// SEG: a stride-2 data gradient's four parity classes as one launch in (image, class | tile row, tile) order, as in gg_halo4w_kernel.
template <bool S2, int NW = 4, bool SEG = false>       // NW = 8: 256-channel tile, two channel halves on one patch (see gg_halo4w_kernel)
__global__ __launch_bounds__(64 * NW, 2) void gg_halo4w_f8_kernel(const GGArgs a, const F8Args f, int tiles_x, int tiles_y) {
  static_assert(!SEG || (!S2 && NW == 4), "SEG is a mode of the plain four-wave kernel");
  constexpr int EPC = 16, ES = 1;
  constexpr int NT = 64 * NW, RPP = NT / 8;
  constexpr int TH = 16, TW = 16, PW = TW + 2, PROWS = (TH + 2) * PW;   // 324 patch rows
  constexpr int BC = 32 * NW, KC = 8;                                    // 8 chunks per row: 128 fp8 channels
  constexpr int PITCH = KC * 16 + 16;                                    // 144 B patch rows
  constexpr int WROW = KC * 16;                                          // 128 B weight rows, chunk c of row r at c ^ ((r >> 1) & 7)
  constexpr int NPL = (PROWS * KC + NT - 1) / NT;                        // 11 (6) patch chunks per thread
  constexpr int NPS = (PROWS + NT - 1) / NT;                             // 2 (1) patch scale words per thread
  constexpr int NWL = BC * KC / NT;                                      // 4 weight pieces per wave and step
  constexpr int NSW = BC / 64;                                           // waves that carry distinct scale pieces
  extern __shared__ __attribute__((aligned(16))) char dsm4f8[];
  char* const s_patch = dsm4f8;                                // [PROWS][PITCH]
  char* const s_w = dsm4f8 + PROWS * PITCH;                    // [2][BC][WROW]
  char* const s_ws = s_w + 2 * BC * WROW;                      // [2][BC] u32: 4 scale bytes of the row's 128-channel K-step
  char* const s_ps = s_ws + 2 * BC * 4;                        // [PROWS] u32: same for the patch pixels

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tile = SEG ? blockIdx.x : xcd_remap(blockIdx.x, a.nwg);
  const int tile_c = tile % a.nct;
  unsigned rest = tile / a.nct;
  const int tx0 = (rest % tiles_x) * TW; rest /= tiles_x;
  int cls = 0;                                                   // SEG: parity class of this workgroup (order: a.seg, conv_halo.hip)
  if (SEG && a.seg == 2) { cls = (int)(rest & 3u); rest >>= 2; }
  const int ty0 = (rest % tiles_y) * TH; rest /= tiles_y;
  if (SEG && a.seg != 2) { cls = (int)(rest & 3u); rest >>= 2; }
  const int img = (int)rest;
  const int ntaps_l = SEG ? (0x4221 >> (4 * cls)) & 15 : a.ntaps;
  const int tap0_l = SEG ? (0x5310 >> (4 * cls)) & 15 : 0;
  const int c0 = tile_c * BC;
  const int cc = tid & 7, r0 = tid >> 3;          // r0 in [0, RPP)
  const int wq = wave & 3, wh = wave >> 2;        // tile rows 4*wq.., channel half wh (0 unless NW = 8)
  const int sy_base = S2 ? (ty0 > 0 ? 2 * (ty0 - 1) : 0) : (ty0 - 1 > 0 ? ty0 - 1 : 0);
  const char* Xb = reinterpret_cast<const char*>(a.x) + ((long long)img * a.Hs + sy_base) * a.Ws * a.ldx * ES;
  const int ldxs = f.ldxs, ldws = 9 * (a.Cred >> 5);
  const char* XSb = reinterpret_cast<const char*>(f.xs) + ((long long)img * a.Hs + sy_base) * a.Ws * ldxs;
  const char* Wb = reinterpret_cast<const char*>(a.w) + (long long)c0 * a.ldw * ES;
  const char* WSb = reinterpret_cast<const char*>(f.ws) + (long long)c0 * ldws;
  const int l15 = lane & 15, g = lane >> 4;
  const int ncbr = a.cch / KC;                    // real 128-channel blocks
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
  unsigned wsoff;                                 // scale piece (waves 0 and 1): LDS row wave*64 + lane
  {
    int row = perm64((wave & (NSW - 1)) * 64 + lane);
    if (c0 + row >= a.Nout) row = a.Nout - 1 - c0;
    wsoff = (unsigned)(row * ldws);
  }
  auto tap_code = [&](int vcb, int tap) {
    const int gt = S2 ? ((0x5310 >> (4 * plane_of(vcb))) & 15) + tap : tap0_l + tap;
    return gt < 8 ? (unsigned)((a.tap_lo >> (8 * gt)) & 0xffull) : (a.tap_hi & 0xffu);
  };
  // Patch addressing as in gg_halo4w_kernel (conv_halo.hip): chunk i of this thread at off0 + i*A + w_i*Bd, the wrap counts and the
  // outside-the-image bits computed once, the scale words' offsets likewise.  Only in the eight-wave instance (+2.5-4 %): the
  // four-wave ones have no registers left for the five values (they spill 8-12 and lose 5-12 %) and re-derive every offset per load.
  constexpr bool PRE = NW == 8;
  constexpr int K1 = RPP / PW, K2 = RPP % PW;
  const int p_rstep = (S2 ? 2 : 1) * a.Ws * a.ldx * ES, p_cstep = (S2 ? 2 : 1) * a.ldx * ES;
  const int p_A = K1 * p_rstep + K2 * p_cstep, p_Bd = p_rstep - PW * p_cstep;
  unsigned p_off0 = 0, p_wlo = 0, p_whb = 0;       // p_whb: w_8.. in bits 0-15, invalid bits of chunk i at bit 16 + i
  unsigned p_soff[NPS];
  if constexpr (PRE) {
    const int py0 = r0 / PW, px0 = r0 - py0 * PW;
    const int o_rows = S2 ? 2 * (ty0 - 1) - sy_base : ty0 - 1 - sy_base;
    const int o_cols = S2 ? 2 * (tx0 - 1) : tx0 - 1;
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
#pragma unroll
    for (int i = 0; i < NPS; ++i) {               // scale words of patch rows tid (and tid + 256)
      const int pr = tid + NT * i;
      const int py = pr / PW, px = pr - py * PW;
      const bool ok = pr < PROWS && (unsigned)(ty0 - 1 + py) < hlim && (unsigned)(tx0 - 1 + px) < wlim;
      p_soff[i] = ok ? (unsigned)(((o_rows + (S2 ? 2 : 1) * py) * a.Ws + o_cols + (S2 ? 2 : 1) * px) * ldxs) : DG_OOB_OFF;
    }
    asm volatile("" : "+v"(p_off0), "+v"(p_wlo), "+v"(p_whb));
  }
  u32x4_t rp[NPL];
  unsigned rps[NPS];
  auto load_patch = [&](int vcb) {
    const int plane = plane_of(vcb), cb = vcb - plane * ncbr;
    const int ppy = plane >> 1, ppx = plane & 1;
    // (bases made provably workgroup-uniform: otherwise each load below sits in its own waterfall loop, see conv_halo.hip)
    auto uni = [](const void* p) {
      const unsigned long long b = (unsigned long long)p;
      return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b));
    };
    const long long pl = (PRE && S2) ? (long long)(ppy * a.Ws + ppx) : 0;   // PRE: the parity plane only moves the uniform bases
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(uni(Xb + (long long)cb * KC * EPC * ES + pl * a.ldx * ES), 0, (int)DG_OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(uni(XSb + cb * 4 + pl * ldxs), 0, (int)DG_OOB_OFF, 0x00020000);
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int w = (int)__builtin_amdgcn_ubfe(i < 8 ? p_wlo : p_whb, i < 8 ? 4 * i : 4 * (i - 8), 4);
        const unsigned bad = (unsigned)__builtin_amdgcn_sbfe((int)p_whb, 16 + i, 1);          // all ones when the chunk is outside
        const unsigned off = ((unsigned)__mul24(w, p_Bd) + p_off0 + (unsigned)(i * p_A)) | bad;  // >= DG_OOB_OFF: the load returns 0
        rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NPS; ++i) {
        rps[i] = __builtin_amdgcn_raw_buffer_load_b32(rxs, p_soff[i], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      int r0v = r0;
      asm volatile("" : "+v"(r0v));
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int pr = r0v + RPP * i;
        const int py = pr / PW, px = pr - py * PW;
        const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
        const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
        const unsigned off = !ok ? DG_OOB_OFF : (unsigned)(((sy - sy_base) * a.Ws + sx) * a.ldx * ES) + cc * 16;
        rp[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NPS; ++i) {               // scale words of patch rows tid (and tid + 256)
        const int pr = tid + NT * i;
        const int py = pr / PW, px = pr - py * PW;
        const int sy = S2 ? 2 * (ty0 - 1 + py) + ppy : ty0 - 1 + py, sx = S2 ? 2 * (tx0 - 1 + px) + ppx : tx0 - 1 + px;
        const bool ok = pr < PROWS && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
        rps[i] = __builtin_amdgcn_raw_buffer_load_b32(rxs, ok ? (unsigned)(((sy - sy_base) * a.Ws + sx) * ldxs) : DG_OOB_OFF, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  char* const st_base = s_patch + r0 * PITCH + cc * 16;
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int pr = r0 + RPP * i;
      if (pr < PROWS) *reinterpret_cast<uint4*>(st_base + i * RPP * PITCH) = __builtin_bit_cast(uint4, rp[i]);
    }
#pragma unroll
    for (int i = 0; i < NPS; ++i)
      if (tid + NT * i < PROWS) *reinterpret_cast<unsigned*>(s_ps + (tid + NT * i) * 4) = rps[i];
  };
  typedef int i32x4h_t __attribute__((ext_vector_type(4)));
  i32x4h_t w_rs, ws_rs;
  int w_dst0 = 0, ws_dst = 0;
  auto dma_setup = [&](int vcb, int tap, int slot) {
    const unsigned code = tap_code(vcb, tap);
    const int cbr = vcb - plane_of(vcb) * ncbr;
    const long long wo = (long long)(code >> 4) * a.Cred + cbr * KC * EPC;
    const unsigned long long wbase = (unsigned long long)(Wb + wo * ES);
    w_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)wbase);
    w_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32) & 0xffff);
    w_rs[2] = (int)DG_OOB_OFF;
    w_rs[3] = 0x00020000;
    w_dst0 = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_w + slot * (BC * WROW) + (wave * 32) * WROW)));
    const unsigned long long sbase = (unsigned long long)(WSb + (long long)(code >> 4) * (a.Cred >> 5) + cbr * 4);
    ws_rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)sbase);
    ws_rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(sbase >> 32) & 0xffff);
    ws_rs[2] = (int)DG_OOB_OFF;
    ws_rs[3] = 0x00020000;
    ws_dst = __builtin_amdgcn_readfirstlane(
        (int)(unsigned long long)((__attribute__((address_space(3))) char*)(s_ws + slot * (BC * 4) + (wave & (NSW - 1)) * 256)));
  };
  auto dma_piece = [&](int i) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(w_dst0 + i * 8 * WROW), "v"(woff[i]), "s"(w_rs) : "memory");
  };
  // every wave issues the scale piece (waves 2/3 re-write the rows of waves 0/1 with the same bytes), so the per-wave count
  // of outstanding vector-memory operations -- what the counted s_waitcnt below relies on -- is the same in all four waves
  auto dma_scales = [&]() {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(ws_dst), "v"(wsoff), "s"(ws_rs) : "memory");
  };
  auto dma_w = [&](int cb, int tap, int slot) {
    dma_setup(cb, tap, slot);
#pragma unroll
    for (int i = 0; i < NWL; ++i) dma_piece(i);
    dma_scales();
  };
  auto barrier_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };
  auto barrier_keep_patch = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPL + NPS) : "memory"); __syncthreads(); };   // patch chunks + scale words

  // accumulators start at the bias (conv_halo.hip: its load latency hides in the prologue instead of sitting in every tile's epilogue)
  f32x4_t acc[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4_t b = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int cbq = c0 + (tid >> 8) * 128 + (j >> 2) * 64 + 16 * ((tid & 63) >> 4) + 4 * (j & 3);
    if (a.bias && cbq < a.Nout) { const float4 b4 = *reinterpret_cast<const float4*>(a.bias + cbq); b = f32x4_t{b4.x, b4.y, b4.z, b4.w}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = b;
  }

  const char* fa_k[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) fa_k[kk] = s_w + (wh * 128 + l15) * WROW + (((kk * 4 + g) ^ ((l15 >> 1) & 7)) * 16);
  const char* const fb_lane = s_patch + l15 * PITCH + g * 16;
  const char* const sa_lane = s_ws + (wh * 128 + l15) * 4 + g;
  const char* const sb_lane = s_ps + l15 * 4 + g;
  // A lane's 32-byte operand = chunk g of the row's first and second 64-byte half.  All fragments of a step are read and
  // consumed INSIDE the step (nothing is carried around the loop: a loop-carried 8-register value gets split into two
  // 4-register halves by SROA and copied back together, 32 v_mov per step), and no read result is ever passed through an
  // asm statement (that forces a full s_waitcnt lgkmcnt(0) right behind the read -- the first version of this kernel
  // paid two serialised LDS latencies per weight fragment that way).
  auto ld32 = [](const char* p0, const char* p1) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p0), hi = *reinterpret_cast<const uint4*>(p1);
    i32x8_t o;
    o[0] = (int)lo.x; o[1] = (int)lo.y; o[2] = (int)lo.z; o[3] = (int)lo.w;
    o[4] = (int)hi.x; o[5] = (int)hi.y; o[6] = (int)hi.z; o[7] = (int)hi.w;
    return o;
  };
  // one tap-step: B fragments (4 pixel rows) + the first weight fragment are read at the top -- the DMA issue of W[s+1] that
  // follows covers their latency -- then weight fragment j+1 is read while the four MFMAs of fragment j run (two fragment
  // registers sets of 8, ping-pong).  The asm fences keep every read in its slot and every MFMA row between its fences
  // (an MFMA is not a memory operation: without the accumulator operands the compiler sinks MFMAs below later reads).
#define F8_FENCE(j) asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]), "+v"(acc[j][2]), "+v"(acc[j][3]) :: "memory")
  auto adv = [&](int& c_, int& t_) { if (++t_ == ntaps_of(c_)) { t_ = 0; ++c_; } };
  auto patch_row = [&](int vcb_, int tap_) {
    const unsigned code = tap_code(vcb_, tap_);
    const int dy = (int)(code & 3u) - 1, dx = (int)((code >> 2) & 3u) - 1;
    return (wq * 4 + 1 + dy) * PW + 1 + dx;
  };

  load_patch(0);
  int cb = 0, tap = 0, cbw = 0, tapw = 0;
  dma_w(0, 0, 0);
  adv(cbw, tapw);                    // -> W[1], issued at the top of step 0
  store_patch();
  barrier_all();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    int ntap = tap + 1, ncbn = cb;
    const int ntaps_cb = ntaps_of(cb);
    if (ntap == ntaps_cb) { ntap = 0; ncbn = cb + 1; }
    const bool swap = ntap == 0 && more;
    const bool patch_now = cb + 1 < ncb && (ntaps_cb == 1 || tap == 0);     // fetch the next block's patch during its predecessor's first step
    const int slot = s & 1, pa = slot * (BC * WROW);
    const int prow = patch_row(cb, tap);
    const char* const pb = fb_lane + prow * PITCH;
    const char* const sap = sa_lane + slot * (BC * 4);
    i32x8_t fb[4], fa[2];
    int sb[4], sa[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fb[i] = ld32(pb + i * PW * PITCH, pb + i * PW * PITCH + 64);
      sb[i] = *reinterpret_cast<const unsigned char*>(sb_lane + (prow + i * PW) * 4);
    }
    fa[0] = ld32(fa_k[0] + pa, fa_k[1] + pa);
    sa[0] = *reinterpret_cast<const unsigned char*>(sap);
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    if (more) {                      // W[s+1] -> the other slot (free since the barrier that ended step s-1)
      dma_setup(cbw, tapw, slot ^ 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) dma_piece(q);
      dma_scales();
    }
    adv(cbw, tapw);
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    // the next block's patch goes out behind the DMA pieces (the barrier's counted wait then leaves it in flight) and before
    // the MFMA rows: here only the B fragments and one weight fragment are live beside the prefetch registers
    if (patch_now) {
      load_patch(cb + 1);
      asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < 7) {
        fa[(j + 1) & 1] = ld32(fa_k[0] + pa + (j + 1) * 16 * WROW, fa_k[1] + pa + (j + 1) * 16 * WROW);
        sa[(j + 1) & 1] = *reinterpret_cast<const unsigned char*>(sap + (j + 1) * 64);
      }
      F8_FENCE(j); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[j & 1], fb[i], acc[j][i], 0, 0, 0, sa[j & 1], 0, sb[i]);
      F8_FENCE(j); __builtin_amdgcn_sched_barrier(0);
    }
    // every wave has read all it needs of this step: the barrier frees slot s&1 (and, at a block end, the patch); W[s+1] has landed
    if (patch_now && !swap) barrier_keep_patch(); else barrier_all();
    if (swap) {                      // channel-block boundary: the single-buffered patch is rewritten, then published
      store_patch();
      barrier_all();
    }
    tap = ntap; cb = ncbn;
  }
#undef F8_FENCE
  if constexpr (SEG) halo_epilogue<bf16_t, 2, true>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g, cls >> 1, cls & 1);
  else halo_epilogue<bf16_t, 2, true>(a, acc, img, ty0, tx0, c0 + wh * 128, wq, 0, l15, g);
}

template <bool S2, int NW = 4, bool SEG = false>
static int gg_launch_halo4w_f8(GGArgs& a, const F8Args& f, int N, hipStream_t st) {
  constexpr int BC = 32 * NW;
  constexpr int LDS_BYTES = 324 * 144 + 2 * BC * 128 + 2 * BC * 4 + 324 * 4;
  DG_SET_MAX_LDS_ONCE((&gg_halo4w_f8_kernel<S2, NW, SEG>), LDS_BYTES);
  const int tiles_x = (a.Wg + 15) / 16, tiles_y = (a.Hg + 15) / 16;
  a.nct = (unsigned)((a.Nout + BC - 1) / BC);
  a.nwg = a.nct * (unsigned)(tiles_x * tiles_y * N) * (SEG ? 4u : 1u);
  g_last_kinds |= 32;
  hipLaunchKernelGGL((gg_halo4w_f8_kernel<S2, NW, SEG>), dim3(a.nwg), dim3(64 * NW), LDS_BYTES, st, a, f, tiles_x, tiles_y);
  return dg_check_launch();
}

int gg_launch_halo_f8(GGArgs& a, const F8Args& f, int N, bool s2, int nw, hipStream_t st) {
  if (nw == 8 && 2ll * a.Ws * a.ldx >= (1ll << 23)) nw = 4;            // the eight-wave instance multiplies the row step in 24 bits
  if (a.seg) return gg_launch_halo4w_f8<false, 4, true>(a, f, N, st);
  if (!s2) return gg_launch_halo4w_f8<false>(a, f, N, st);
  return nw == 8 ? gg_launch_halo4w_f8<true, 8>(a, f, N, st) : gg_launch_halo4w_f8<true>(a, f, N, st);
}
