// Conv 3x3 weight gradient (and the GP double-backward term): dw[co][tap][ci] += sum_p u[p,co] * x[src(p,tap),ci]
//
// GEMM view: M = Cout, N = Cin (one tap per workgroup), K = output pixels (N*Ho*Wo, up to 33 M at
// cfg2) split over workgroups; partial tiles are accumulated into the fp32 gradient with
// global_atomic_add_f32.  Both operands are "K-major" in NHWC memory (pixel index is the slow
// dimension), so tiles are staged as [pixel][channel] in LDS (coalesced 16-B loads along channels)
// and fragments are read TRANSPOSED:
//   bf16 : ds_read_b64_tr_b16 (hardware 4x16 transpose) feeding v_mfma_f32_32x32x16_bf16
//   fp32 : ds_read_b32 (one k per lane, conflict-free rows) feeding v_mfma_f32_32x32x2_f32
// The 32x32 shapes put 32 consecutive ci on the lanes of one accumulator register, so every atomic
// wave-instruction adds two full 128-B row segments (the full-rate shape for float atomics).
#include "dg_internal.h"

#include <stdlib.h>

struct WGArgs {
  const void* x; const void* u; float* dw;
  long long ldx, ldu;
  int H, W, Ho, Wo, stride, Cin, Cout;
  int u_ps, cps_chunks;
  int Mpix, ppb;
  int nci_t;
  float* db;       // IM2COL mode: bias gradient (column 18 of the im2col operand is the constant 1)
  // dense-block mode of the wide kernel (dg_conv3x3_wgrad_dense): conv k = 1..nconv reads input tiles 0..k-1 of a shared slab and
  // its adjoint is channel tile k-1 of the adjoint slab: only the (adjoint tile t, input tile <= t) pairs exist, and pair rows go
  // to conv t+1's own gradient (row stride 9 * (t + 1) * 128) / bias gradient
  int tri;
  float* dwk[8]; float* dbk[8];
  // deterministic mode (dg_internal.h DetPlan): split `by` accumulates into copy `by` of the target region inside the workspace
  // instead of the gradient itself: dw / dwk[] / dbk[] keep their offsets from det_base, db sits at det_db_off of the copy
  float* det_ws; const float* det_base; long long det_stride, det_db_off;
  const unsigned char* ex; const unsigned char* eu;   // fp8 kernel: E8M0 exponent byte per 32-channel block of x / of the adjoint
  // wide kernels, > 72 tiles per pixel range: launch order in UNITS of `grp` tiles (a group of adjoint tiles x all input tiles x 3 tap
  // rows of one pixel range) that stay on one XCD (wg_group_order)
  int grp, ngrp, nunits, gtiles;
};

// > 72 tiles per pixel range (the 512 / 1024-channel layers): the plain XCD remap would put 96-192 workgroups that want the same
// pixel range on 64 resident slots of one XCD (measured -5...-12 % in round 1); the launch order puts workgroup i on XCD i % 8, i.e.
// tile index % 8 -- with the input tile as the fastest tile coordinate every XCD then owns 1 / 8 of the input tiles (or one of
// 1 / 2 / 4) and reads ALL of the adjoint: fabric bytes ~ 8 dy + max(1, 8 / nci_t) x.  Alternative: units of <= 64 tiles -- cg
// adjoint-channel tiles x all input tiles x 3 tap rows of one pixel range -- unit u on XCD u % 8, its tiles back to back: ~ dy +
// ngrp x.  The cheaper estimate wins (rocprofv3 FETCH_SIZE, profiles/r04_wg_order_ab.log, batch 32: 512 -> 1024 at 128^2 11.0 -> 3.5 GB per
// launch and +2.5 % (bf16) / +2 % (fp8) in time; 1024 -> 1024 stride 2 keeps the launch order: 3.9 GB against 10.3 grouped, -6 %).
static void wg_group_order(WGArgs& a, int nco_t, int ntiles, int splits) {
  a.grp = 0;
  if (a.tri || ntiles <= 72) return;
  const int per_co = 3 * a.nci_t;
  int cg = 0;
  for (int c = 1; c <= nco_t; ++c)
    if (nco_t % c == 0 && c * per_co <= 64) cg = c;
  if (!cg) return;
  const double dy_b = (double)a.Mpix * a.Cout, x_b = (double)a.Mpix * a.stride * a.stride * a.Cin;
  const double in_order = 8.0 * dy_b + x_b * (a.nci_t >= 8 ? 1.0 : 8.0 / a.nci_t), grouped = dy_b + x_b * (nco_t / cg);
  if (grouped > 0.8 * in_order) return;
  a.grp = cg * per_co; a.ngrp = nco_t / cg; a.nunits = splits * a.ngrp; a.gtiles = ntiles;
}
// number of workgroups of a grouped launch: every XCD gets the same number of unit slots
static unsigned wg_group_blocks(const WGArgs& a) { return 8u * (unsigned)((a.nunits + 7) / 8) * (unsigned)a.grp; }

// decode of a workgroup's (tile, pixel split); false = a padding workgroup of a grouped launch
__device__ __forceinline__ bool wg_decode(const WGArgs& a, int& bx, int& by) {
  const unsigned lin0 = blockIdx.y * gridDim.x + blockIdx.x;
  if (a.grp) {
    const unsigned xcd = lin0 & 7u, slot = lin0 >> 3;
    const unsigned unit = (slot / (unsigned)a.grp) * 8u + xcd, within = slot % (unsigned)a.grp;
    if (unit >= (unsigned)a.nunits) return false;
    by = (int)(unit / (unsigned)a.ngrp); bx = (int)((unit % (unsigned)a.ngrp) * (unsigned)a.grp + within);
    return true;
  }
  const unsigned lin = gridDim.x <= 72 ? xcd_remap(lin0, gridDim.x * gridDim.y) : lin0;
  bx = (int)(lin % gridDim.x); by = (int)(lin / gridDim.x);
  return true;
}
#define WG_DET_PTR(a, p, by) ((a).det_ws ? (a).det_ws + (long long)(by) * (a).det_stride + ((p) - (a).det_base) : (p))
#define WG_DET_DB(a, by) ((a).det_ws ? (a).det_ws + (long long)(by) * (a).det_stride + (a).det_db_off : (a).db)

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4w_t;
#define WG_OOB_OFF 0x80000000u

// ROWSTEP: Wo % KP == 0, so every K-step (KP consecutive output pixels) lies inside ONE output row: image / row
// / first column are workgroup-uniform scalars and each thread's byte offsets are loop constants (raw buffer
// loads, out-of-image taps get an out-of-range offset and read zeros).
// IM2COL (layers with <= 2 real input channels, stride 1, e.g. the critic's first conv at 1024^2): instead of one
// workgroup per tap, the input operand is the im2col row [9 taps x 2 channels | 1 | 0...] (64 columns) built on
// the fly, so the adjoint (the only large tensor) is read ONCE for all taps and for the bias gradient.
template <typename T, int BCO, int BCI, bool ROWSTEP, int KP, bool IM2COL = false>   // KP = pixels per K-step
__global__ __launch_bounds__(256) void wg_kernel(const WGArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int CPRU = BCO / EPC, CPRX = BCI / EPC;
  constexpr int NU = KP * CPRU / 256, NX = KP * CPRX / 256;   // 16-B chunks per thread
  static_assert(NU >= 1 && NX >= 1, "tile too small");
  constexpr int FA = BCO / 64, FB = BCI / 64;    // 32x32 fragments per wave (2x2 waves)
  // bf16 rows are padded by 64 B: with 256-B (or 128-B) rows every K row of a ds_read_b64_tr_b16 block lands on the same
  // banks (4-way conflict, measured 60 % of LDS cycles); a pitch of 64 B mod 256 B spreads the 32 lanes of a half-wave
  // (2 column groups x 4 rows x 4 column quads, 8 B each) over all 64 banks.
  constexpr int PADE = sizeof(T) == 2 ? 32 : 0;
  constexpr int LDU = BCO + PADE, LDX = BCI + PADE;
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_dsm[];   // 2 * KP * (LDU + LDX) elements
  T* const smem = reinterpret_cast<T*>(wg_dsm);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware remap of the linear workgroup id: the 9 tap-workgroups (and channel tiles) of one pixel range become
  // consecutive ids on ONE XCD, so the tiles they all re-read are served by that XCD's L2 instead of the fabric
  // (measured: +15-35 % on layers with <= 72 tiles per pixel range, -5-12 % on the widest layers, hence the gate)
  const unsigned lin0 = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned lin = gridDim.x <= 72 ? xcd_remap(lin0, gridDim.x * gridDim.y) : lin0;
  const int bx = (int)(lin % gridDim.x), by = (int)(lin / gridDim.x);
  const int ci_t = IM2COL ? 0 : bx % a.nci_t;
  const int tap = IM2COL ? 4 : (bx / a.nci_t) % 9;
  const int co_t = IM2COL ? bx : bx / (a.nci_t * 9);
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int dr = tap / 3 - 1, dc = tap % 3 - 1;
  const int pbeg = by * a.ppb;
  const int pend = min(a.Mpix, pbeg + a.ppb);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ U = reinterpret_cast<const T*>(a.u);

  uint4 ru[NU], rx[NX];
  // ---- ROWSTEP state: scalar position of the current K-step and constant per-thread offsets
  int s_n = 0, s_ho = 0, s_wo = 0;
  unsigned uoffc[NU], xoffc[NX];
  int xrow[NX];
  constexpr int TPC = EPC / 2;          // taps per 16-B chunk of the im2col row
  unsigned imoff[IM2COL ? NX : 1][TPC];  // per-thread byte offsets of the chunk's taps (relative to pixel (ho-1, wo0-1))
  if constexpr (ROWSTEP) {
    s_wo = pbeg % a.Wo; const int t = pbeg / a.Wo; s_ho = t % a.Ho; s_n = t / a.Ho;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = tid + 256 * i, row = e / CPRU, col = e % CPRU;
      const int co = co0 + col * EPC;
      if (co >= a.Cout) { uoffc[i] = WG_OOB_OFF; continue; }
      if (!a.u_ps) uoffc[i] = (unsigned)(((long long)row * a.ldu + co) * (int)sizeof(T));
      else {
        const int cchunk = co / EPC, q = cchunk / a.cps_chunks, c = cchunk - q * a.cps_chunks;
        uoffc[i] = (unsigned)((((long long)(q >> 1) * (2 * a.Wo) + 2 * row + (q & 1)) * a.ldu + c * EPC) * (int)sizeof(T));
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, row = e / CPRX, col = e % CPRX;
      const int ci = ci0 + col * EPC;
      xrow[i] = row;
      xoffc[i] = ci < a.Cin ? (unsigned)(((long long)row * a.stride * a.ldx + ci) * (int)sizeof(T)) : WG_OOB_OFF;
      if constexpr (IM2COL) {
#pragma unroll
        for (int tt = 0; tt < TPC; ++tt) {
          const int tp = col * TPC + tt;      // tap index of this pair of columns
          imoff[i][tt] = tp < 9 ? (unsigned)((((long long)(tp / 3) * a.W + row + tp % 3) * a.ldx) * (int)sizeof(T)) : WG_OOB_OFF;
        }
      }
    }
  }
  auto gload_row = [&]() {
    // scalar bases of this K-step
    const long long ub = !a.u_ps ? ((long long)(s_n * a.Ho + s_ho) * a.Wo + s_wo) * a.ldu
                                 : ((long long)(s_n * 2 * a.Ho + 2 * s_ho) * (2 * a.Wo) + 2 * s_wo) * a.ldu;
    const int hi = s_ho * a.stride + dr;
    const bool row_ok = (unsigned)hi < (unsigned)a.H;
    const int wi0 = s_wo * a.stride + dc;
    const long long xb = ((long long)(s_n * a.H + hi) * a.W + wi0) * a.ldx;
    __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)(U + ub), 0, (int)WG_OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(X + xb), 0, (int)WG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NU; ++i) ru[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rU, uoffc[i], 0, 0));
    if constexpr (IM2COL) {
      // base = pixel (s_ho - 1, s_wo - 1) of image s_n; tap (r, s) of output pixel wo0+row is pixel (s_ho-1+r, s_wo-1+row+s)
      const long long xb2 = ((long long)(s_n * a.H + s_ho - 1) * a.W + s_wo - 1) * a.ldx;
      __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)(X + xb2), 0, (int)WG_OOB_OFF, 0x00020000);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int col = (tid + 256 * i) % CPRX;
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int tt = 0; tt < TPC; ++tt) {
          const int tp = col * TPC + tt;
          const int hh = s_ho - 1 + tp / 3, ww = s_wo - 1 + xrow[i] + tp % 3;
          const unsigned vo = (tp < 9 && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W) ? imoff[i][tt] : WG_OOB_OFF;
          if constexpr (sizeof(T) == 2) {
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rI, vo, 0, 0);
            if (tp == 9) v = 0x3f80u;                          // bf16 (1.0, 0.0): the bias column
            w[tt] = v;
          } else {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2w_t;
            u32x2w_t v = __builtin_amdgcn_raw_buffer_load_b64(rI, vo, 0, 0);
            if (tp == 9) { v[0] = 0x3f800000u; v[1] = 0u; }
            w[2 * tt] = v[0]; w[2 * tt + 1] = v[1];
          }
        }
        rx[i] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int wi = wi0 + xrow[i] * a.stride;
        const unsigned vo = (row_ok && (unsigned)wi < (unsigned)a.W) ? xoffc[i] : WG_OOB_OFF;
        rx[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, 0, 0));
      }
    }
    s_wo += KP;
    if (s_wo >= a.Wo) { s_wo = 0; if (++s_ho == a.Ho) { s_ho = 0; ++s_n; } }
  };
  // Each thread owns NU chunks of the adjoint tile and NX chunks of the input tile; chunk e = tid + 256*i is
  // (row = e / chunks_per_row, col = e % chunks_per_row).  The pixel coordinates of every owned row are
  // tracked INCREMENTALLY (the K loop advances by KP pixels per step), so the loop has no integer division.
  int un[NU], uho[NU], uwo[NU];       // (image, ho, wo) of the adjoint rows (only needed for the shuffled layout)
  int xn[NX], xho[NX], xwo[NX];
  auto init_pos = [&](int p, int& n, int& ho, int& wo) {
    wo = p % a.Wo; const int t = p / a.Wo; ho = t % a.Ho; n = t / a.Ho;
  };
  auto advance_pos = [&](int& n, int& ho, int& wo) {
    wo += KP;
    while (wo >= a.Wo) { wo -= a.Wo; if (++ho == a.Ho) { ho = 0; ++n; } }
  };
#pragma unroll
  for (int i = 0; i < NU; ++i) init_pos(pbeg + (tid + 256 * i) / CPRU, un[i], uho[i], uwo[i]);
#pragma unroll
  for (int i = 0; i < NX; ++i) init_pos(pbeg + (tid + 256 * i) / CPRX, xn[i], xho[i], xwo[i]);
  auto gload_gen = [&](int pb) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = tid + 256 * i, row = e / CPRU, col = e % CPRU;
      const int p = pb + row, co = co0 + col * EPC;
      const bool ok = p < pend && co < a.Cout;
      long long off;
      if (!a.u_ps) off = (long long)p * a.ldu + co;
      else {
        const int cchunk = co / EPC, q = cchunk / a.cps_chunks, c = cchunk - q * a.cps_chunks;
        off = (long long)(((un[i] * 2 * a.Ho + 2 * uho[i] + (q >> 1))) * (2 * a.Wo) + 2 * uwo[i] + (q & 1)) * a.ldu + c * EPC;
      }
      uint4 v = *reinterpret_cast<const uint4*>(U + (ok ? off : 0ll));
      ru[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      advance_pos(un[i], uho[i], uwo[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, row = e / CPRX, col = e % CPRX;
      const int p = pb + row, ci = ci0 + col * EPC;
      const int hi = xho[i] * a.stride + dr, wi = xwo[i] * a.stride + dc;
      const bool ok = p < pend && ci < a.Cin && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      const long long off = (long long)((xn[i] * a.H + hi) * a.W + wi) * a.ldx + ci;
      uint4 v = *reinterpret_cast<const uint4*>(X + (ok ? off : 0ll));
      rx[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      advance_pos(xn[i], xho[i], xwo[i]);
    }
  };
  auto gload = [&](int pb) {
    if constexpr (ROWSTEP) gload_row(); else gload_gen(pb);
  };
  auto lstore = [&](int buf) {
    T* su = smem + buf * KP * (LDU + LDX);
    T* sx = su + KP * LDU;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = tid + 256 * i, row = e / CPRU, col = e % CPRU;
      *reinterpret_cast<uint4*>(su + row * LDU + col * EPC) = ru[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, row = e / CPRX, col = e % CPRX;
      *reinterpret_cast<uint4*>(sx + row * LDX + col * EPC) = rx[i];
    }
  };

  f32x16_t acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wco = wave & 1, wci = wave >> 1;
  const int r32 = lane & 31, h = lane >> 5;
  const int nsteps = (pend - pbeg + KP - 1) / KP;
  if (nsteps <= 0) return;

  // store-before-barrier pipeline: loads of step ks+1 are issued before the MFMA phase of step ks and stored after it.
  // (The write-after-barrier order with a full-step latency budget was measured: +5 % on the compute-bound layers,
  // -10 % on the 512^2/1024^2 layers, -7 % on the whole step's weight gradients.)
  gload(pbeg);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nsteps; ++ks) {
    const bool more = ks + 1 < nsteps;
    if (more) gload(pbeg + (ks + 1) * KP);
    const T* su = smem + cur * KP * (LDU + LDX);
    const T* sx = su + KP * LDU;
    if constexpr (sizeof(T) == 2) {
      // lane -> (row q of a 4x16 block, column group pp) address for the transposed read
      const int grp16 = (lane >> 4) & 1, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
#pragma unroll
      for (int kk = 0; kk < KP / 16; ++kk) {
        bf16x8_t fa[FA], fb[FB];
#pragma unroll
        for (int f = 0; f < FA; ++f) {
          const int ch = wco * (BCO / 2) + 32 * f + 16 * grp16 + 4 * pp;
          const int k0 = kk * 16 + 8 * h + q;
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(su + k0 * LDU + ch));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(su + (k0 + 4) * LDU + ch));
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fa[f] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int f = 0; f < FB; ++f) {
          const int ch = wci * (BCI / 2) + 32 * f + 16 * grp16 + 4 * pp;
          const int k0 = kk * 16 + 8 * h + q;
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sx + k0 * LDX + ch));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sx + (k0 + 4) * LDX + ch));
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fb[f] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
          for (int j = 0; j < FB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int k = 0; k < KP; k += 2) {
        float fa[FA], fb[FB];
#pragma unroll
        for (int f = 0; f < FA; ++f) fa[f] = su[(k + h) * LDU + wco * (BCO / 2) + 32 * f + r32];
#pragma unroll
        for (int f = 0; f < FB; ++f) fb[f] = sx[(k + h) * LDX + wci * (BCI / 2) + 32 * f + r32];
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
          for (int j = 0; j < FB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // D[row = co][col = ci]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const long long ldw = 9ll * a.Cin;
  float* const dwp = WG_DET_PTR(a, a.dw, by);
  float* const dbp = a.db ? WG_DET_DB(a, by) : nullptr;
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      const int ci = ci0 + wci * (BCI / 2) + 32 * j + r32;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = co0 + wco * (BCO / 2) + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if constexpr (IM2COL) {
          if (co < a.Cout) {
            if (ci < 18) atomicAdd(dwp + (long long)co * ldw + (ci >> 1) * a.Cin + (ci & 1), acc[i][j][reg]);
            else if (ci == 18 && dbp) atomicAdd(dbp + co, acc[i][j][reg]);
          }
        } else {
          if (co < a.Cout && ci < a.Cin) atomicAdd(dwp + (long long)co * ldw + tap * a.Cin + ci, acc[i][j][reg]);
        }
      }
    }
}

// Split-K factor of a weight-gradient launch: ntiles x splits workgroups, the LARGEST count that still fits `target` (a whole
// number of rounds of the chip's resident-workgroup slots, or the atomics budget `cap` if that is smaller).  Rounding the
// split count up instead spilled a few workgroups into an extra, nearly empty round (96 tiles x 22 splits = 2112 = 4.125
// rounds of 512 slots: -5 % against 1536 or 3072).  Returns the split count and the pixels per workgroup (multiple of 64).
static int wg_pick_splits(int ntiles, long long target, long long cap, int Mpix, int* ppb) {
  if (cap < 512) cap = 512;
  if (target > cap) target = cap;
  int splits = (int)(target / ntiles);
  const int max_splits = (Mpix + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  *ppb = ((Mpix + splits - 1) / splits + 63) / 64 * 64;
  return (Mpix + *ppb - 1) / *ppb;
}

// Deterministic mode: plan `splits` copies of the launch's target region -- [dw | db] (Cout * 9 * Cin + Cout floats), or in dense-block
// mode the span of the flat gradient buffer that holds the block's weight and bias gradients -- and return the split count granted.
static int wg_det_begin(WGArgs& a, int splits, hipStream_t st, DetPlan* p, float** lo_out, long long* span_out) {
  p->ws = nullptr; p->copies = splits;
  if (!dg_det_on()) return splits;
  float* lo = a.dw; float* hi = a.dw + (long long)a.Cout * 9 * a.Cin;
  if (a.tri) {
    lo = hi = nullptr;
    long long sum = 0;
    const int n = a.Cout / 128;
    for (int k = 0; k < n; ++k) {
      float* b = a.dwk[k]; float* e = b + 128ll * 9 * (k + 1) * 128;
      if (!lo || b < lo) lo = b;
      if (!hi || e > hi) hi = e;
      sum += e - b;
      if (a.dbk[k]) { if (a.dbk[k] < lo) lo = a.dbk[k]; if (a.dbk[k] + 128 > hi) hi = a.dbk[k] + 128; sum += 128; }
    }
    if (hi - lo > 2 * sum) { p->copies = 1; return 1; }     // gradients scattered over the address space: one split, unique writers
  }
  const long long span = hi - lo;
  const long long stride = (span + (a.tri ? 0 : a.Cout) + 3) / 4 * 4;
  if (dg_det_begin(stride, splits, st, p) != DG_OK) return -1;
  if (p->ws) { a.det_ws = p->ws; a.det_base = lo; a.det_stride = stride; a.det_db_off = span; }
  *lo_out = lo; *span_out = span;
  return p->copies;
}
static int wg_det_end(const WGArgs& a, const DetPlan& p, float* lo, long long span, hipStream_t st) {
  if (!p.ws) return DG_OK;
  int rc = dg_det_reduce(p, 0, lo, span, st);
  if (rc == DG_OK && !a.tri && a.db) rc = dg_det_reduce(p, a.det_db_off, a.db, a.Cout, st);
  return rc;
}

// resident workgroups per CU of a kernel instance (LDS- or register-limited), for sizing launches in whole rounds
template <typename K>
static int wg_resident(K kernel, int lds_bytes) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, (size_t)lds_bytes) != hipSuccess || n < 1) n = 2;
  return n;
}

template <typename T>
static int wg_launch_im2col(WGArgs& a, hipStream_t st) {
  const bool big_co = a.Cout > 64;
  const int bco = big_co ? 128 : 64;
  const int nco_t = (a.Cout + bco - 1) / bco;
  constexpr int PADE = sizeof(T) == 2 ? 32 : 0;
  const int lds = 2 * 32 * (bco + 64 + 2 * PADE) * (int)sizeof(T);
  static std::atomic<int> occ_cache[2] = {{0}, {0}};
  int occ = occ_cache[big_co].load(std::memory_order_relaxed);
  if (!occ) {
    occ = big_co ? wg_resident(wg_kernel<T, 128, 64, true, 32, true>, lds) : wg_resident(wg_kernel<T, 64, 64, true, 32, true>, lds);
    occ_cache[big_co].store(occ, std::memory_order_relaxed);
  }
  // two whole rounds of the resident slots (2304 workgroups on 5 x 256 slots were 1.8 rounds)
  int splits = 2 * 256 * occ / nco_t;
  const int max_splits = (a.Mpix + 2047) / 2048;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.ppb = ((a.Mpix + splits - 1) / splits + 31) / 32 * 32;
  splits = (a.Mpix + a.ppb - 1) / a.ppb;
  DetPlan plan; float* lo = nullptr; long long span = 0;
  const int granted = wg_det_begin(a, splits, st, &plan, &lo, &span);
  if (granted < 0) return DG_ERR_LAUNCH;
  if (granted != splits) { a.ppb = ((a.Mpix + granted - 1) / granted + 31) / 32 * 32; splits = (a.Mpix + a.ppb - 1) / a.ppb; }
  dim3 grid(nco_t, splits);
  if (big_co) hipLaunchKernelGGL((wg_kernel<T, 128, 64, true, 32, true>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((wg_kernel<T, 64, 64, true, 32, true>), grid, dim3(256), lds, st, a);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return wg_det_end(a, plan, lo, span, st);
}

template <typename T>
static int wg_launch(WGArgs& a, hipStream_t st) {
  const bool big_co = a.Cout > 64, big_ci = a.Cin > 64;
  const int bco = big_co ? 128 : 64, bci = big_ci ? 128 : 64;
  const int nco_t = (a.Cout + bco - 1) / bco;
  a.nci_t = (a.Cin + bci - 1) / bci;
  const int ntiles = nco_t * 9 * a.nci_t;
  // every workgroup ends with a 64-KB (tile) atomic accumulate at ~1.3 TB/s chip-wide: cap the workgroup count so
  // that this traffic stays below ~1/4 of the MFMA time (estimated at 600 TFLOP/s), but keep >= 512 workgroups
  const double flops = 2.0 * 9 * a.Cin * (double)a.Cout * a.Mpix;
  const long long cap = (long long)(flops * 5.4e-4 / (bco * bci * 4.0));
  const bool rs = a.Wo % 32 == 0;
  // 64-pixel K-steps (half the barriers, half the resident workgroups) were measured equal to 32-pixel steps, and
  // 256x128 / 128x256 tiles (2 workgroups per CU, 25 % fewer staged bytes per flop) 8-20 % SLOWER than 128x128 with 3
  // workgroups per CU: neither is built.
  constexpr int PADE = sizeof(T) == 2 ? 32 : 0;
  DetPlan plan; float* lo = nullptr; long long span = 0;
#define WG_LAUNCH1(BCO, BCI, RS, KPV)                                                                 \
  do {                                                                                                \
    const int lds = 2 * KPV * (BCO + BCI + 2 * PADE) * (int)sizeof(T);                                \
    if (lds > 65536) DG_SET_MAX_LDS_ONCE((&wg_kernel<T, BCO, BCI, RS, KPV>), lds);                    \
    static std::atomic<int> occ_cache{0};                                                             \
    int occ = occ_cache.load(std::memory_order_relaxed);                                              \
    if (!occ) { occ = wg_resident(wg_kernel<T, BCO, BCI, RS, KPV>, lds); occ_cache.store(occ, std::memory_order_relaxed); } \
    /* 3 rounds of the resident slots, 2 when more than 3 workgroups share a CU */                    \
    const long long target = (long long)(occ > 3 ? 2 : 3) * 256 * occ;                                \
    int splits = wg_pick_splits(ntiles, target, cap, a.Mpix, &a.ppb);                                 \
    const int granted = wg_det_begin(a, splits, st, &plan, &lo, &span);                               \
    if (granted < 0) return DG_ERR_LAUNCH;                                                            \
    if (granted != splits) splits = wg_pick_splits(ntiles, (long long)ntiles * granted, 1ll << 40, a.Mpix, &a.ppb); \
    dim3 grid(ntiles, splits);                                                                        \
    hipLaunchKernelGGL((wg_kernel<T, BCO, BCI, RS, KPV>), grid, dim3(256), lds, st, a);               \
  } while (0)
#define WG_LAUNCH(BCO, BCI)                                                                           \
  do {                                                                                                \
    if (rs) WG_LAUNCH1(BCO, BCI, true, 32);                                                           \
    else WG_LAUNCH1(BCO, BCI, false, 32);                                                             \
  } while (0)
  if (big_co && big_ci) WG_LAUNCH(128, 128);
  else if (big_co) WG_LAUNCH(128, 64);
  else if (big_ci) WG_LAUNCH(64, 128);
  else WG_LAUNCH(64, 64);
#undef WG_LAUNCH
#undef WG_LAUNCH1
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return wg_det_end(a, plan, lo, span, st);
}

// ------------------------------------------------------------------------------------------------------------------
// Row-of-taps variant (stride 1, Wo % 32 == 0): one workgroup = (tap ROW dr, co tile, ci tile, pixel range) and computes
// the three taps dc = -1, 0, +1 of that row at once.  They share the adjoint tile and read the SAME input tile at row
// offsets 0 / 1 / 2 (34 input pixels for 32 output pixels), so 12.7 KB are staged per 1.6 MFLOP instead of 16 KB per
// 1.0 MFLOP -- ablation builds of the per-tap kernel put half of its time in the global->register->LDS staging (the
// VGPR->LDS store path, ~79 B/clk/CU), 15 % in the MFMAs.  BCO = 64 keeps the three accumulator sets at 96 registers.
// The kernel is LDS-bandwidth-bound (transposing reads + staging writes), so the wave tile is chosen for fragment bytes per
// MFMA: a wave owns ALL 64 adjoint channels x 32 input channels -- the two adjoint fragments are read once per 16 pixels
// and shared by the three taps, each tap adds one input fragment: 5 KB per 6 MFMAs (a 32 x 64 wave tile reads 7 KB).
template <typename T, int BCO>
__global__ __launch_bounds__(256, 3) void wg3_kernel(const WGArgs a) {
  constexpr int EPC = DT<T>::EPC, BCI = 128, KP = 32, XR = KP + 2;
  constexpr int CPRU = BCO / EPC, CPRX = BCI / EPC;
  constexpr int NU = (KP * CPRU + 255) / 256, NX = (XR * CPRX + 255) / 256;
  constexpr int FA = BCO / 32, FB = BCI / 128;      // wave = all BCO adjoint channels x 32 input channels (see above)
  constexpr int PADE = sizeof(T) == 2 ? 32 : 0;
  constexpr int LDU = BCO + PADE, LDX = BCI + PADE;
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_dsm[];   // 2 * (KP * LDU + XR * LDX) elements
  T* const smem = reinterpret_cast<T*>(wg_dsm);
  constexpr int BUF = KP * LDU + XR * LDX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lin0 = blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned lin = gridDim.x <= 72 ? xcd_remap(lin0, gridDim.x * gridDim.y) : lin0;
  const int bx = (int)(lin % gridDim.x), by = (int)(lin / gridDim.x);
  const int ci_t = bx % a.nci_t;
  const int trow = (bx / a.nci_t) % 3;
  const int co_t = bx / (a.nci_t * 3);
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int dr = trow - 1;
  const int pbeg = by * a.ppb;
  const int pend = min(a.Mpix, pbeg + a.ppb);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ U = reinterpret_cast<const T*>(a.u);

  uint4 ru[NU], rx[NX];
  int s_wo = pbeg % a.Wo, s_ho, s_n;
  { const int t = pbeg / a.Wo; s_ho = t % a.Ho; s_n = t / a.Ho; }
  unsigned uoffc[NU], xoffc[NX];
  int xrow[NX];
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    const int e = tid + 256 * i, row = e / CPRU, col = e % CPRU;
    const int co = co0 + col * EPC;
    if (row >= KP || co >= a.Cout) { uoffc[i] = WG_OOB_OFF; continue; }
    if (!a.u_ps) uoffc[i] = (unsigned)(((long long)row * a.ldu + co) * (int)sizeof(T));
    else {
      const int cchunk = co / EPC, q = cchunk / a.cps_chunks, c = cchunk - q * a.cps_chunks;
      uoffc[i] = (unsigned)((((long long)(q >> 1) * (2 * a.Wo) + 2 * row + (q & 1)) * a.ldu + c * EPC) * (int)sizeof(T));
    }
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + 256 * i, row = e / CPRX, col = e % CPRX;
    const int ci = ci0 + col * EPC;
    xrow[i] = row;
    xoffc[i] = (row < XR && ci < a.Cin) ? (unsigned)(((long long)row * a.ldx + ci) * (int)sizeof(T)) : WG_OOB_OFF;
  }
  auto gload = [&]() {
    const long long ub = !a.u_ps ? ((long long)(s_n * a.Ho + s_ho) * a.Wo + s_wo) * a.ldu
                                 : ((long long)(s_n * 2 * a.Ho + 2 * s_ho) * (2 * a.Wo) + 2 * s_wo) * a.ldu;
    const int hi = s_ho + dr;
    const bool row_ok = (unsigned)hi < (unsigned)a.H;
    const int wi0 = s_wo - 1;                                   // input column of tile row 0
    // the descriptor is based one pixel before the row segment; column -1 of the image gets an out-of-range offset
    const long long xb = ((long long)(s_n * a.H + (row_ok ? hi : 0)) * a.W + wi0) * a.ldx;
    __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)(U + ub), 0, (int)WG_OOB_OFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(X + xb), 0, (int)WG_OOB_OFF, 0x00020000);
#pragma unroll
    for (int i = 0; i < NU; ++i) ru[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rU, uoffc[i], 0, 0));
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int wi = wi0 + xrow[i];
      const unsigned vo = (row_ok && (unsigned)wi < (unsigned)a.W) ? xoffc[i] : WG_OOB_OFF;
      rx[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, 0, 0));
    }
    s_wo += KP;
    if (s_wo >= a.Wo) { s_wo = 0; if (++s_ho == a.Ho) { s_ho = 0; ++s_n; } }
  };
  auto lstore = [&](int buf) {
    T* su = smem + buf * BUF;
    T* sx = su + KP * LDU;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = tid + 256 * i, row = e / CPRU, col = e % CPRU;
      if (row < KP) *reinterpret_cast<uint4*>(su + row * LDU + col * EPC) = ru[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i, row = e / CPRX, col = e % CPRX;
      if (row < XR) *reinterpret_cast<uint4*>(sx + row * LDX + col * EPC) = rx[i];
    }
  };

  f32x16_t acc[3][FA][FB];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
      for (int j = 0; j < FB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][i][j][e] = 0.f;

  const int wci = wave;
  const int r32 = lane & 31, h = lane >> 5;
  const int nsteps = (pend - pbeg + KP - 1) / KP;
  if (nsteps <= 0) return;

  gload();
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nsteps; ++ks) {
    const bool more = ks + 1 < nsteps;
    if (more) gload();
    const T* su = smem + cur * BUF;
    const T* sx = su + KP * LDU;
    if constexpr (sizeof(T) == 2) {
      const int grp16 = (lane >> 4) & 1, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#pragma unroll
      for (int kk = 0; kk < KP / 16; ++kk) {
        bf16x8_t fa[FA];
        const int k0 = kk * 16 + 8 * h + q;
#pragma unroll
        for (int f = 0; f < FA; ++f) {
          const int ch = 32 * f + 16 * grp16 + 4 * pp;
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(su + k0 * LDU + ch));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(su + (k0 + 4) * LDU + ch));
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fa[f] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          bf16x8_t fb[FB];
#pragma unroll
          for (int f = 0; f < FB; ++f) {
            const int ch = wci * (BCI / 4) + 32 * f + 16 * grp16 + 4 * pp;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sx + (k0 + s) * LDX + ch));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sx + (k0 + s + 4) * LDX + ch));
            s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fb[f] = __builtin_bit_cast(bf16x8_t, v);
          }
#pragma unroll
          for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j)
              acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[s][i][j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll 2
      for (int k = 0; k < KP; k += 2) {
        float fa[FA];
#pragma unroll
        for (int f = 0; f < FA; ++f) fa[f] = su[(k + h) * LDU + 32 * f + r32];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          float fb[FB];
#pragma unroll
          for (int f = 0; f < FB; ++f) fb[f] = sx[(k + h + s) * LDX + wci * (BCI / 4) + 32 * f + r32];
#pragma unroll
          for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j)
              acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[s][i][j], 0, 0, 0);
        }
      }
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  const long long ldw = 9ll * a.Cin;
  float* const dwp = WG_DET_PTR(a, a.dw, by);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = trow * 3 + s;
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        const int ci = ci0 + wci * (BCI / 4) + 32 * j + r32;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int co = co0 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (co < a.Cout && ci < a.Cin) atomicAdd(dwp + (long long)co * ldw + tap * a.Cin + ci, acc[s][i][j][reg]);
        }
      }
  }
}

template <typename T>
static int wg3_launch(WGArgs& a, hipStream_t st) {
  constexpr int BCO = 64, BCI = 128;
  const int nco_t = (a.Cout + BCO - 1) / BCO;
  a.nci_t = (a.Cin + BCI - 1) / BCI;
  const int ntiles = nco_t * 3 * a.nci_t;
  const double flops = 2.0 * 9 * a.Cin * (double)a.Cout * a.Mpix;
  const long long cap = (long long)(flops * 5.4e-4 / (3.0 * BCO * BCI * 4.0));     // same atomics-traffic budget as wg_launch
  int splits = wg_pick_splits(ntiles, 2304, cap, a.Mpix, &a.ppb);                   // 3 rounds of 768 slots (3 per CU)
  DetPlan plan; float* lo = nullptr; long long span = 0;
  const int granted = wg_det_begin(a, splits, st, &plan, &lo, &span);
  if (granted < 0) return DG_ERR_LAUNCH;
  if (granted != splits) splits = wg_pick_splits(ntiles, (long long)ntiles * granted, 1ll << 40, a.Mpix, &a.ppb);
  constexpr int PADE = sizeof(T) == 2 ? 32 : 0;
  const int lds = 2 * (32 * (BCO + PADE) + 34 * (BCI + PADE)) * (int)sizeof(T);
  if (lds > 65536) DG_SET_MAX_LDS_ONCE((&wg3_kernel<T, BCO>), lds);
  hipLaunchKernelGGL((wg3_kernel<T, BCO>), dim3(ntiles, splits), dim3(256), lds, st, a);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return wg_det_end(a, plan, lo, span, st);
}

// ------------------------------------------------------------------------------------------------------------------
// Wide row-of-taps kernel (bf16): 128 adjoint channels x 128 input channels x the 3 taps of one row per workgroup.
// The weight gradient re-reads its operands through L2 once per (co tile, ci tile, tap row) and removing the global
// loads from wg3_kernel alone buys 30 %, so the lever is operand bytes per flop: this tile moves 5.4 KB per MFLOP against
// 8.1 KB (64 x 128 x 3) and 16 KB (per-tap 128 x 128).  192 accumulator registers leave no room for staging registers:
// the tiles go global -> LDS by DMA (buffer_load ... lds), three buffers, issued two 32-pixel steps ahead, ONE barrier per
// step.  LDS rows are 256 B unpadded (the DMA writes lanes linearly); the 16-byte chunk index is XOR-swizzled with
// (row & 3) << 2 on the SOURCE side, which keeps the four rows of a transposing read in four different 64-B bank groups.
// Wave w owns all 128 adjoint channels x input channels [32w, 32w + 32): per 16 pixels 4 adjoint fragments shared by the
// three taps + 1 input fragment per tap = 7 KB of LDS reads per 12 MFMAs.
template <bool S2>
__global__ __launch_bounds__(256, 2) void wg3w_kernel(const WGArgs a) {
  constexpr int BCO = 128, BCI = 128, KP = 32;
  // x tile: 34 rows (stride 1) / 65 rows (S2: input columns 2wo - 1 .. 2wo + 63, tap s of output pixel k = row 2k + s),
  // rounded up to whole 4-row DMA pieces (the surplus rows are zero-filled and never read)
  constexpr int XROWS = S2 ? 65 : KP + 2, NXP = S2 ? 4 : 2, LASTP = 4 * NXP, XALLOC = 4 * (LASTP + 1);
  constexpr int ROWB = 256, SU_B = KP * ROWB, SX_B = XALLOC * ROWB, BUFB = SU_B + SX_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char wgw_dsm[];            // 3 * BUFB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, by;
  if (!wg_decode(a, bx, by)) return;
  int ci_t = bx % a.nci_t;
  int trow = (bx / a.nci_t) % 3;
  int co_t = bx / (a.nci_t * 3);
  if (a.tri) {                                  // bx = 3 * pair + tap row; pair p = t (t + 1) / 2 + input tile, input tile <= t
    trow = bx % 3;
    const int p = bx / 3;
    co_t = 0;
    while ((co_t + 1) * (co_t + 2) / 2 <= p) ++co_t;
    ci_t = p - co_t * (co_t + 1) / 2;
  }
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int dr = trow - 1;
  const int pbeg = by * a.ppb;
  const int pend = min(a.Mpix, pbeg + a.ppb);
  const int nsteps = (pend - pbeg + KP - 1) / KP;
  if (nsteps <= 0) return;
  const char* X = reinterpret_cast<const char*>(a.x);
  const char* U = reinterpret_cast<const char*>(a.u);

  // DMA lane constants: a piece = 4 rows x 256 B; lane -> (row 4p + lane / 16, physical chunk lane % 16)
  const int drow = lane >> 4, lchunk = (lane & 15) ^ (drow << 2);
  // S2: the rows of a transposing read are 2 apart, so the swizzle key is (row >> 1) & 3 (piece p = wave + 4j: p & 1 = wave & 1)
  const int lcx = S2 ? (lane & 15) ^ (((((wave & 1) << 1) + (drow >> 1)) & 3) << 2) : lchunk;
  const int lcx_last = S2 ? (lane & 15) ^ ((drow >> 1) << 2) : lchunk;
  unsigned uoff[2], xoff[NXP + 1];
  int xr[NXP + 1];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 4 * (wave + 4 * j) + drow, co = co0 + lchunk * 8;
    if (co >= a.Cout) uoff[j] = WG_OOB_OFF;
    else if (!a.u_ps) uoff[j] = (unsigned)(((long long)row * a.ldu + co) * 2);
    else {
      const int cchunk = co / 8, q = cchunk / a.cps_chunks, c = cchunk - q * a.cps_chunks;
      uoff[j] = (unsigned)((((long long)(q >> 1) * (2 * a.Wo) + 2 * row + (q & 1)) * a.ldu + c * 8) * 2);
    }
  }
#pragma unroll
  for (int j = 0; j <= NXP; ++j) {
    const int row = 4 * (j < NXP ? wave + 4 * j : LASTP) + drow, ci = ci0 + (j < NXP ? lcx : lcx_last) * 8;
    xr[j] = row;
    xoff[j] = (row < XROWS && ci < a.Cin) ? (unsigned)(((long long)row * a.ldx + ci) * 2) : WG_OOB_OFF;
  }
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)wgw_dsm));
  typedef int i32x4w_t __attribute__((ext_vector_type(4)));
  auto dma = [&](unsigned m0v, unsigned voff, const i32x4w_t& rs) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rs) : "memory");
  };
  auto make_rs = [&](const char* base) {
    const unsigned long long b = (unsigned long long)base;
    i32x4w_t rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    rs[2] = (int)WG_OOB_OFF;
    rs[3] = 0x00020000;
    return rs;
  };
  int s_wo = pbeg % a.Wo, s_ho, s_n;
  { const int t = pbeg / a.Wo; s_ho = t % a.Ho; s_n = t / a.Ho; }
  // tile t (32 output pixels of one row + the 34 input pixels under them) -> buffer buf
  auto issue = [&](int buf, int t) {
    const long long ub = !a.u_ps ? ((long long)(s_n * a.Ho + s_ho) * a.Wo + s_wo) * a.ldu
                                 : ((long long)(s_n * 2 * a.Ho + 2 * s_ho) * (2 * a.Wo) + 2 * s_wo) * a.ldu;
    const int hi = (S2 ? 2 * s_ho : s_ho) + dr;
    const bool row_ok = (unsigned)hi < (unsigned)a.H;
    const int wi0 = (S2 ? 2 * s_wo : s_wo) - 1;                 // input column of tile row 0
    const long long xb = ((long long)(s_n * a.H + (row_ok ? hi : 0)) * a.W + wi0) * a.ldx;
    const i32x4w_t rsU = make_rs(U + ub * 2), rsX = make_rs(X + xb * 2);
    const unsigned m0b = lds0 + (unsigned)buf * BUFB;
    dma(m0b + (unsigned)wave * 1024u, uoff[0], rsU);
    dma(m0b + (unsigned)(wave + 4) * 1024u, uoff[1], rsU);
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
      const unsigned vo = (row_ok && (unsigned)(wi0 + xr[j]) < (unsigned)a.W) ? xoff[j] : WG_OOB_OFF;
      dma(m0b + SU_B + (unsigned)(wave + 4 * j) * 1024u, vo, rsX);
    }
    if ((t & 3) == wave) {                                      // the last rows (+ zero rows): one wave per step, in turn
      const unsigned vo = (row_ok && (unsigned)(wi0 + xr[NXP]) < (unsigned)a.W) ? xoff[NXP] : WG_OOB_OFF;
      dma(m0b + SU_B + (unsigned)LASTP * 1024u, vo, rsX);
    }
    s_wo += KP;
    if (s_wo >= a.Wo) { s_wo = 0; if (++s_ho == a.Ho) { s_ho = 0; ++s_n; } }
  };

  f32x16_t acc[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s][f][e] = 0.f;

  // fragment read offsets (bytes inside a buffer): row 8h + q (+ tap), 64-B group XOR (row & 3), 32 * grp16 + 8 * pp inside
  const int h = lane >> 5, grp16 = (lane >> 4) & 1, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, r32 = lane & 31;
  const int bu = (8 * h + q) * ROWB + (q << 6) + 32 * grp16 + 8 * pp;          // adjoint fragment f at bu ^ (f << 6)
  int bxs[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int row = S2 ? 2 * (8 * h + q) + s : 8 * h + q + s;
    bxs[s] = SU_B + row * ROWB + ((wave ^ ((S2 ? row >> 1 : row) & 3)) << 6) + 32 * grp16 + 8 * pp;
  }
  constexpr int XK = S2 ? 2 : 1;                                 // tile rows per output pixel

  // bias gradient (column sums of the adjoint) on the side: every workgroup that stages this adjoint tile holds its fragments in
  // registers.  The duty rotates over the K steps among the group of workgroups that share the tile and the pixel range (input
  // tiles x 3 tap rows: step ks belongs to member ks mod group), and inside the workgroup wave w sums fragment w: left to wave 0
  // of ONE workgroup per tile (round 2) those few workgroups ran 10-20 % longer than the rest of their round.  Deterministic mode keeps
  // the single writer (one workgroup per split and tile adds into the split's copy: a fixed order).
  float* const db_out = a.tri ? (a.dbk[co_t] ? WG_DET_PTR(a, a.dbk[co_t], by) : nullptr)
                              : (a.db ? WG_DET_DB(a, by) + co0 : nullptr);         // bias gradient of this adjoint tile's rows
  const bool has_db = db_out != nullptr;
  const int db_group = a.det_ws ? 1 : 3 * (a.tri ? co_t + 1 : a.nci_t);
  const bool db_member = a.det_ws ? (trow == 1 && ci_t == 0) : true;
  int db_cnt = a.det_ws ? 0 : ci_t * 3 + trow;                  // steps until this workgroup's next turn
  float dbacc = 0.f;

  issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  // a wave has NXP + 2 or NXP + 3 pieces per tile in flight: "at most NXP + 2 outstanding" = everything older than the
  // newest tile has landed
  auto wait_older = [&](bool newest_in_flight) {
    if (!newest_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (S2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  wait_older(nsteps > 1);
  __syncthreads();
  int cur = 0;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  for (int ks = 0; ks < nsteps; ++ks) {
    const bool ahead = ks + 2 < nsteps;
    int nb = cur + 2; if (nb >= 3) nb -= 3;
    if (ahead) issue(nb, ks + 2);            // buffer (ks + 2) % 3 was last read in step ks - 1, behind that step's barrier
    const bool db_now = has_db && db_member && db_cnt == 0;
    db_cnt = db_cnt == 0 ? db_group - 1 : db_cnt - 1;
    const unsigned char* sb = wgw_dsm + cur * BUFB;
    // input fragments are read one tap ahead of the MFMAs that use them (two fragment registers sets of 4): the LDS latency
    // of tap s+1 runs under the four MFMAs of tap s instead of in front of its own
    auto read_fb = [&](int s, int kk) {
      const unsigned char* pb = sb + bxs[s] + kk * 16 * XK * ROWB;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb + 4 * XK * ROWB));
      const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8_t, v);
    };
    bf16x8_t fb = read_fb(0, 0);
#pragma unroll
    for (int kk = 0; kk < KP / 16; ++kk) {
      bf16x8_t fa[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const unsigned char* pa = sb + (bu ^ (f << 6)) + kk * 16 * ROWB;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + 4 * ROWB));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fa[f] = __builtin_bit_cast(bf16x8_t, v);
      }
      if (db_now) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (f == wave) {                           // (wave is uniform: a scalar branch; fa[] must be indexed at compile time)
            const u32x4w_t w4 = __builtin_bit_cast(u32x4w_t, fa[f]);
#pragma unroll
            for (int e = 0; e < 4; ++e) dbacc += __uint_as_float(w4[e] << 16) + __uint_as_float(w4[e] & 0xffff0000u);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        bf16x8_t fbn = fb;
        const bool last = s == 2 && kk == KP / 16 - 1;
        __builtin_amdgcn_sched_barrier(0);
        if (!last) fbn = read_fb(s == 2 ? 0 : s + 1, s == 2 ? kk + 1 : kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[s][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f], fb, acc[s][f], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);       // 192 accumulator registers: one input fragment in use, one in flight
        fb = fbn;
      }
    }
    wait_older(ahead);
    __syncthreads();
    if (++cur == 3) cur = 0;
  }

  if (has_db && db_member && co0 + 32 * wave + r32 < a.Cout) atomicAdd(db_out + 32 * wave + r32, dbacc);
  // epilogue: one lane-constant 32-bit offset, everything else of an element's address is workgroup-uniform (scalar base)
  const int cin_w = a.tri ? (co_t + 1) * BCI : a.Cin;                  // input channels of the conv these rows belong to
  const int cow0 = a.tri ? 0 : co0;                                    // first gradient row of this tile inside that conv
  float* const dw_out = WG_DET_PTR(a, a.tri ? a.dwk[co_t] : a.dw, by);
  const long long ldw = 9ll * cin_w;
  const int ci = ci0 + wave * 32 + r32;
  const unsigned lane_off = (unsigned)((((long long)cow0 + 4 * h) * ldw + ci) * 4);
  const bool ci_ok = ci < cin_w;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = trow * 3 + s;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int cor = 32 * f + (reg & 3) + 8 * (reg >> 2);             // + co0 + 4h
        char* const base = reinterpret_cast<char*>(dw_out) + ((long long)cor * ldw + (long long)tap * cin_w) * 4;
        if (ci_ok && co0 + cor + 4 * h < a.Cout) atomicAdd(reinterpret_cast<float*>(base + lane_off), acc[s][f][reg]);
      }
  }
}

template <bool S2>
static int wg3w_launch(WGArgs& a, hipStream_t st) {
  constexpr int BCO = 128, BCI = 128;
  const int nco_t = (a.Cout + BCO - 1) / BCO;
  a.nci_t = (a.Cin + BCI - 1) / BCI;
  const int npairs = a.tri ? nco_t * (nco_t + 1) / 2 : nco_t * a.nci_t;            // dense-block mode: input tile <= adjoint tile
  const int ntiles = 3 * npairs;
  const double flops = 2.0 * 9 * BCO * (double)BCI * npairs * a.Mpix;
  const long long cap = (long long)(flops * 5.4e-4 / (3.0 * BCO * BCI * 4.0));     // same atomics-traffic budget as wg_launch
  int splits = wg_pick_splits(ntiles, 1536, cap, a.Mpix, &a.ppb);                   // 3 rounds of 512 slots (2 per CU)
  DetPlan plan; float* lo = nullptr; long long span = 0;
  const int granted = wg_det_begin(a, splits, st, &plan, &lo, &span);
  if (granted < 0) return DG_ERR_LAUNCH;
  if (granted != splits) splits = wg_pick_splits(ntiles, (long long)ntiles * granted, 1ll << 40, a.Mpix, &a.ppb);
  constexpr int lds = 3 * (32 * 256 + (S2 ? 68 : 36) * 256);
  if (lds > 65536) DG_SET_MAX_LDS_ONCE((&wg3w_kernel<S2>), lds);
  wg_group_order(a, nco_t, ntiles, splits);
  hipLaunchKernelGGL((wg3w_kernel<S2>), a.grp ? dim3(wg_group_blocks(a)) : dim3(ntiles, splits), dim3(256), lds, st, a);
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return wg_det_end(a, plan, lo, span, st);
}

// ------------------------------------------------------------------------------------------------------------------
// fp8 weight gradient (BASELINE configs[4]): the wide row-of-taps kernel on v_mfma_scale_f32_32x32x64_f8f6f4.
// The contraction runs over PIXELS, so the MXFP8 forms of the conv path (one scale per pixel and 32 channels) cannot feed it:
// a K block of 32 pixels would need one scale.  The operands here are E4M3 tensors with a scale that does not vary along
// pixels -- one E8M0 exponent per 32-channel block of the whole tensor (`ex` / `eu`; x = q * 2^(e - 127)) -- so every window
// of 64 pixels is a valid K step for all three taps of a kernel row, and the two per-block scales of a fragment pair go into
// the MFMA's scale operands (each lane supplies its row's / column's exponent; both K blocks get the same one).
// Register maps measured with tools/fp8_probe3.hip: A / B lane l = row / column l % 32, 32 bytes; byte j of lane half
// l / 32 is the same k for A and B (so any assignment of the 64 pixels to (half, byte) works as long as both operands use
// it); D as the bf16 32x32 shapes.  ds_read_b64_tr_b8: in a group of 16 lanes, lane 8p + i receives byte i of the eight
// 8-byte rows supplied by lanes 2j + p (j = 0..7): source lanes 2j / 2j + 1 point at channels c .. c+7 / c+8 .. c+15 of
// pixel j, and the 16 lanes of the group end up with 16 consecutive channels x 8 consecutive pixels.
// Tile and schedule as wg3w_kernel (128 adjoint x 128 input channels x 3 taps, DMA-staged three-buffer ring, one barrier per
// step) with 64-pixel steps: tile rows are 128 B, a DMA piece = 8 rows, LDS 3 x 17 KB.  The 32-byte chunk index of a row is
// XOR-ed with (row >> 1) & 3 on the source side of the DMA: the eight rows of a transposing read then cover all 64 banks.
// Strides 1 and 2, channel counts that are multiples of 128, output rows of a multiple of 64 pixels.
typedef __attribute__((ext_vector_type(8))) int i32x8w_t;
typedef __attribute__((ext_vector_type(2))) int i32x2w_t;
typedef __attribute__((address_space(3))) i32x2w_t* lds_i32x2_ptr;

template <bool S2>
__global__ __launch_bounds__(256, 2) void wg3w_f8_kernel(const WGArgs a) {
  constexpr int BCO = 128, BCI = 128, KP = 64, ROWB = 128;
  // x tile: KP + 2 rows (stride 1) / 2 KP + 1 rows (S2: input columns 2wo - 1 .. 2wo + 127, tap s of output pixel k = row 2k + s) in
  // DMA pieces of 8 rows.  S2: the eight rows of a transposing read are two apart, i.e. all in the same 128-byte half of the 256-byte
  // bank span -- consecutive pieces are therefore skewed by one row (piece pitch 1152 B): the rows of a read that fall into the next
  // piece land in the other half, and the 32-byte chunk XOR separates the (at most four) rows inside a half.
  constexpr int XK = S2 ? 2 : 1, XROWS = S2 ? 2 * KP + 1 : KP + 2, NXW = S2 ? 4 : 2, XPIECES = 4 * NXW + 1, XPITCH = S2 ? 1152 : 1024;
  constexpr int SU_B = KP * ROWB, SX_B = XPIECES * XPITCH, BUFB = SU_B + SX_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char wg8_dsm[];          // 3 * BUFB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx, by;
  if (!wg_decode(a, bx, by)) return;
  int ci_t = bx % a.nci_t, trow = (bx / a.nci_t) % 3, co_t = bx / (a.nci_t * 3);
  if (a.tri) {                                  // dense-block mode as in wg3w_kernel: bx = 3 * pair + tap row, input tile <= adjoint tile
    trow = bx % 3;
    const int p = bx / 3;
    co_t = 0;
    while ((co_t + 1) * (co_t + 2) / 2 <= p) ++co_t;
    ci_t = p - co_t * (co_t + 1) / 2;
  }
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int dr = trow - 1;
  const int pbeg = by * a.ppb;
  const int pend = min(a.Mpix, pbeg + a.ppb);
  const int nsteps = (pend - pbeg + KP - 1) / KP;
  if (nsteps <= 0) return;
  const char* X = reinterpret_cast<const char*>(a.x);
  const char* U = reinterpret_cast<const char*>(a.u);

  // DMA lane constants: a piece = 8 rows x 128 B; lane -> (row 8p + lane / 8, physical 16-byte chunk lane % 8), which holds the
  // logical chunk with the 32-byte index XOR-ed by (row >> 1) & 3 (8p does not reach those bits).  ONE lane offset per tensor
  // (row-in-piece and chunk); the piece's first row goes into the instruction's scalar offset, the channel tile into the base --
  // per-piece lane offsets (8 registers) made the 192 + 32 accumulator / fragment registers spill inside the loop
  const int drow = lane >> 3, pchunk = lane & 7;
  const int lchunk = ((((pchunk >> 1) ^ ((drow >> 1) & 3)) << 1) | (pchunk & 1)) * 16;          // byte offset inside the 128-byte row
  const unsigned ulane = (unsigned)(drow * (int)a.ldu + lchunk), xlane = (unsigned)(drow * (int)a.ldx + lchunk);
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)wg8_dsm));
  typedef int i32x4w_t __attribute__((ext_vector_type(4)));
  auto dma = [&](unsigned m0v, unsigned voff, const i32x4w_t& rs, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory");
  };
  auto make_rs = [&](const char* base) {
    const unsigned long long b = (unsigned long long)base;
    i32x4w_t rs;
    rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    rs[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    rs[2] = (int)WG_OOB_OFF;
    rs[3] = 0x00020000;
    return rs;
  };
  int s_wo = pbeg % a.Wo, s_ho, s_n;
  { const int t = pbeg / a.Wo; s_ho = t % a.Ho; s_n = t / a.Ho; }
  // tile t (64 output pixels of one row + the input pixels under them) -> buffer buf.  A lane whose input pixel lies outside
  // the image gets the out-of-range offset (the range check looks at the lane offset only: the load returns zeros).
  auto issue = [&](int buf, int t) {
    const long long ub = ((long long)(s_n * a.Ho + s_ho) * a.Wo + s_wo) * a.ldu + co0;
    const int hi = XK * s_ho + dr;
    const bool row_ok = (unsigned)hi < (unsigned)a.H;
    const int wi0 = XK * s_wo - 1;                              // input column of tile row 0
    const long long xb = ((long long)(s_n * a.H + (row_ok ? hi : 0)) * a.W + wi0) * a.ldx + ci0;
    const i32x4w_t rsU = make_rs(U + ub), rsX = make_rs(X + xb);
    const unsigned m0b = lds0 + (unsigned)buf * BUFB;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma(m0b + (unsigned)(wave + 4 * j) * 1024u, ulane, rsU, (unsigned)((wave + 4 * j) * 8 * (int)a.ldu));
#pragma unroll
    for (int j = 0; j < NXW; ++j) {
      const int row0 = 8 * (wave + 4 * j);
      const unsigned vo = (row_ok && (unsigned)(wi0 + row0 + drow) < (unsigned)a.W) ? xlane : WG_OOB_OFF;
      dma(m0b + SU_B + (unsigned)(wave + 4 * j) * (unsigned)XPITCH, vo, rsX, (unsigned)(row0 * (int)a.ldx));
    }
    if ((t & 3) == wave) {                                      // the last piece (2 / 1 of its rows are read): one wave per step, in turn
      constexpr int R0 = 8 * (XPIECES - 1);
      const unsigned vo = (row_ok && drow < XROWS - R0 && (unsigned)(wi0 + R0 + drow) < (unsigned)a.W) ? xlane : WG_OOB_OFF;
      dma(m0b + SU_B + (unsigned)(XPIECES - 1) * (unsigned)XPITCH, vo, rsX, (unsigned)(R0 * (int)a.ldx));
    }
    s_wo += KP;
    if (s_wo >= a.Wo) { s_wo = 0; if (++s_ho == a.Ho) { s_ho = 0; ++s_n; } }
  };

  f32x16_t acc[3][4];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[s][f][e] = 0.f;

  // fragment read offsets inside a buffer.  Lane l: half h = l / 32 (pixels 32h ..), channel group G = (l / 16) & 1, source role
  // sl = l % 16: pixel row sl / 2 of the eight, channels + 8 * (sl & 1).  Read q of a fragment covers pixels 32h + 8q .. + 7.
  const int h = lane >> 5, G = (lane >> 4) & 1, sl = lane & 15, prow = sl >> 1, low = 16 * G + 8 * (sl & 1), r32 = lane & 31;
  const int bu = (32 * h + prow) * ROWB + (((prow >> 1) & 3) << 5) + low;            // adjoint fragment f, read q: (bu ^ (f << 5)) + q * 8 * ROWB
  int bxs[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int row = XK * (32 * h + prow) + s;                  // tile row of this lane's pixel for read q = 0 (read q: + 8 XK q rows)
    bxs[s] = SU_B + row * ROWB + (S2 ? (row >> 3) * (XPITCH - 1024) : 0) + ((wave ^ ((row >> 1) & 3)) << 5) + low;
  }
  constexpr int XQ = 8 * XK * ROWB + (S2 ? 2 * (XPITCH - 1024) : 0);      // byte step between the reads q, q + 1 of an input fragment
  // exponent bytes: adjoint fragment f = 32-channel block co0 / 32 + f, this wave's input columns = block ci0 / 32 + wave
  const unsigned eu4 = *reinterpret_cast<const unsigned*>(a.eu + (co0 >> 5));
  const int sxb = (int)a.ex[(ci0 >> 5) + wave];

  issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  auto wait_older = [&](bool newest_in_flight) {        // a tile = NXW + 2 or NXW + 3 pieces per wave: everything older than the newest has landed
    if (!newest_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (S2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  wait_older(nsteps > 1);
  __syncthreads();
  int cur = 0;
  for (int ks = 0; ks < nsteps; ++ks) {
    const bool ahead = ks + 2 < nsteps;
    int nb = cur + 2; if (nb >= 3) nb -= 3;
    if (ahead) issue(nb, ks + 2);            // buffer (ks + 2) % 3 was last read in step ks - 1, behind that step's barrier
    const unsigned char* sb = wg8_dsm + cur * BUFB;
    auto read_frag = [&](int off, int qstep) {
      i32x8w_t v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const i32x2w_t t = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2_ptr)(sb + off + q * qstep));
        v[2 * q] = t[0]; v[2 * q + 1] = t[1];
      }
      return v;
    };
    i32x8w_t fb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) fb[s] = read_frag(bxs[s], XQ);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      // 192 accumulator + 24 input-fragment registers leave room for ONE adjoint fragment: its read latency is in the open (the
      // second workgroup of the CU covers it); a second one in flight spilled 13 registers
      __builtin_amdgcn_sched_barrier(0);
      const i32x8w_t fa = read_frag(bu ^ (f << 5), 8 * ROWB);
      __builtin_amdgcn_sched_barrier(0);
      const int sua = (int)((eu4 >> (8 * f)) & 0xffu);
#pragma unroll
      for (int s = 0; s < 3; ++s)
        acc[s][f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[s], acc[s][f], 0, 0, 0, sua, 0, sxb);
      // (the fence keeps the next fragment's reads behind these MFMAs: hoisted, all four adjoint fragments are live at once)
      asm volatile("" : "+v"(acc[0][f]), "+v"(acc[1][f]), "+v"(acc[2][f]) :: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    wait_older(ahead);
    __syncthreads();
    if (++cur == 3) cur = 0;
  }

  // epilogue (wg3w_kernel's): one lane-constant 32-bit offset, the rest of an element's address is workgroup-uniform
  const int cin_w = a.tri ? (co_t + 1) * BCI : a.Cin;                  // input channels of the conv these rows belong to
  const int cow0 = a.tri ? 0 : co0;                                    // first gradient row of this tile inside that conv
  float* const dw_out = WG_DET_PTR(a, a.tri ? a.dwk[co_t] : a.dw, by);
  const long long ldw = 9ll * cin_w;
  const int ci = ci0 + wave * 32 + r32;
  const unsigned lane_off = (unsigned)((((long long)cow0 + 4 * h) * ldw + ci) * 4);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = trow * 3 + s;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int cor = 32 * f + (reg & 3) + 8 * (reg >> 2);             // + co0 + 4h
        char* const base = reinterpret_cast<char*>(dw_out) + ((long long)cor * ldw + (long long)tap * cin_w) * 4;
        atomicAdd(reinterpret_cast<float*>(base + lane_off), acc[s][f][reg]);
      }
  }
}

// launch of the fp8 kernel for a prepared WGArgs (x / u / ex / eu / geometry / Mpix set; tri + dwk[] in dense-block mode)
static int wg3w_f8_launch(WGArgs& a, hipStream_t st) {
  constexpr int BCO = 128, BCI = 128;
  const int nco_t = a.Cout / BCO;
  a.nci_t = a.Cin / BCI;
  const int npairs = a.tri ? nco_t * (nco_t + 1) / 2 : nco_t * a.nci_t;
  const int ntiles = 3 * npairs;
  const double flops = 2.0 * 9 * BCO * (double)BCI * npairs * a.Mpix;
  const long long cap = (long long)(flops * 5.4e-4 / (3.0 * BCO * BCI * 4.0));     // the atomics-traffic budget of the bf16 launchers
  int splits = wg_pick_splits(ntiles, 1536, cap, a.Mpix, &a.ppb);                   // 3 rounds of 512 slots (2 per CU)
  DetPlan plan; float* lo = nullptr; long long span = 0;
  const int granted = wg_det_begin(a, splits, st, &plan, &lo, &span);
  if (granted < 0) return DG_ERR_LAUNCH;
  if (granted != splits) splits = wg_pick_splits(ntiles, (long long)ntiles * granted, 1ll << 40, a.Mpix, &a.ppb);
  wg_group_order(a, nco_t, ntiles, splits);
  const dim3 grid = a.grp ? dim3(wg_group_blocks(a)) : dim3(ntiles, splits);
  if (a.stride == 2) {
    constexpr int lds2 = 3 * (64 * 128 + 17 * 1152);
    DG_SET_MAX_LDS_ONCE((&wg3w_f8_kernel<true>), lds2);
    hipLaunchKernelGGL(wg3w_f8_kernel<true>, grid, dim3(256), lds2, st, a);
  } else {
    constexpr int lds = 3 * (64 * 128 + 9 * 1024);
    hipLaunchKernelGGL(wg3w_f8_kernel<false>, grid, dim3(256), lds, st, a);
  }
  if (dg_check_launch() != DG_OK) return DG_ERR_LAUNCH;
  return wg_det_end(a, plan, lo, span, st);
}

// dw[co][tap][ci] (fp32, accumulated into) += sum_p dy[p, co] * x[src(p, tap), ci] for E4M3 operands with per-32-channel-block
// exponents: x = xq * 2^(ex[ci / 32] - 127), dy = dyq * 2^(ey[co / 32] - 127) (`ldx` / `ldy` of g = pixel strides in bytes).
extern "C" int dg_conv3x3_wgrad_f8(const dg_conv_geom* g, const void* xq, const void* ex, const void* dyq, const void* ey, float* dw,
                                   void* stream) {
  if (!g || !xq || !ex || !dyq || !ey || !dw) return DG_ERR_BAD_ARG;
  if (g->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;               // (the dtype of the tensors the fp8 forms stand for)
  if (g->N <= 0 || g->H <= 0 || g->W <= 0 || (g->stride != 1 && g->stride != 2) || g->pixel_shuffle) return DG_ERR_BAD_SHAPE;
  if (g->stride == 2 && ((g->H | g->W) & 1)) return DG_ERR_BAD_SHAPE;
  if (g->Cin <= 0 || g->Cout <= 0 || g->Cin % 128 || g->Cout % 128 || (g->W / g->stride) % 64) return DG_ERR_BAD_SHAPE;
  if (g->ldx < g->Cin || g->ldy < g->Cout || g->ldx % 16 || g->ldy % 16) return DG_ERR_BAD_SHAPE;
  WGArgs a{};
  a.x = xq; a.u = dyq; a.dw = dw; a.ldx = g->ldx; a.ldu = g->ldy;
  a.ex = (const unsigned char*)ex; a.eu = (const unsigned char*)ey;
  a.H = g->H; a.W = g->W; a.stride = g->stride; a.Ho = g->H / g->stride; a.Wo = g->W / g->stride;
  a.Cin = g->Cin; a.Cout = g->Cout; a.u_ps = 0; a.cps_chunks = 1;
  const long long mp = (long long)g->N * a.Ho * a.Wo;
  if (mp >= (1ll << 31) || (long long)(a.W + 8) * a.ldx >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  a.Mpix = (int)mp;
  return wg3w_f8_launch(a, reinterpret_cast<hipStream_t>(stream));
}

// The dense-block launch (dg_conv3x3_wgrad_dense) on the fp8 kernel: xq / dyq = uniform-scale E4M3 copies of the activation and
// adjoint slabs [N, H, W, nconv * 128] (pixel strides g->ldx / g->ldy in bytes), ex / ey = their nconv * 4 block exponents; weight
// gradients only (the bias gradients are column sums of the adjoint: dg_colsum).  W % 64 == 0.
extern "C" int dg_conv3x3_wgrad_dense_f8(const dg_conv_geom* g, int nconv, const void* xq, const void* ex, const void* dyq, const void* ey,
                                         float* const* dw, void* stream) {
  if (!g || !xq || !ex || !dyq || !ey || !dw) return DG_ERR_BAD_ARG;
  if (g->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  if (nconv < 1 || nconv > 8 || g->N <= 0 || g->H <= 0 || g->W <= 0 || g->stride != 1 || g->pixel_shuffle || g->W % 64) return DG_ERR_BAD_SHAPE;
  if (g->Cin != nconv * 128 || g->Cout != nconv * 128 || g->ldx % 16 || g->ldy % 16 || g->ldx < g->Cin || g->ldy < g->Cout) return DG_ERR_BAD_SHAPE;
  for (int k = 0; k < nconv; ++k)
    if (!dw[k]) return DG_ERR_BAD_ARG;
  WGArgs a{};
  a.x = xq; a.u = dyq; a.dw = dw[0]; a.ldx = g->ldx; a.ldu = g->ldy;
  a.ex = (const unsigned char*)ex; a.eu = (const unsigned char*)ey;
  a.H = g->H; a.W = g->W; a.stride = 1; a.Ho = g->H; a.Wo = g->W;
  a.Cin = g->Cin; a.Cout = g->Cout; a.u_ps = 0; a.cps_chunks = 1;
  const long long mp = (long long)g->N * a.Ho * a.Wo;
  if (mp >= (1ll << 31) || (long long)(a.W + 8) * a.ldx >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  a.Mpix = (int)mp;
  a.tri = 1;
  for (int k = 0; k < nconv; ++k) { a.dwk[k] = dw[k]; a.dbk[k] = nullptr; }
  return wg3w_f8_launch(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dg_colsum(int dtype, const void* dy, int64_t rows_outer, int64_t ld_outer, int64_t rows_inner, int64_t ld, int C,
                         float* db, void* stream);

extern "C" int dg_conv3x3_wgrad(const dg_conv_geom* g, const void* x, const void* dy, float* dw, float* db, void* stream) {
  if (!g || !x || !dy || !dw) return DG_ERR_BAD_ARG;
  if (g->dtype != DG_F32 && g->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  if (g->N <= 0 || g->H <= 0 || g->W <= 0 || (g->stride != 1 && g->stride != 2)) return DG_ERR_BAD_SHAPE;
  if (g->stride == 2 && ((g->H | g->W) & 1)) return DG_ERR_BAD_SHAPE;
  if (g->Cin % 8 || g->Cout % 16 || g->Cin <= 0 || g->Cout <= 0) return DG_ERR_BAD_SHAPE;
  if (g->pixel_shuffle && (g->stride != 1 || (g->Cout / 4) % 16)) return DG_ERR_BAD_SHAPE;
  const int epc = g->dtype == DG_F32 ? 4 : 8;
  // the im2col kernel (<= 2 real input channels) gathers single (channel 0, channel 1) pairs, so its x may be the COMPACT
  // tensor [N, H, W, 2] (ldx = 2); every other kernel reads whole 16-byte channel chunks
  const bool im2col_shape = g->cin_real > 0 && g->cin_real <= 2 && g->stride == 1 && !g->pixel_shuffle && (g->W / g->stride) % 32 == 0 &&
                            getenv("DG_WG_NOIM2COL") == nullptr;
  if ((im2col_shape ? (g->ldx % 2 || g->ldx < 2) : (g->ldx % epc != 0)) || g->ldy % epc) return DG_ERR_BAD_SHAPE;
  WGArgs a{};
  a.x = x; a.u = dy; a.dw = dw; a.ldx = g->ldx; a.ldu = g->ldy;
  a.H = g->H; a.W = g->W; a.stride = g->stride; a.Ho = g->H / g->stride; a.Wo = g->W / g->stride;
  a.Cin = g->Cin; a.Cout = g->Cout;
  a.u_ps = g->pixel_shuffle; a.cps_chunks = g->pixel_shuffle ? g->Cout / 4 / epc : 1;
  const long long mp = (long long)g->N * a.Ho * a.Wo;
  if (mp >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  a.Mpix = (int)mp;
  a.db = db;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  static const bool no_im2col = getenv("DG_WG_NOIM2COL") != nullptr;
  if (!no_im2col && g->cin_real > 0 && g->cin_real <= 2 && g->stride == 1 && !g->pixel_shuffle && a.Wo % 32 == 0)
    return g->dtype == DG_F32 ? wg_launch_im2col<float>(a, st) : wg_launch_im2col<bf16_t>(a, st);
  // narrow row-of-taps kernel (wg3) against the per-tap kernel: +4-10 % with >= 2 input-channel tiles or multi-megapixel
  // batches, -20 % on the 128-channel 128^2 layers; the wide kernel (wg3w) takes every layer it is eligible for
  constexpr bool no_rows = false, no_wide = false, no_wide_s2 = false;
  const bool wide = g->dtype == DG_BF16 && a.Cout >= 128 && a.Cin >= 128;
  // (the -20 % of the row-of-taps shape on 128-channel 128^2 layers was the narrow wg3 kernel with the old split rounding: the
  // wide kernel is +19-37 % there too)
  const bool rows = !no_rows && g->stride == 1 && a.Wo % 32 == 0 && a.Cout >= 64 &&
                    (a.Cin >= 256 || (a.Cin >= 64 && a.Mpix >= (1 << 22)) || (wide && !no_wide));
  const bool wide_s1 = rows && wide && !no_wide;
  // stride 2 on the wide kernel (x tile of 65 input pixels per 32 output pixels)
  const bool wide_s2 = !no_rows && !no_wide_s2 && wide && g->stride == 2 && a.Wo % 32 == 0 && !g->pixel_shuffle;
  constexpr bool no_fused_db = false;
  const bool fused_db = db && (wide_s1 || wide_s2) && !g->pixel_shuffle && !no_fused_db;   // the wide kernel sums the adjoint itself
  if (db && !fused_db) {   // bias gradient as a separate column-sum pass over the adjoint
    if (g->pixel_shuffle) return DG_ERR_BAD_ARG;
    int rc = dg_colsum(g->dtype, dy, mp, g->ldy, 1, g->ldy, g->Cout, db, stream);
    if (rc) return rc;
  }
  a.db = fused_db ? db : nullptr;
  if (wide_s1) return wg3w_launch<false>(a, st);
  if (rows) return g->dtype == DG_F32 ? wg3_launch<float>(a, st) : wg3_launch<bf16_t>(a, st);
  if (wide_s2) return wg3w_launch<true>(a, st);
  return g->dtype == DG_F32 ? wg_launch<float>(a, st) : wg_launch<bf16_t>(a, st);
}

// Weight (and bias) gradients of ALL convs of a dense block in one launch (generator.py:14-41: conv k reads slab channels
// [0, k * 128) and writes 128 channels).  x = the block's activation slab [N, H, W, nconv * 128], dy = its adjoint slab (channels
// [(k - 1) * 128, k * 128) = the adjoint of conv k's output); dw[k - 1] / db[k - 1] = conv k's fp32 gradients ([128][9][k * 128] /
// [128], accumulated into).  Five separate launches give the first convs 3 tiles each: their split-K leaves 196 KB of atomics per
// 100 MFLOP (766-875 TFLOP/s); the 15 (adjoint tile, input tile) pairs of a block in one grid have 15x longer pixel ranges per
// workgroup.  bf16, stride 1, 128 channels per conv, rows of a multiple of 32 pixels.
extern "C" int dg_conv3x3_wgrad_dense(const dg_conv_geom* g, int nconv, const void* x, const void* dy, float* const* dw,
                                      float* const* db, void* stream) {
  if (!g || !x || !dy || !dw) return DG_ERR_BAD_ARG;
  if (g->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  if (nconv < 1 || nconv > 8 || g->N <= 0 || g->H <= 0 || g->W <= 0 || g->stride != 1 || g->pixel_shuffle || g->W % 32) return DG_ERR_BAD_SHAPE;
  if (g->Cin != nconv * 128 || g->Cout != nconv * 128 || g->ldx % 8 || g->ldy % 8 || g->ldx < g->Cin || g->ldy < g->Cout) return DG_ERR_BAD_SHAPE;
  for (int k = 0; k < nconv; ++k)
    if (!dw[k]) return DG_ERR_BAD_ARG;
  WGArgs a{};
  a.x = x; a.u = dy; a.dw = dw[0]; a.ldx = g->ldx; a.ldu = g->ldy;
  a.H = g->H; a.W = g->W; a.stride = 1; a.Ho = g->H; a.Wo = g->W;
  a.Cin = g->Cin; a.Cout = g->Cout;
  a.u_ps = 0; a.cps_chunks = 1;
  const long long mp = (long long)g->N * a.Ho * a.Wo;
  if (mp >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  a.Mpix = (int)mp;
  a.db = nullptr;
  a.tri = 1;
  for (int k = 0; k < nconv; ++k) { a.dwk[k] = dw[k]; a.dbk[k] = db ? db[k] : nullptr; }
  return wg3w_launch<false>(a, reinterpret_cast<hipStream_t>(stream));
}
