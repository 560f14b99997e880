// Host side of the conv family: lowering of a 3x3 conv layer to gather-GEMM descriptors (dg_conv3x3_plan, no GPU needed),
// the choice of kernel for a descriptor, and the C ABI entry points.
#include "gg_common.h"

thread_local int g_last_kinds = 0;

// ------------------------------------------------------------------------------------ host side
static int gg_validate(const dg_gg_desc* d, bool f8 = false, bool compact_src = false) {
  if (d->dtype != DG_F32 && d->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  const int epc = f8 ? 16 : d->dtype == DG_F32 ? 4 : 8;
  if (d->N <= 0 || d->Hs <= 0 || d->Ws <= 0 || d->Hg <= 0 || d->Wg <= 0 || d->Hd <= 0 || d->Wd <= 0) return DG_ERR_BAD_SHAPE;
  if (d->Cred <= 0 || d->Cred % 8 || d->Nout <= 0 || d->Nout % 16) return DG_ERR_BAD_SHAPE;
  if (d->ntaps < 1 || d->ntaps > 9) return DG_ERR_BAD_SHAPE;
  // compact_src: the im2col kernel gathers single (channel 0, channel 1) pairs, so its source may be a tensor that stores the
  // real channels only (pixel stride 2: one dword in bf16) instead of the 16-channel padded form -- 8x fewer cache lines per gather
  if ((compact_src ? d->lds % 2 || d->lds < 2 : d->lds % epc) || d->ldd % 4 || d->ldw % epc) return DG_ERR_BAD_SHAPE;
  for (int t = 0; t < d->ntaps; ++t) {
    if (d->tap_dy[t] < -1 || d->tap_dy[t] > 1 || d->tap_dx[t] < -1 || d->tap_dx[t] > 1) return DG_ERR_BAD_ARG;
    if (d->tap_w[t] < 0 || d->tap_w[t] > 8) return DG_ERR_BAD_ARG;
  }
  if (d->src_ps && ((d->Cred / 4) % 8)) return DG_ERR_BAD_SHAPE;
  if (d->dst_ps && ((d->Nout / 4) % 16)) return DG_ERR_BAD_SHAPE;
  if ((long long)d->N * d->Hg * d->Wg >= (1ll << 31)) return DG_ERR_BAD_SHAPE;
  // every destination pixel must be inside the destination tensor
  if (d->dst_ps) {
    if (2 * d->Hg > d->Hd || 2 * d->Wg > d->Wd) return DG_ERR_BAD_SHAPE;
  } else {
    if ((d->Hg - 1) * d->dy_mul + d->dy_off >= d->Hd || (d->Wg - 1) * d->dx_mul + d->dx_off >= d->Wd) return DG_ERR_BAD_SHAPE;
    if (d->dy_off < 0 || d->dx_off < 0 || d->dy_mul < 1 || d->dx_mul < 1) return DG_ERR_BAD_SHAPE;
  }
  return DG_OK;
}

// The halo kernel forms its patch offsets with 24-bit multiplies of the row step (bytes between two patch rows: 1 / 2 / 4 source
// rows for the plain / stride-2 / pixel-shuffled gathers).  Decided HERE, on the descriptor as planned: the row-tiled fallback knows
// neither the plane-regrouped tap codes of the stride-2 forward nor the merged parity classes of a stride-2 data gradient.
bool gg_halo_row_step_ok(long long Ws, long long ldx, int dtype, int mult) {
  return mult * Ws * ldx * (dtype == DG_F32 ? 4 : 2) < (1ll << 23);
}

// Kernel choice for one descriptor (bf16 / fp32).
static int gg_launch(GGArgs& a, int dtype, int N, hipStream_t st) {
  if (a.seg) {     // merged classes (seg_dgrad_desc has checked shape and row step): the halo kernel is the only one that knows them
    if (!gg_halo_row_step_ok(a.Ws, a.ldx, dtype, 1)) return DG_ERR_BAD_SHAPE;
    return gg_launch_halo(a, dtype, N, false, false, 4, st);
  }
  // halo kernel: the patch is sized for tap shifts in [-1, 1] around a unit-stride grid (stride-1 forward, all data gradients;
  // also single-tap launches: the 1-tap parity class of a stride-2 data gradient)
  if (a.sy_mul == 1 && a.sx_mul == 1 && !a.src_ps && a.cch % 8 == 0 && a.Nout > 64 && a.Hg >= 8 && a.Wg >= 8 && a.Hs == a.Hg && a.Ws == a.Wg &&
      gg_halo_row_step_ok(a.Ws, a.ldx, dtype, 1))
    return gg_launch_halo(a, dtype, N, false, false, 4, st);
  // pixel-shuffled sources (data gradients of the up-sampling convs) whose channel quarters hold whole 64-channel blocks
  if (a.sy_mul == 1 && a.sx_mul == 1 && a.src_ps && a.cps_src_chunks % 8 == 0 && a.cch % 8 == 0 && a.Nout > 64 &&
      a.Hg >= 8 && a.Wg >= 8 && a.ntaps >= 2 && a.Hs == a.Hg && a.Ws == a.Wg && gg_halo_row_step_ok(a.Ws, a.ldx, dtype, 4))
    return gg_launch_halo(a, dtype, N, false, true, 4, st);
  // stride-2 forward through the four parity planes of the input
  if (a.sy_mul == 2 && a.sx_mul == 2 && !a.src_ps && a.cch % 16 == 0 && a.Nout > 64 && a.Hg >= 8 &&
      a.Wg >= 8 && a.Hs == 2 * a.Hg && a.Ws == 2 * a.Wg && a.dy_mul == 1 && a.dx_mul == 1 && gg_halo_row_step_ok(a.Ws, a.ldx, dtype, 2))
  {
    GGArgs b = a;
    if (gg_regroup_taps_by_plane(b)) {
      // >= 256 output channels (bf16): eight waves share one parity-plane patch between two 128-channel halves
      return gg_launch_halo(b, dtype, N, true, false, (dtype == DG_BF16 && b.Nout % 256 == 0) ? 8 : 4, st);
    }
  }
  if (a.Nout <= 16 && a.sy_mul == 1 && a.sx_mul == 1 && !a.src_ps && !a.dst_ps && a.cch % 8 == 0 && a.Hg >= 8 &&
      a.Wg >= 8 && a.ntaps >= 2 && a.Hs == a.Hg && a.Ws == a.Wg && !a.mask_bits && !a.out_bits)
    return gg_launch_halo16(a, dtype, N, st);
  return gg_launch_rows(a, dtype, st);
}

// fp8 launches: only the shapes the four-wave halo kernel takes (the critic's wide layers); everything else is refused
static int gg_launch_f8(GGArgs& a, const F8Args& f, int N, hipStream_t st) {
  if (a.cch % 8 || a.Nout <= 64 || a.Hg < 8 || a.Wg < 8 || a.src_ps) return DG_ERR_BAD_SHAPE;    // (a pixel-shuffled DESTINATION is the epilogue's business)
  if (a.sy_mul == 1 && a.sx_mul == 1 && a.Hs == a.Hg && a.Ws == a.Wg) return gg_launch_halo_f8(a, f, N, false, 4, st);
  if (a.sy_mul == 2 && a.sx_mul == 2 && a.Hs == 2 * a.Hg && a.Ws == 2 * a.Wg && a.dy_mul == 1 && a.dx_mul == 1 && !a.dst_ps) {
    GGArgs b = a;
    if (gg_regroup_taps_by_plane(b)) {
      return gg_launch_halo_f8(b, f, N, true, b.Nout % 256 == 0 ? 8 : 4, st);
    }
  }
  return DG_ERR_BAD_SHAPE;
}
static int gather_gemm_impl(const dg_gg_desc* d, const dg_epilogue* ep, const void* x, const void* w, void* y, void* stream,
                            bool im2col_small, const dg_f8_operands* f8 = nullptr, bool seg = false) {
  if (!d || !x || !w || !y) return DG_ERR_BAD_ARG;
  if (f8 && (d->dtype != DG_BF16 || !f8->xs || !f8->ws || d->Cred % 128)) return DG_ERR_BAD_SHAPE;
  int rc = gg_validate(d, f8 != nullptr, im2col_small);
  if (rc) return rc;
  const int epc = f8 ? 16 : d->dtype == DG_F32 ? 4 : 8;
  GGArgs a{};
  a.x = x; a.w = w; a.y = y;
  a.ldx = d->lds; a.ldw = d->ldw; a.ldy = d->ldd;
  a.M = d->N * d->Hg * d->Wg; a.Hg = d->Hg; a.Wg = d->Wg; a.Hs = d->Hs; a.Ws = d->Ws;
  a.Cred = d->Cred; a.cch = d->Cred / epc; a.ntaps = d->ntaps; a.kchunks = d->ntaps * a.cch;
  a.sy_mul = d->sy_mul; a.sx_mul = d->sx_mul;
  a.tap_lo = 0; a.tap_hi = 0;
  for (int t = 0; t < d->ntaps; ++t) {
    unsigned long long code = (unsigned)(d->tap_dy[t] + 1) | ((unsigned)(d->tap_dx[t] + 1) << 2) | ((unsigned)d->tap_w[t] << 4);
    if (t < 8) a.tap_lo |= code << (8 * t); else a.tap_hi = (unsigned)code;
  }
  a.Nout = d->Nout; a.Hd = d->Hd; a.Wd = d->Wd;
  a.dy_mul = d->dy_mul; a.dx_mul = d->dx_mul; a.dy_off = d->dy_off; a.dx_off = d->dx_off;
  a.src_ps = d->src_ps; a.dst_ps = d->dst_ps;
  a.cps_src_chunks = d->src_ps ? d->Cred / 4 / epc : 1;
  a.cps_dst = d->dst_ps ? d->Nout / 4 : d->Nout;
  a.s1 = a.s2 = 1.f; a.mask_slope = 1.f; a.act_slope = 1.f;
  if (ep) {
    a.bias = ep->bias; a.has_act = ep->has_act; a.act_slope = ep->act_slope;
    a.r1 = ep->r1; a.ldr1 = ep->ldr1; a.s1 = ep->s1;
    a.r2 = ep->r2; a.ldr2 = ep->ldr2; a.s2 = ep->s2;
    a.mask = ep->mask; a.ldmask = ep->ldmask; a.mask_slope = ep->mask_slope;
    a.mask_c0 = ep->mask_c0; a.mask_last = ep->mask_last;
    if (a.mask_c0 < 0 || a.mask_c0 % 16 || ((a.mask_c0 || a.mask_last) && !a.mask)) return DG_ERR_BAD_ARG;
    a.accumulate = ep->accumulate;
    a.mask_bits = ep->mask_bits; a.out_bits = ep->out_bits;
    a.out_q = ep->out_q; a.out_qs = ep->out_qs;
    a.out_u = ep->out_u; a.out_ue = (const unsigned char*)ep->out_ue;
    if ((a.out_u != nullptr) != (a.out_ue != nullptr)) return DG_ERR_BAD_ARG;
    a.no_y = ep->skip_y ? 1 : 0;
    if (a.no_y && ((!a.out_q && !a.out_u) || a.accumulate)) return DG_ERR_BAD_ARG;          // (the first-layer launcher checks its own shapes)
    // the uniform-scale copy WITHOUT the MXFP8 one and the magnitude census: the first-layer kernel's (gg_launch_im2col_t refuses other shapes)
    a.out_amax = (unsigned*)ep->out_amax;
    if (((a.out_u && !a.out_q) || a.out_amax) && !im2col_small) return DG_ERR_BAD_SHAPE;
    if (a.out_u && (d->dtype != DG_BF16 || d->Nout < 128 || d->Nout % 64 || d->dst_ps)) return DG_ERR_BAD_SHAPE;
    // the MXFP8 copy is written by the 64-channel wave-tile epilogues of bf16 launches: same shape rules as the bit masks
    if ((a.out_q != nullptr) != (a.out_qs != nullptr)) return DG_ERR_BAD_ARG;
    if (a.out_q && (d->dtype != DG_BF16 || d->Nout < 128 || d->Nout % 64)) return DG_ERR_BAD_SHAPE;
    // scale bytes per DESTINATION pixel: a pixel-shuffled output has Nout / 4 channels per pixel (the 32-channel blocks of a channel
    // quarter stay together: quarters of a multiple of 32 channels)
    const int cdst = d->dst_ps ? d->Nout / 4 : d->Nout;
    if (a.out_q && d->dst_ps && (cdst % 32 || a.out_u || a.mask_bits || a.out_bits)) return DG_ERR_BAD_SHAPE;
    a.ldqs = ep->ldqs > 0 ? (int)ep->ldqs : cdst / 32;
    a.qs_shift = 0;
    if (a.out_q && (a.ldqs != cdst / 32 || d->dst_ps)) {   // strided scale rows / shuffled pixels: the kernel splits a (pixel, 16-channel group) index by a shift
      const int n16 = cdst / 16;
      if ((n16 & (n16 - 1)) || a.ldqs < cdst / 32) return DG_ERR_BAD_SHAPE;
      while ((1 << a.qs_shift) < n16) ++a.qs_shift;
    }
    // bit masks need 64-channel wave tiles (Nout >= 128 selects them in every dispatch path) and plain destinations
    if ((a.mask_bits || a.out_bits) && (d->Nout < 128 || d->Nout % 64 || d->dst_ps || (a.mask_bits && a.mask))) return DG_ERR_BAD_SHAPE;
    if ((a.r1 && a.ldr1 % 4) || (a.r2 && a.ldr2 % 4) || (a.mask && a.ldmask % 4)) return DG_ERR_BAD_SHAPE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  a.seg = seg ? 1 : 0;
  if (seg) {
    // class order of the merged stride-2 data-gradient launch (conv_halo.hip): classes interleaved per row of tiles while the nine
    // weight taps of the layer fit an XCD's L2 beside everything else (<= 2 MB: the 128- and 256-channel layers, +3-4 %), else one
    // class per image at a time (512 / 1024 channels: interleaved classes stream four tap sets at once, -2 %).
    if ((long long)d->Cred * d->Nout * 18 <= (2ll << 20)) a.seg = 2;
  }
  if (f8) {
    // ldxs < 0: ONE scale row for every pixel (a uniform-scale source, dg_epilogue.out_u / out_ue: the kernel's scale fetches all hit row 0)
    F8Args f{(const unsigned char*)f8->xs, (const unsigned char*)f8->ws, f8->ldxs > 0 ? (int)f8->ldxs : f8->ldxs < 0 ? 0 : d->Cred / 32};
    if (f8->ldxs >= 0 && (f.ldxs < d->Cred / 32 || f.ldxs % 4)) return DG_ERR_BAD_SHAPE;          // the kernel fetches 4 scale bytes per pixel and K-step as one dword
    return gg_launch_f8(a, f, d->N, st);
  }
  static const bool no_im2col = getenv("DG_GG_NOIM2COL") != nullptr;
  if (im2col_small && !no_im2col) return gg_launch_im2col(a, d->dtype, st);
  return gg_launch(a, d->dtype, d->N, st);
}

extern "C" int dg_gather_gemm(const dg_gg_desc* d, const dg_epilogue* ep, const void* x, const void* w,
                              void* y, void* stream) {
  return gather_gemm_impl(d, ep, x, w, y, stream, false);
}

static int geom_validate(const dg_conv_geom* g) {
  if (!g) return DG_ERR_BAD_ARG;
  if (g->dtype != DG_F32 && g->dtype != DG_BF16) return DG_ERR_BAD_DTYPE;
  if (g->N <= 0 || g->H <= 0 || g->W <= 0) return DG_ERR_BAD_SHAPE;
  if (g->stride != 1 && g->stride != 2) return DG_ERR_BAD_SHAPE;
  if (g->stride == 2 && ((g->H | g->W) & 1)) return DG_ERR_BAD_SHAPE;
  if (g->Cin <= 0 || g->Cin % 8 || g->Cout <= 0 || g->Cout % 16) return DG_ERR_BAD_SHAPE;
  if (g->pixel_shuffle && (g->stride != 1 || (g->Cout / 4) % 16)) return DG_ERR_BAD_SHAPE;
  return DG_OK;
}

extern "C" int dg_conv3x3_plan(const dg_conv_geom* g, int kind, dg_gg_desc* out) {
  int rc = geom_validate(g);
  if (rc) return rc;
  if (!out || (kind != 0 && kind != 1)) return DG_ERR_BAD_ARG;
  const int Ho = g->H / g->stride, Wo = g->W / g->stride;
  if (kind == 0) {
    dg_gg_desc d{};
    d.dtype = g->dtype; d.N = g->N; d.Hs = g->H; d.Ws = g->W; d.Cred = g->Cin; d.lds = g->ldx; d.src_ps = 0;
    d.Hg = Ho; d.Wg = Wo; d.sy_mul = g->stride; d.sx_mul = g->stride;
    d.ntaps = 9;
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s) { d.tap_dy[r * 3 + s] = r - 1; d.tap_dx[r * 3 + s] = s - 1; d.tap_w[r * 3 + s] = r * 3 + s; }
    d.Nout = g->Cout; d.ldw = 9ll * g->Cin;
    d.ldd = g->ldy; d.dy_mul = d.dx_mul = 1; d.dy_off = d.dx_off = 0;
    if (g->pixel_shuffle) { d.dst_ps = 1; d.Hd = 2 * Ho; d.Wd = 2 * Wo; }
    else { d.dst_ps = 0; d.Hd = Ho; d.Wd = Wo; }
    out[0] = d;
    return 1;
  }
  // data gradient: source = dy over the Ho x Wo output grid, destination = dx over H x W
  int nd = 0;
  const int st = g->stride;
  for (int ph = 0; ph < st; ++ph)
    for (int pw = 0; pw < st; ++pw) {
      dg_gg_desc d{};
      d.dtype = g->dtype; d.N = g->N; d.Hs = Ho; d.Ws = Wo; d.Cred = g->Cout; d.lds = g->ldy;
      d.src_ps = g->pixel_shuffle;
      d.Hg = g->H / st; d.Wg = g->W / st; d.sy_mul = 1; d.sx_mul = 1;
      d.ntaps = 0;
      for (int r = 0; r < 3; ++r) {
        if ((ph + 1 - r) % st) continue;
        for (int s = 0; s < 3; ++s) {
          if ((pw + 1 - s) % st) continue;
          // ho = (hi + 1 - r)/st with hi = gy*st + ph
          d.tap_dy[d.ntaps] = (ph + 1 - r) / st; d.tap_dx[d.ntaps] = (pw + 1 - s) / st;
          d.tap_w[d.ntaps] = r * 3 + s;
          ++d.ntaps;
        }
      }
      d.Nout = g->Cin; d.ldw = 9ll * g->Cout;
      d.Hd = g->H; d.Wd = g->W; d.ldd = g->ldx;
      d.dy_mul = d.dx_mul = st; d.dy_off = ph; d.dx_off = pw; d.dst_ps = 0;
      out[nd++] = d;
    }
  return nd;
}

extern "C" int dg_conv3x3_fwd(const dg_conv_geom* g, const dg_epilogue* ep, const void* x, const void* w_fwd,
                              void* y, void* stream) {
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 0, d);
  if (n < 0) return n;
  const bool small = g->cin_real > 0 && g->cin_real <= 2 && g->stride == 1 && !g->pixel_shuffle;
  return gather_gemm_impl(&d[0], ep, x, w_fwd, y, stream, small);
}

// The four parity-class descriptors of a stride-2 data gradient as ONE descriptor for the merged launch of the halo kernel
// (conv_halo.hip, SEG): the classes' taps concatenated in class order (1 / 2 / 2 / 4); the destination offset is the workgroup's
// class parity.  Any epilogue; whole 64-channel reduction blocks (8 chunks), > 64 outputs, tiles of >= 8 x 8.
static bool seg_dgrad_desc(const dg_gg_desc* d, int n, dg_gg_desc* out) {
  const int epc = d[0].dtype == DG_F32 ? 4 : 8;
  if (n != 4 || d[0].src_ps || d[0].dst_ps || (d[0].Cred / epc) % 8 || d[0].Nout <= 64 || d[0].Hg < 8 || d[0].Wg < 8) return false;
  if (d[0].ntaps != 1 || d[1].ntaps != 2 || d[2].ntaps != 2 || d[3].ntaps != 4) return false;
  if (!gg_halo_row_step_ok(d[0].Ws, d[0].lds, d[0].dtype, 1)) return false;     // oversized rows: four launches on the row-tiled kernel
  *out = d[3];
  out->ntaps = 0;
  for (int c = 0; c < 4; ++c) {
    if (d[c].dy_off != (c >> 1) || d[c].dx_off != (c & 1) || d[c].dy_mul != 2 || d[c].dx_mul != 2) return false;
    for (int t = 0; t < d[c].ntaps; ++t, ++out->ntaps) {
      out->tap_dy[out->ntaps] = d[c].tap_dy[t]; out->tap_dx[out->ntaps] = d[c].tap_dx[t]; out->tap_w[out->ntaps] = d[c].tap_w[t];
    }
  }
  return out->ntaps == 9;
}

extern "C" int dg_conv3x3_dgrad(const dg_conv_geom* g, const dg_epilogue* ep, const void* dy, const void* w_dgrad,
                                void* dx, void* stream) {
  if (g && g->Cin % 16) return DG_ERR_BAD_SHAPE;  // dx channels are a GEMM N dimension
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 1, d);
  if (n < 0) return n;
  dg_gg_desc ds;
  if (seg_dgrad_desc(d, n, &ds)) return gather_gemm_impl(&ds, ep, dy, w_dgrad, dx, stream, false, nullptr, true);
  for (int i = 0; i < n; ++i) {
    int rc = dg_gather_gemm(&d[i], ep, dy, w_dgrad, dx, stream);
    if (rc) return rc;
  }
  return DG_OK;
}

// MXFP8 operands (csrc/quant.hip), bf16 output / epilogue tensors.  q->ldxq replaces the source's pixel stride of `g`.
extern "C" int dg_conv3x3_fwd_f8(const dg_conv_geom* g, const dg_epilogue* ep, const dg_f8_operands* q, void* y, void* stream) {
  if (!q || !g || g->dtype != DG_BF16 || g->Cin % 128) return DG_ERR_BAD_SHAPE;       // (pixel-shuffled OUTPUT: the common epilogue's)
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 0, d);
  if (n < 0) return n;
  d[0].lds = q->ldxq;
  return gather_gemm_impl(&d[0], ep, q->xq, q->wq, y, stream, false, q);
}

extern "C" int dg_conv3x3_dgrad_f8(const dg_conv_geom* g, const dg_epilogue* ep, const dg_f8_operands* q, void* dx, void* stream) {
  if (!q || !g || g->dtype != DG_BF16 || g->pixel_shuffle || g->Cout % 128 || g->Cin % 16) return DG_ERR_BAD_SHAPE;
  dg_gg_desc d[4];
  g_last_kinds = 0;
  int n = dg_conv3x3_plan(g, 1, d);
  if (n < 0) return n;
  dg_gg_desc ds;
  if (seg_dgrad_desc(d, n, &ds) && ds.Cred % 128 == 0) {       // stride 2: the four parity classes as one launch (conv_halo_f8.hip, SEG)
    ds.lds = q->ldxq;
    return gather_gemm_impl(&ds, ep, q->xq, q->wq, dx, stream, false, q, true);
  }
  for (int i = 0; i < n; ++i) {
    d[i].lds = q->ldxq;
    int rc = gather_gemm_impl(&d[i], ep, q->xq, q->wq, dx, stream, false, q);
    if (rc) return rc;
  }
  return DG_OK;
}

extern "C" int dg_last_conv_kernels(void) { return g_last_kinds; }

// Host-side probe of the planner (no GPU): how a stride-2 data gradient of geometry `g` would be launched --
// 1 = one merged launch of the halo kernel (conv_halo.hip, SEG), 4 = one launch per parity class, < 0 = dg_status.
extern "C" int dg_conv3x3_dgrad_launches(const dg_conv_geom* g) {
  dg_gg_desc d[4], ds;
  const int n = dg_conv3x3_plan(g, 1, d);
  if (n < 0) return n;
  return seg_dgrad_desc(d, n, &ds) ? 1 : n;
}
