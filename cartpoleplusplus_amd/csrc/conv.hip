// Host-side launchers of the conv kernels (geometry selection + profiling brackets).
#include "conv_dw_kyo.h"
#include "conv_k16.h"
#include "conv_dw16.h"
#include "conv_dwb16.h"
#include <cstdlib>
#include <cstring>

static int pick_xtw(int in_mode, int W) {
  if (in_mode == IN_F16_WHITEN || in_mode == IN_F32_WHITEN) return 4;   // conv1: always 64-wide tiles
  return W > 32 ? 4 : (W > 16 ? 2 : 1);
}

// largest staging chunk (16 / 8 / 4 bytes) that the image base, the image stride and the row length are multiples of;
// 0 if not even one element fits a 4-byte chunk boundary pattern
static int chunk_bytes_for(const ConvArgs& a, int in_mode, int cin) {
  if (in_mode == IN_DY) return 16;
  const long esz = in_mode == IN_F16_WHITEN ? 2 : 4;
  for (int chb = 16; chb >= 4; chb >>= 1)
    if (((uintptr_t)a.in) % chb == 0 && (a.in_bstride * esz) % chb == 0 && ((long)a.W * cin * esz) % chb == 0) return chb;
  return 0;
}

// aligned 16-byte row staging needs 16-byte aligned image bases
static int vec_ok_for(const ConvArgs& a, int in_mode) {
  if (in_mode == IN_DY) return 0;
  const long esz = in_mode == IN_F16_WHITEN ? 2 : 4;
  return (((uintptr_t)a.in) % 16 == 0) && ((a.in_bstride * esz) % 16 == 0);
}

static void set_tiles(ConvArgs& a, int cin, int xtw) {
  const int th = conv_th(cin, xtw);
  a.tiles_x = (a.W + 16 * xtw - 1) / (16 * xtw);
  a.tiles_y = (a.H + th - 1) / th;
  a.ntiles = a.B * a.tiles_x * a.tiles_y;
}

// The f16-pipe kernels' operand windows start up to 128 bytes in front of an image and end up to 256 bytes behind it
// (conv_k16.h / conv_dw16.h; the overhang is masked): the image batch -- or, when images are addressed through replay slots,
// the store it starts -- must be one of the library's own guard-banded allocations.
static int guard_band_check(const ConvArgsN& batch, int cin, const char* who) {
  for (int i = 0; i < batch.n; ++i) {
    const ConvArgs& a = batch.a[i];
    const size_t bytes = a.img_slot ? (size_t)a.in_bstride * 2 : (size_t)a.B * a.in_bstride * 2;    // slots: validated by the replay memory
    if (!cpp_arena_covers(a.in, bytes, 128, 256)) {
      cpp_set_error("%s on the f16 pipes: the state batch at %p is not inside a guard-banded allocation of this library", who, a.in);
      return 1;
    }
  }
  return 0;
}

bool conv1_f16_pipes_ok(const cpp_ctx* ctx, int cin, int H, int W, int B, bool batch_norm) {
  if (ctx && ctx->conv1_f32) return false;      // (nearly constant channels: the f32-input kernels, cpp_ctx::conv1_f32)
  static const bool off = cpp_switch_off("CPP_CONV_K16") ||
                          cpp_switch_off("CPP_CONV_KYO");
  if (off || B < 2 || H < 4) return false;
  ConvArgsN q; memset(&q, 0, sizeof(q));
  q.n = 1; q.a[0].H = H; q.a[0].W = W; q.a[0].B = B; q.a[0].nout = KYO_NO; q.a[0].in_bstride = (long)H * W * cin;
  float dummy = 0.f;
  if (batch_norm) q.a[0].dy_dense = &dummy;
  bool f = false, d = false; int grid = 0;
  (void)conv_fwd_k16_dispatch(nullptr, cin, 5, IN_F16_WHITEN, batch_norm, q, &f);          // (dry run: no context, no launch)
  (void)conv_dw16_dispatch(nullptr, cin, 5, IN_F16_WHITEN, batch_norm, q, &grid, &d);
  return f && d;
}

bool conv12_b16_ok(const cpp_ctx* ctx, int cin, int H, int W, int B) {
  if (ctx && ctx->conv1_f32) return false;
  static const bool off = cpp_switch_off("CPP_CONV_K16") ||
                          cpp_switch_off("CPP_CONV_KYO") ||
                          cpp_switch_off("CPP_CONV_B16");
  if (off || B < 2 || H < 4 || (H & 1) || (W & 1)) return false;
  ConvArgsN q; memset(&q, 0, sizeof(q));
  q.n = 1; q.a[0].H = H; q.a[0].W = W; q.a[0].B = B; q.a[0].nout = KYO_NO; q.a[0].in_bstride = (long)H * W * cin;
  bool f1 = false, f2 = false;
  (void)conv_fwd_k16_dispatch(nullptr, cin, 5, IN_F16_WHITEN, false, q, &f1);            // (dry runs)
  q.a[0].H = H / 2; q.a[0].W = W / 2; q.a[0].in_bstride = (long)(H / 2) * (W / 2) * KYO_NO;
  (void)conv_fwd_kb16_dispatch(nullptr, KYO_NO, 5, IN_F32_PLAIN, q, &f2);
  return f1 && f2;
}

int launch_conv_fwd_multi(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, int epi, const ConvArgs* list, int n) {
  if (n < 1 || n > CONV_BATCH_MAX) { cpp_set_error("conv: batch of %d networks", n); return 1; }
  const int xtw = pick_xtw(in_mode, list[0].W);
  ConvArgsN batch; batch.n = n;
  for (int i = 0; i < n; ++i) {
    ConvArgs& a = batch.a[i];
    a = list[i];
    if (a.H != list[0].H || a.W != list[0].W || a.B != list[0].B || a.nout != list[0].nout) {
      cpp_set_error("conv: batched networks differ in geometry"); return 1;
    }
    set_tiles(a, cin, xtw);
    a.cin_rt = cin;
    a.vec_ok = vec_ok_for(a, in_mode);
    if (a.nout > CPP_NOUT_MAX) { cpp_set_error("conv: nout %d > 16", a.nout); return 1; }
  }
  const ConvArgs& a = batch.a[0];
  prof_begin(ctx);
  int rc;
  // (ky,o)-column kernel: the default for the pooled forward layers whose rows can be staged as aligned 16-byte
  // chunks (CPP_CONV_KYO=0 selects the (ky,(kx,c)) x o kernel for A/B measurements)
  static const bool no_kyo = cpp_switch_off("CPP_CONV_KYO");
  const bool dx_mode = in_mode == IN_DY || in_mode == IN_F32_FLIP;      // dX passes: plain rows out
  const bool plain_fwd = epi == EPI_PLAIN && !dx_mode;                   // batch norm: plain conv output
  bool kyo = !no_kyo && a.nout <= 10 && a.H >= 2;
  int chb = 16;
  for (int i = 0; i < n && kyo; ++i) {
    const int c = chunk_bytes_for(batch.a[i], in_mode, cin);
    if (c == 0) kyo = false; else if (c < chb) chb = c;
  }
  // CPP_CONV_KYO23=0 keeps the narrow layers (conv2 / conv3) on the old kernel
  static const bool no_kyo23 = cpp_switch_off("CPP_CONV_KYO23");
  if ((in_mode == IN_F32_PLAIN || dx_mode) && no_kyo23) kyo = false;
  // conv1 of f16 images: f16 matrix pipes with f32-exact operands (conv_k16.h); CPP_CONV_K16=0 keeps the f32 MFMA kernel.
  // (B = 1 stays on the f32 kernel: action_given is bit-identical to a row of cpp_net_forward_each)
  static const bool no_k16 = cpp_switch_off("CPP_CONV_K16");
  if (!no_kyo && !no_k16 && !ctx->conv1_f32 && in_mode == IN_F16_WHITEN && !dx_mode && a.B >= 2 && a.nout <= 10 && a.H >= 2) {
    bool handled = false;
    if (int grc = guard_band_check(batch, cin, "conv1 forward")) { prof_end(ctx, kid); return grc; }
    rc = conv_fwd_rs16_dispatch(ctx, cin, ks, in_mode, plain_fwd, batch, &handled);
    if (handled) { prof_end(ctx, kid == K_CONV1_FWD ? K_CONV1_FWD_F16X3 : kid); return rc; }
    rc = conv_fwd_k16_dispatch(ctx, cin, ks, in_mode, plain_fwd, batch, &handled);
    if (handled) { prof_end(ctx, kid == K_CONV1_FWD ? K_CONV1_FWD_F16X3 : kid); return rc; }
  }
  for (int i = 0; i < n; ++i)
    if (batch.a[i].img_slot) { cpp_set_error("conv forward: images addressed through replay slots need the f16-pipe conv1 kernel"); prof_end(ctx, kid); return 1; }
  if (!no_kyo && in_mode == IN_F32_PLAIN && !dx_mode && !plain_fwd && epi == EPI_RELU_POOL && a.wscale == 0.f &&
      conv3_img_ok(cin, ks, a.H, a.W, a.nout)) {      // conv3 of 16x16 inputs: whole images in LDS
    rc = launch_conv3_img(ctx, batch);
    prof_end(ctx, kid);
    return rc;
  }
  if (!no_kyo && in_mode == IN_F32_PLAIN && !dx_mode && !plain_fwd && epi == EPI_RELU_POOL && a.wscale == 0.f) {      // conv3 at rows of 32 / 64 pixels: row-streaming on the bf16 pipes (conv_fw_rs.h)
    bool handled = false;
    rc = conv_fw_rs_dispatch(ctx, cin, ks, in_mode, epi, batch, &handled);
    if (handled) { prof_end(ctx, kid); return rc; }
  }
  if (!no_kyo && in_mode == IN_F32_PLAIN && !dx_mode && !plain_fwd && a.in_b16 != nullptr) {      // conv2 from conv1's bf16 planes
    bool handled = false;
    rc = conv_fwd_kb16_dispatch(ctx, cin, ks, in_mode, batch, &handled);
    if (handled) { prof_end(ctx, kid); return rc; }
  }
  if (!no_kyo && in_mode == IN_DY) {      // conv2's / conv3's dX on the bf16 pipes, row-streaming (conv_dx_rs.h)
    bool handled = false;
    rc = conv_dx_rs_dispatch(ctx, cin, ks, in_mode, batch, &handled);
    if (handled) { prof_end(ctx, kid); return rc; }
  }
  if (kyo) {
    // few workgroups (the dX passes carry two networks: one wave per SIMD): split the images into two bands of rows
    // (CPP_CONV_BANDS=0: whole images)
    static const bool no_bands = cpp_switch_off("CPP_CONV_BANDS");
    const int ipw = a.W > 32 ? 1 : (a.W > 16 ? 2 : 4);
    const int wgs = n * ((a.B + ipw - 1) / ipw);
    static const int nb_want = cpp_switch_int("CPP_CONV_NBANDS", 2);
    if (!no_bands && wgs <= ctx->num_cus && a.H >= 16 && (a.H % 4) == 0 && (in_mode == IN_F32_PLAIN || in_mode == IN_DY)) {
      const int nb = (nb_want == 4 && (a.H % 8) == 0 && a.H >= 32) ? 4 : 2;
      for (int i = 0; i < n; ++i) { batch.a[i].nbands = nb; batch.a[i].band_rows = a.H / nb; }
    }
    bool handled = false;
    rc = plain_fwd ? conv_fwd_kyo_dispatch_plain(ctx, cin, ks, in_mode, chb, batch, &handled)
       : (in_mode == IN_F32_PLAIN || dx_mode) ? conv_fwd_kyo_dispatch_l23(ctx, cin, ks, in_mode, chb, batch, &handled)
                                              : conv_fwd_kyo_dispatch_l1(ctx, cin, ks, in_mode, chb, batch, &handled);
    if (handled) { prof_end(ctx, kid); return rc; }
  }
  if (in_mode == IN_F16_WHITEN || in_mode == IN_F32_WHITEN) {
    rc = conv_fwd_dispatch_l1(ctx, cin, ks, xtw, in_mode, epi, batch);
  } else if (in_mode == IN_F32_FLIP) {               // earlier kernel: plain rows in, weights flipped at load time
    for (int i = 0; i < n; ++i) batch.a[i].flip = 1;
    rc = conv_fwd_dispatch_l23(ctx, cin, ks, xtw, IN_F32_PLAIN, epi, batch);
  } else {
    rc = conv_fwd_dispatch_l23(ctx, cin, ks, xtw, in_mode, epi, batch);
  }
  prof_end(ctx, kid);
  return rc;
}

int launch_conv_fwd(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, int epi, ConvArgs a) {
  return launch_conv_fwd_multi(ctx, kid, cin, ks, in_mode, epi, &a, 1);
}

size_t conv_dw_partial_floats(cpp_ctx* ctx, int cin, int ks, int nout) {
  return (size_t)(ctx->num_cus * 4) * (size_t)(ks * ks * cin * nout + nout);   // one partial per resident workgroup
}

// the per-workgroup partials of a dW launch are summed by conv_dw_reduce_kernel: queued here, sent by flush_dw_reduce
static int queue_dw_reductions(cpp_ctx* ctx, const ConvArgsN& batch, int grid, int nw, int nout, float* const* grad_w, float* const* grad_b) {
  for (int i = 0; i < batch.n; ++i) {
    if (ctx->npending == DW_REDUCE_MAX) { const int rc = flush_dw_reduce(ctx); if (rc) return rc; }
    DwReduceDesc& d = ctx->pending[ctx->npending++];
    d.partial = batch.a[i].partial; d.nblocks = grid; d.pstride = nw + nout; d.nw = nw; d.nout = nout;
    d.grad_w = grad_w[i]; d.grad_b = grad_b[i];
    d.sq_part = nullptr;
    const int grp = ctx->sq_conv_group[i];
    if (ctx->sq_part && grp >= 0 && ctx->sq_n[grp] >= 0) {
      const int nb = (nw + nout + 63) / 64;
      if (ctx->sq_n[grp] + nb <= SQ_REGION) { d.sq_part = ctx->sq_part + grp * SQ_REGION + ctx->sq_n[grp]; ctx->sq_n[grp] += nb; }
      else ctx->sq_n[grp] = -1;                      // (cannot happen for the reference's layer sizes; the caller falls back to sumsq)
    }
  }
  return 0;
}

int launch_conv_dw_multi(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, const ConvArgs* list, int n,
                         float* const* grad_w, float* const* grad_b) {
  if (n < 1 || n > CONV_BATCH_MAX) { cpp_set_error("conv dW: batch of %d networks", n); return 1; }
  const int xtw = pick_xtw(in_mode, list[0].W);
  const int nout = list[0].nout;
  const int nw = ks * ks * cin * nout;
  ConvArgsN batch; batch.n = n;
  for (int i = 0; i < n; ++i) {
    ConvArgs& a = batch.a[i];
    a = list[i];
    if (a.H != list[0].H || a.W != list[0].W || a.B != list[0].B || a.nout != nout) {
      cpp_set_error("conv dW: batched networks differ in geometry"); return 1;
    }
    set_tiles(a, cin, xtw);
    a.cin_rt = cin;
    a.vec_ok = vec_ok_for(a, in_mode);
    a.pstride = nw + nout;
  }
  int grid = 0, rc;
  prof_begin(ctx);
  // (ky,o)-column kernel for the 5x5 layers (CPP_CONV_KYO=0: old kernel); dense dY rows (batch norm) only exist there
  static const bool no_kyo = cpp_switch_off("CPP_CONV_KYO");
  const bool dense = batch.a[0].dy_dense != nullptr;
  bool kyo = dense || (!no_kyo && ks == 5 && nout <= 10 && (batch.a[0].H % 2) == 0 && batch.a[0].H >= 4);
  int chb = 16;
  for (int i = 0; i < n && kyo; ++i) {
    const int c = chunk_bytes_for(batch.a[i], in_mode, cin);
    if (c == 0) kyo = false; else if (c < chb) chb = c;
  }
  bool handled = false;
  // conv1 of f16 images: f16 matrix pipes with f32-exact operands (conv_dw16.h); CPP_CONV_K16=0 keeps the f32 MFMA kernel
  static const bool no_k16 = cpp_switch_off("CPP_CONV_K16");
  if (!no_kyo && !no_k16 && !ctx->conv1_f32 && in_mode == IN_F16_WHITEN) {
    if (int grc = guard_band_check(batch, cin, "conv1 dW")) { prof_end(ctx, kid); return grc; }
    const bool ride_open = ctx->ride != nullptr && !ctx->ride_done;
    rc = conv_dw16_dispatch(ctx, cin, ks, in_mode, dense, batch, &grid, &handled);
    if (handled) kid = kid == K_CONV1_DW ? K_CONV1_DW_F16X3 : kid;
    if (handled && ride_open && ctx->ride_done) kid = K_CONV1_DW_GATHER;      // the next minibatch's sample pass left with it
  }
  // conv2 (f32 activations in): bf16 pipes, three exact pieces per operand (conv_dwb16.h); CPP_CONV_B16=0 keeps the f32 MFMA kernel
  static const bool no_b16 = cpp_switch_off("CPP_CONV_B16");
  static const bool no_dwb16 = cpp_switch_off("CPP_CONV2_DW_B16");      // conv2's dW alone back on the f32 MFMA kernel (A/B)
  if (!handled && !no_kyo && !no_b16 && !no_dwb16 && !dense && in_mode == IN_F32_PLAIN)      // 32-wide inputs: one wave per unit (conv_dw_rs.h)
    rc = conv_dw_rs_dispatch(ctx, cin, ks, in_mode, batch, &grid, &handled);
  if (!handled && !no_kyo && !no_b16 && !no_dwb16 && !dense && in_mode == IN_F32_PLAIN)
    rc = conv_dwb16_dispatch(ctx, cin, ks, in_mode, batch, &grid, &handled);
  if (!handled)
    for (int i = 0; i < n; ++i)
      if (batch.a[i].img_slot) { cpp_set_error("conv dW: images addressed through replay slots need the f16-pipe conv1 kernel"); prof_end(ctx, kid); return 1; }
  if (!handled && kyo) rc = conv_dw_kyo_dispatch(ctx, cin, ks, in_mode, chb, dense, batch, &grid, &handled);
  if (dense && !handled) {
    cpp_set_error("conv dW from dense dY rows (batch norm): no kernel for %dx%d, %d channels, %dx%d taps, %d-byte rows chunks",
                  batch.a[0].H, batch.a[0].W, cin, ks, ks, chb);
    prof_end(ctx, kid);
    return 1;
  }
  if (handled) {
  } else if (in_mode == IN_F16_WHITEN || in_mode == IN_F32_WHITEN)
    rc = conv_dw_dispatch_l1(ctx, cin, ks, xtw, in_mode, batch, &grid);
  else
    rc = conv_dw_dispatch_l23(ctx, cin, ks, xtw, in_mode, batch, &grid);
  prof_end(ctx, kid);
  if (rc) return rc;
  return queue_dw_reductions(ctx, batch, grid, nw, nout, grad_w, grad_b);
}

int launch_conv_dw(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, ConvArgs a, float* grad_w,
                   float* grad_b) {
  return launch_conv_dw_multi(ctx, kid, cin, ks, in_mode, &a, 1, &grad_w, &grad_b);
}

int flush_dw_reduce(cpp_ctx* ctx) {
  if (ctx->npending == 0) return 0;
  DwReduceBatch rb;
  rb.n = ctx->npending; rb.block_start[0] = 0;
  for (int i = 0; i < rb.n; ++i) {
    rb.d[i] = ctx->pending[i];
    rb.block_start[i + 1] = rb.block_start[i] + (rb.d[i].nw + rb.d[i].nout + 63) / 64;
  }
  ctx->npending = 0;
  if (ctx->ride && !ctx->ride_done) {                 // the next minibatch's sample + statistics pass shares the launch
    ctx->ride_done = true;
    return launch_reduce_gather(ctx, rb, *ctx->ride, ctx->ride_dtype);
  }
  // the statistics of a sample pass that has ALREADY left (with conv1's dW) can be finished here: the tables are then in memory before
  // the optimiser's launch starts, whose conv1 image rider needs them
  const StatsRide* st = (ctx->st_ride && !ctx->st_ride_done && (!ctx->ride || ctx->ride_done)) ? ctx->st_ride : nullptr;
  if (st) ctx->st_ride_done = true;
  prof_begin(ctx);
  int rc = launch_dw_reduce_batch(ctx, rb, st);
  prof_end(ctx, K_DW_REDUCE);
  return rc;
}
