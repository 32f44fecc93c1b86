// networks: layout (TF variable order), workspaces, launch sequences of the conv trunk / MLP stacks, cpp_net_* entry points
#include "rt_internal.h"

static int net_build(cpp_net* n) {
  const cpp_net_spec& s = n->spec;
  long off = 0;
  int h = s.H, w = s.W, cin = s.C;
  if (s.pixel) {
    for (int i = 0; i < 3; ++i) {
      ConvL L; L.H = h; L.W = w; L.Cin = cin; L.ks = kConvKs[i]; L.Hp = h / 2; L.Wp = w / 2;
      L.w_off = off; off += (long)L.ks * L.ks * cin * kConvOut; L.b_off = off; off += kConvOut;
      n->conv.push_back(L);
      VarInfo vw; vw.name = std::string(kConvNames[i]) + "/weights"; vw.rank = 4;
      vw.shape[0] = L.ks; vw.shape[1] = L.ks; vw.shape[2] = cin; vw.shape[3] = kConvOut; vw.offset = L.w_off;
      VarInfo vb; vb.name = std::string(kConvNames[i]) + (s.use_batch_norm ? "/BatchNorm/beta" : "/biases"); vb.rank = 1;
      vb.shape[0] = kConvOut; vb.shape[1] = vb.shape[2] = vb.shape[3] = 0; vb.offset = L.b_off;
      n->vars.push_back(vw); n->vars.push_back(vb);
      cin = kConvOut; h /= 2; w /= 2;
    }
    if (h < 1 || w < 1) { cpp_set_error("image %dx%d too small for three 2x2 pools", s.H, s.W); return CPP_ERR_ARG; }
    n->flat = h * w * kConvOut;
    n->state_elems = (long)s.H * s.W * s.C;
  } else {
    n->flat = s.state_elems;
    n->state_elems = s.state_elems;
  }
  const int A = s.action_dim;
  auto add_fc = [&](const std::string& name, int n_in, int n_out, int act, int cat) {
    FcL L; L.n_in = n_in; L.n_out = n_out; L.act = act; L.cat = cat; L.w_off = off; L.name = name;
    off += (long)n_in * n_out + n_out;
    n->fc.push_back(L);
    VarInfo vw; vw.name = name + "/weights"; vw.rank = 2; vw.shape[0] = n_in; vw.shape[1] = n_out;
    vw.shape[2] = vw.shape[3] = 0; vw.offset = L.w_off;
    VarInfo vb; vb.name = name + "/biases"; vb.rank = 1; vb.shape[0] = n_out;
    vb.shape[1] = vb.shape[2] = vb.shape[3] = 0; vb.offset = L.w_off + (long)n_in * n_out;
    n->vars.push_back(vw); n->vars.push_back(vb);
  };
  n->cat_layer = -1;
  int n_in = n->flat;
  const int hid_act = s.use_dropout ? GE_RELU_DROPOUT : GE_RELU;       // hidden_layers_starting_at with opts (base_network.py:69-70)
  if (s.kind == CPP_ACTOR) {
    for (int i = 0; i < s.n_hidden; ++i) { add_fc("h" + std::to_string(i), n_in, s.hidden[i], hid_act, 0); n_in = s.hidden[i]; }
    add_fc("output_action", n_in, A, GE_TANH, 0);                       // ddpg_cartpole.py:95-100
  } else if (s.kind == CPP_HEAD) {                                      // naf_cartpole.py:104-109,156-161,180-184
    for (int i = 0; i < s.n_hidden; ++i) { add_fc("h" + std::to_string(i), n_in, s.hidden[i], hid_act, 0); n_in = s.hidden[i]; }
    add_fc("fc", n_in, s.head_out, s.head_act == 2 ? GE_TANH : GE_NONE, 0);
  } else if (s.pixel) {                                                 // ddpg_cartpole.py:168-171 (intent)
    add_fc("hidden1", n_in, 200, GE_RELU, 0);
    add_fc("hidden2", 200, 50, GE_RELU, 0);
    add_fc("hidden3", 50 + A, 50, GE_RELU, 1); n->cat_layer = 2;
    add_fc("q_value", 50, 1, GE_NONE, 0);
  } else {                                                              // ddpg_cartpole.py:174-177
    n_in += A;
    for (int i = 0; i < s.n_hidden; ++i) { add_fc("h" + std::to_string(i), n_in, s.hidden[i], GE_RELU, i == 0); n_in = s.hidden[i]; }
    n->cat_layer = 0;
    add_fc("q_value", n_in, 1, GE_NONE, 0);
  }
  n->nparams = off;
  return CPP_OK;
}

static int ws_alloc(cpp_net* n, Workspace& w, int from_layer, bool trunk) {
  const int mb = n->maxB;
  const int nfc = (int)n->fc.size();
  w.fcin.assign(nfc, nullptr);
  w.dz.assign(nfc, nullptr);
  for (int l = from_layer; l < nfc; ++l) {
    const FcL& L = n->fc[l];
    RC(dalloc(n->arena, &w.fcin[l], (size_t)mb * (L.n_in + 1)));
    RC(launch_fill(n->ctx, w.fcin[l], L.n_in + 1, L.n_in, 1, mb, 1.0f));   // the bias "ones" column
    RC(dalloc(n->arena, &w.dz[l], (size_t)mb * L.n_out));
  }
  RC(dalloc(n->arena, &w.out, (size_t)mb * n->fc.back().n_out));
  if (trunk && n->spec.pixel) {
    for (int i = 0; i < 3; ++i) {
      const ConvL& L = n->conv[i];
      const size_t pe = (size_t)mb * L.Hp * L.Wp * kConvOut;
      if (i < 2) RC(dalloc(n->arena, &w.pool[i], pe)); else w.pool[i] = w.fcin[0];
      if (i == 0 && !n->spec.use_batch_norm) RC(dalloc(n->arena, &w.pool_b16, 3 * pe));
      RC(dalloc(n->arena, &w.amax[i], pe));
      RC(dalloc(n->arena, &w.dpool[i], pe));
      if (i == 0) RC(dalloc(n->arena, &w.dpool_imax, (size_t)mb * DX_IMAX_SLOTS));
      if (n->spec.use_batch_norm) {
        RC(dalloc(n->arena, &w.z[i], (size_t)mb * L.H * L.W * kConvOut));
        RC(dalloc(n->arena, &w.bn_stat[i], (size_t)2 * kConvOut));
      }
    }
  }
  return CPP_OK;
}

extern "C" int cpp_net_create(cpp_ctx* ctx, const cpp_net_spec* spec, int max_batch, cpp_net** out) {
  ARG_CHECK(ctx && spec && out, "cpp_net_create: NULL argument");
  ARG_CHECK(max_batch >= 1, "cpp_net_create: max_batch %d", max_batch);
  ARG_CHECK(spec->kind == CPP_ACTOR || spec->kind == CPP_CRITIC || spec->kind == CPP_HEAD, "cpp_net_create: kind %d", spec->kind);
  if (spec->kind == CPP_HEAD) ARG_CHECK(spec->head_out >= 1 && spec->head_out <= 64 && (spec->head_act == 0 || spec->head_act == 2),
                                        "cpp_net_create: head_out %d head_act %d", spec->head_out, spec->head_act);
  ARG_CHECK(spec->action_dim >= 1 && spec->action_dim <= 16, "cpp_net_create: action_dim %d", spec->action_dim);
  ARG_CHECK(spec->n_hidden >= 0 && spec->n_hidden <= 8, "cpp_net_create: n_hidden %d", spec->n_hidden);
  if (spec->pixel) ARG_CHECK(spec->H >= 8 && spec->W >= 8 && spec->C >= 1 && spec->C <= CPP_MAX_CHANNELS,
                             "cpp_net_create: pixel dims %dx%dx%d", spec->H, spec->W, spec->C);
  else ARG_CHECK(spec->state_elems >= 1, "cpp_net_create: state_elems %d", spec->state_elems);
  if (spec->kind == CPP_ACTOR || (spec->kind == CPP_CRITIC && !spec->pixel)) ARG_CHECK(spec->n_hidden >= 1, "cpp_net_create: need hidden layers");
  HIP_CHECK(hipSetDevice(ctx->device));
  cpp_net* n = new cpp_net();
  n->ctx = ctx; n->spec = *spec; n->maxB = max_batch; n->arena.stream = ctx->stream;
  n->grads = nullptr; n->own_grads = nullptr; n->stage_state = nullptr; n->stage_action = nullptr;
  n->stage_out = nullptr; n->dw_partial[0] = n->dw_partial[1] = n->dw_partial[2] = nullptr; n->white = nullptr; n->white_rows = nullptr; n->stats_part = nullptr;
  n->img_slot = nullptr; n->use_b16 = false; n->wimg = nullptr; n->wimg_key = nullptr;
  n->is_training = true; n->drop_counter = nullptr; n->bn_part = nullptr; n->bn_means = nullptr; n->bn_scratch = nullptr;
  int rc = net_build(n);
  if (rc) { delete n; return rc; }
  auto fail = [&](int r) { n->arena.release(); delete n; return r; };
  if ((rc = dalloc(n->arena, &n->params, (size_t)n->nparams))) return fail(rc);
  if (spec->use_dropout && (rc = n->arena.alloc((void**)&n->drop_counter, sizeof(uint64_t), true))) return fail(rc);
  if ((rc = ws_alloc(n, n->ws[0], 0, true))) return fail(rc);
  if (spec->kind == CPP_CRITIC) {
    n->ws[1] = n->ws[0];
    if ((rc = ws_alloc(n, n->ws[1], n->cat_layer, false))) return fail(rc);
    for (int l = 0; l < n->cat_layer; ++l) { n->ws[1].fcin[l] = n->ws[0].fcin[l]; n->ws[1].dz[l] = n->ws[0].dz[l]; }
    for (int i = 0; i < 3; ++i) { n->ws[1].pool[i] = n->ws[0].pool[i]; n->ws[1].amax[i] = n->ws[0].amax[i]; n->ws[1].dpool[i] = n->ws[0].dpool[i]; n->ws[1].dpool_imax = n->ws[0].dpool_imax;
                                  n->ws[1].z[i] = n->ws[0].z[i]; n->ws[1].bn_stat[i] = n->ws[0].bn_stat[i]; }
  }
  if (spec->pixel) {
    for (int i = 0; i < 3; ++i)
      if ((rc = dalloc(n->arena, &n->dw_partial[i], conv_dw_partial_floats(ctx, n->conv[i].Cin, n->conv[i].ks, kConvOut)))) return fail(rc);
    if ((rc = n->arena.alloc(&n->wimg, conv_rs16_image_bytes(), true))) return fail(rc);
    if ((rc = dalloc(n->arena, &n->white, (size_t)2 * spec->C))) return fail(rc);
    if ((rc = dalloc(n->arena, &n->white_rows, (size_t)max_batch * 2 * spec->C))) return fail(rc);
    if ((rc = dalloc(n->arena, &n->stats_part, (size_t)2 * max_batch * 2 * spec->C))) return fail(rc);
    if (spec->use_batch_norm) {
      size_t pd = (size_t)2 * max_batch * 2 * kConvOut;
      if (pd < bn_part_doubles(kConvOut)) pd = bn_part_doubles(kConvOut);
      if ((rc = n->arena.alloc((void**)&n->bn_part, pd * sizeof(double), false))) return fail(rc);
      if ((rc = dalloc(n->arena, &n->bn_means, (size_t)2 * kConvOut))) return fail(rc);
      if ((rc = dalloc(n->arena, &n->bn_scratch, (size_t)kConvOut))) return fail(rc);
    }
  }
  HIP_CHECK(ctx_sync_stream(ctx));
  *out = n;
  return CPP_OK;
}

extern "C" int cpp_net_destroy(cpp_net* n) {
  if (!n) return CPP_OK;
  (void)hipSetDevice(n->ctx->device);
  (void)ctx_sync_stream(n->ctx);
  n->arena.release();
  delete n;
  return CPP_OK;
}

extern "C" int64_t cpp_net_num_params(const cpp_net* n) { return n ? n->nparams : -1; }
extern "C" int cpp_net_num_vars(const cpp_net* n) { return n ? (int)n->vars.size() : -1; }
extern "C" int cpp_net_var_info(const cpp_net* n, int i, char* name, int cap, int* rank, int shape[4], int64_t* offset) {
  ARG_CHECK(n && i >= 0 && i < (int)n->vars.size(), "cpp_net_var_info: index %d", i);
  const VarInfo& v = n->vars[i];
  if (name && cap > 0) { strncpy(name, v.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (rank) *rank = v.rank;
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = v.shape[k];
  if (offset) *offset = v.offset;
  return CPP_OK;
}

extern "C" int cpp_net_set_params(cpp_net* n, const float* host, int64_t cnt) {
  ARG_CHECK(n && host, "cpp_net_set_params: NULL argument");
  ARG_CHECK(cnt == n->nparams, "cpp_net_set_params: got %ld values, network has %ld", (long)cnt, n->nparams);
  n->wimg_key = nullptr;
  HIP_CHECK(hipMemcpyAsync(n->params, host, cnt * sizeof(float), hipMemcpyHostToDevice, n->ctx->stream));
  HIP_CHECK(ctx_sync_stream(n->ctx));
  return CPP_OK;
}
extern "C" int cpp_net_get_params(cpp_net* n, float* host, int64_t cnt) {
  ARG_CHECK(n && host, "cpp_net_get_params: NULL argument");
  ARG_CHECK(cnt == n->nparams, "cpp_net_get_params: asked %ld values, network has %ld", (long)cnt, n->nparams);
  HIP_CHECK(hipMemcpyAsync(host, n->params, cnt * sizeof(float), hipMemcpyDeviceToHost, n->ctx->stream));
  HIP_CHECK(ctx_sync_stream(n->ctx));
  return CPP_OK;
}
extern "C" int cpp_net_get_grads(cpp_net* n, float* host, int64_t cnt) {
  ARG_CHECK(n && host, "cpp_net_get_grads: NULL argument");
  ARG_CHECK(cnt == n->nparams, "cpp_net_get_grads: asked %ld values, network has %ld", (long)cnt, n->nparams);
  if (!n->grads) { cpp_set_error("cpp_net_get_grads: network has no train op (init_ops_for_training not called)"); return CPP_ERR_STATE; }
  HIP_CHECK(hipMemcpyAsync(host, n->grads, cnt * sizeof(float), hipMemcpyDeviceToHost, n->ctx->stream));
  HIP_CHECK(ctx_sync_stream(n->ctx));
  return CPP_OK;
}

extern "C" int cpp_net_soft_update(cpp_net* target, const cpp_net* source, float coeff) {
  ARG_CHECK(target && source, "cpp_net_soft_update: NULL argument");
  ARG_CHECK(coeff >= 0.f && coeff <= 1.f, "affine_combo_coeff %g outside [0,1]", coeff);    // base_network.py:22
  ARG_CHECK(target->nparams == source->nparams, "cpp_net_soft_update: shapes differ (%ld vs %ld)",
            target->nparams, source->nparams);                                             // base_network.py:30
  target->wimg_key = nullptr;
  return launch_soft_update(target->ctx, target->params, source->params, target->nparams, nullptr, nullptr, 0, coeff);
}

// --- launch sequences -------------------------------------------------------------------------
int gemm(cpp_ctx* ctx, const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn,
                float* C, long ldc, int M, int N, int K, int epi, const float* Y, long ldy,
                int accumulate) {
  GemmArgs g; memset(&g, 0, sizeof(g)); g.accumulate = accumulate; g.A = A; g.sAm = sAm; g.sAk = sAk; g.B = Bm; g.sBk = sBk; g.sBn = sBn; g.C = C; g.ldc = ldc;
  g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.epi = epi;
  return launch_gemm(ctx, g);
}


// launch descriptors of conv layer i of a network (forward, dW, dX)
ConvArgs conv_fwd_args(cpp_net* n, Workspace& w, int i, const void* state, int dtype, const float* white, int B, int* mode,
                              long white_bstride) {
  const ConvL& L = n->conv[i];
  ConvArgs a; memset(&a, 0, sizeof(a));
  if (i == 0) { a.in = state; a.in_bstride = n->state_elems; a.scale = white; a.shift = white + n->spec.C; a.white_bstride = white_bstride;
                a.img_slot = n->img_slot; a.wimg = n->wimg; a.wimg_key = n->wimg_key;
                *mode = dtype == CPP_F16 ? IN_F16_WHITEN : IN_F32_WHITEN; }
  else { a.in = w.pool[i - 1]; a.in_bstride = (long)L.H * L.W * L.Cin; *mode = IN_F32_PLAIN; }
  a.w = n->params + L.w_off; a.bias = n->params + L.b_off;
  a.out = w.pool[i]; a.out_bstride = (i == 2) ? (long)n->flat + 1 : (long)L.Hp * L.Wp * kConvOut;
  a.out_amax = w.amax[i];
  a.B = B; a.H = L.H; a.W = L.W; a.nout = kConvOut;
  if (n->use_b16 && w.pool_b16) {
    const long plane = (long)n->maxB * n->conv[0].Hp * n->conv[0].Wp * kConvOut;      // halves per plane
    if (i == 0) { a.out_b16 = w.pool_b16; a.out_b16_plane = plane; }
    if (i == 1) { a.in_b16 = w.pool_b16; a.plane_stride = plane * 2; }
  }
  return a;
}
void conv_dy_desc(cpp_net* n, Workspace& w, int i, ConvArgs& a, int B) {
  const ConvL& L = n->conv[i];
  a.dy.dpool = w.dpool[i]; a.dy.pool = w.pool[i]; a.dy.amax = w.amax[i];
  a.dy.dpool_bstride = (i == 2) ? (long)n->flat : (long)L.Hp * L.Wp * kConvOut;
  a.dy.pool_bstride = (i == 2) ? (long)n->flat + 1 : (long)L.Hp * L.Wp * kConvOut;
  a.dy.Hp = L.Hp; a.dy.Wp = L.Wp;
  a.B = B; a.H = L.H; a.W = L.W;
}
ConvArgs conv_dw_args(cpp_net* n, Workspace& w, int i, const void* state, int dtype, const float* white, int B, int* mode) {
  const ConvL& L = n->conv[i];
  ConvArgs d; memset(&d, 0, sizeof(d));
  conv_dy_desc(n, w, i, d, B);
  if (i == 0) { d.in = state; d.in_bstride = n->state_elems; d.scale = white; d.shift = white + n->spec.C; d.img_slot = n->img_slot;
                *mode = dtype == CPP_F16 ? IN_F16_WHITEN : IN_F32_WHITEN; }
  else { d.in = w.pool[i - 1]; d.in_bstride = (long)L.H * L.W * L.Cin; *mode = IN_F32_PLAIN; }
  d.nout = kConvOut; d.partial = n->dw_partial[i];
  // conv1's dW: the bound of |dpool[0]| per image that conv2's dX left (the same predicate as that launcher's: conv_dx_rs_dispatch)
  if (i == 0 && !n->spec.use_batch_norm && w.dpool_imax && (n->conv[1].W == 32 || n->conv[1].W == 64) &&
      conv_dx_rs_ok(n->ctx, kConvOut, n->conv[1].ks, n->conv[1].H, n->conv[1].W, kConvOut)) d.dy.imax = w.dpool_imax;
  return d;
}
ConvArgs conv_dx_args(cpp_net* n, Workspace& w, int i, int B) {
  const ConvL& L = n->conv[i];
  ConvArgs x; memset(&x, 0, sizeof(x));
  conv_dy_desc(n, w, i, x, B);
  x.w = n->params + L.w_off; x.nout = L.Cin;
  x.out = w.dpool[i - 1]; x.out_bstride = (long)L.H * L.W * L.Cin;
  if (i == 1 && !n->spec.use_batch_norm) x.dx_imax = w.dpool_imax;      // (conv2's dX on conv_dx_rs.h leaves the bound conv1's dW scales by)
  return x;
}

// slim.batch_norm's epsilon; the moving variance stays at its initial 1 (never updated by the reference's train ops)
static const double kBnEps = 1e-3;

// descriptor of layer i of a batch-norm network for the bn.hip launches
BnNet bn_net_desc(cpp_net* n, Workspace& w, int i) {
  const ConvL& L = n->conv[i];
  BnNet d; memset(&d, 0, sizeof(d));
  d.z = w.z[i]; d.stat = w.bn_stat[i]; d.beta = n->params + L.b_off;
  d.pool = w.pool[i]; d.pool_bstride = (i == 2) ? (long)n->flat + 1 : (long)L.Hp * L.Wp * kConvOut; d.amax = w.amax[i];
  d.dpool = w.dpool[i]; d.dpool_bstride = (i == 2) ? (long)n->flat : (long)L.Hp * L.Wp * kConvOut;
  d.part = n->bn_part; d.means = n->bn_means; d.dbeta = n->grads ? n->grads + L.b_off : nullptr;
  return d;
}
BnBatch bn_batch(cpp_net* const* nets, int nn, int i, int B) {
  BnBatch bb; memset(&bb, 0, sizeof(bb));
  const ConvL& L = nets[0]->conv[i];
  bb.count = nn; bb.B = B; bb.H = L.H; bb.W = L.W; bb.C = kConvOut;
  for (int k = 0; k < nn; ++k) bb.n[k] = bn_net_desc(nets[k], nets[k]->ws[0], i);
  return bb;
}

// conv trunk (pixel) or state conversion (low-dim) into ws.fcin[0]
// conv1 on the f16 pipes can leave bf16 planes of pool1 for a conv2 forward on the bf16 pipes (same launch sequence only)
bool trunk_b16(const cpp_net* n, int dtype, int B, long white_bstride) {
  return n->spec.pixel && !n->spec.use_batch_norm && dtype == CPP_F16 && white_bstride == 0 &&
         conv12_b16_ok(n->ctx, n->conv[0].Cin, n->conv[0].H, n->conv[0].W, B);
}

int net_forward_trunk(cpp_net* n, Workspace& w, const void* state, int dtype, const float* white, int B,
                             long white_bstride) {
  cpp_ctx* ctx = n->ctx;
  n->use_b16 = trunk_b16(n, dtype, B, white_bstride);
  if (!n->spec.pixel)
    return launch_state_to_f32(ctx, w.fcin[0], n->fc[0].n_in + 1, state, dtype, n->state_elems, B);
  for (int i = 0; i < 3; ++i) {
    int mode;
    ConvArgs a = conv_fwd_args(n, w, i, state, dtype, white, B, &mode, white_bstride);
    if (i == 0) n->wimg_key = nullptr;
    if (!n->spec.use_batch_norm) {
      RC(launch_conv_fwd(ctx, kFwdKid[i], n->conv[i].Cin, n->conv[i].ks, mode, EPI_RELU_POOL, a));
    } else if (!n->is_training) {
      // inference: (z - 0) / sqrt(1 + eps) + beta  ==  the fused kernel with scaled weights and beta as the bias
      a.wscale = (float)(1.0 / sqrt(1.0 + kBnEps));
      RC(launch_conv_fwd(ctx, kFwdKid[i], n->conv[i].Cin, n->conv[i].ks, mode, EPI_RELU_POOL, a));
    } else {
      const ConvL& L = n->conv[i];
      ConvArgs p = a;                                   // plain conv output (no bias) -> statistics -> BN + ReLU + pool
      p.out = w.z[i]; p.out_bstride = (long)L.H * L.W * kConvOut; p.out_amax = nullptr; p.bias = nullptr;
      RC(launch_conv_fwd(ctx, kFwdKid[i], L.Cin, L.ks, mode, EPI_PLAIN, p));
      BnBatch bb; memset(&bb, 0, sizeof(bb));
      bb.count = 1; bb.B = B; bb.H = L.H; bb.W = L.W; bb.C = kConvOut; bb.n[0] = bn_net_desc(n, w, i);
      RC(launch_bn_forward(ctx, bb, kBnEps));
    }
  }
  return CPP_OK;
}

// the same for several batch-norm networks in training mode, layer by layer: ONE plain-conv launch for all of them
// (the (ky,o) kernel needs the four networks of a minibatch to fill the chip), then statistics + BN/ReLU/pool per network
int nets_forward_trunk_bn(cpp_ctx* ctx, cpp_net* const* nets, int nn, const void* const* states, const float* const* whites,
                                 int dtype, int B) {
  for (int i = 0; i < 3; ++i) {
    const ConvL& L = nets[0]->conv[i];
    ConvArgs full[CONV_BATCH_MAX], plain[CONV_BATCH_MAX]; int mode = 0;
    for (int k = 0; k < nn; ++k) {
      full[k] = conv_fwd_args(nets[k], nets[k]->ws[0], i, states[k], dtype, whites[k], B, &mode);
      plain[k] = full[k];
      plain[k].out = nets[k]->ws[0].z[i]; plain[k].out_bstride = (long)L.H * L.W * kConvOut; plain[k].out_amax = nullptr; plain[k].bias = nullptr;
    }
    RC(launch_conv_fwd_multi(ctx, kFwdKid[i], L.Cin, L.ks, mode, EPI_PLAIN, plain, nn));
    RC(launch_bn_forward(ctx, bn_batch(nets, nn, i, B), kBnEps));
  }
  return CPP_OK;
}

// fully connected layers [from, end); `action` (device, (B, A)) is spliced in front of the cat layer
GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc,
                        int M, int N, int K, int epi) { return mk_gemm(A, sAm, sAk, Bm, sBk, sBn, C, ldc, M, N, K, epi, nullptr, 0); }
int net_forward_fc(cpp_net* n, Workspace& w, int from, int B, const float* action) {
  cpp_ctx* ctx = n->ctx;
  const int nfc = (int)n->fc.size(), A = n->spec.action_dim;
  for (int l = from; l < nfc; ++l) {
    const FcL& L = n->fc[l];
    if (L.cat) {
      if (!action) { cpp_set_error("critic forward needs an action batch"); return CPP_ERR_ARG; }
      RC(launch_copy_cols(ctx, w.fcin[l], L.n_in + 1, L.n_in - A, action, A, 0, A, B));
    }
    float* C = (l + 1 < nfc) ? w.fcin[l + 1] : w.out;
    const long ldc = (l + 1 < nfc) ? n->fc[l + 1].n_in + 1 : L.n_out;
    GemmArgs g = mk_gemm(w.fcin[l], L.n_in + 1, 1, n->params + L.w_off, L.n_out, 1, C, ldc, B, L.n_out, L.n_in + 1, L.act, nullptr, 0);
    set_dropout(g, n, l);
    RC(launch_gemm(ctx, g));
  }
  return from == 0 ? bump_dropout(n) : CPP_OK;
}

// conv trunk backward from w.dpool[2] (= d flat): dW/db of the three convs, dX for conv3/conv2
// batch norm (training mode): the gradient w.r.t. the plain conv output is dense -- reductions over the pooled tensors,
// dz written over z, then dW and dX from dense rows.  dbeta goes straight into the '<conv>/BatchNorm/beta' slot.
static int net_backward_conv_bn(cpp_net* n, Workspace& w, int B, const void* state, int dtype, const float* white) {
  cpp_ctx* ctx = n->ctx;
  for (int i = 2; i >= 0; --i) {
    const ConvL& L = n->conv[i];
    const long zbs = (long)L.H * L.W * kConvOut;
    BnBatch bb; memset(&bb, 0, sizeof(bb));
    bb.count = 1; bb.B = B; bb.H = L.H; bb.W = L.W; bb.C = kConvOut; bb.n[0] = bn_net_desc(n, w, i);
    RC(launch_bn_backward(ctx, bb));
    int mode;
    ConvArgs d = conv_dw_args(n, w, i, state, dtype, white, B, &mode);
    d.dy_dense = w.z[i]; d.dy_dense_bstride = zbs;
    RC(launch_conv_dw(ctx, kDwKid[i], L.Cin, L.ks, mode, d, n->grads + L.w_off, n->bn_scratch));
    if (i > 0) {
      ConvArgs x; memset(&x, 0, sizeof(x));
      x.in = w.z[i]; x.in_bstride = zbs; x.w = n->params + L.w_off; x.nout = L.Cin;
      x.out = w.dpool[i - 1]; x.out_bstride = (long)L.H * L.W * L.Cin;
      x.B = B; x.H = L.H; x.W = L.W;
      RC(launch_conv_fwd(ctx, kDxKid[i], kConvOut, L.ks, IN_F32_FLIP, EPI_PLAIN, x));
    }
  }
  return CPP_OK;
}

int net_backward_conv(cpp_net* n, Workspace& w, int B, const void* state, int dtype, const float* white) {
  cpp_ctx* ctx = n->ctx;
  if (n->spec.use_batch_norm) return net_backward_conv_bn(n, w, B, state, dtype, white);
  for (int i = 2; i >= 0; --i) {
    const ConvL& L = n->conv[i];
    int mode;
    ConvArgs d = conv_dw_args(n, w, i, state, dtype, white, B, &mode);
    RC(launch_conv_dw(ctx, kDwKid[i], L.Cin, L.ks, mode, d, n->grads + L.w_off, n->grads + L.b_off));
    if (i > 0)      // dX -> gradient w.r.t. the previous pooled output (conv1's input is data: no dX)
      RC(launch_conv_fwd(ctx, kDxKid[i], kConvOut, L.ks, IN_DY, EPI_PLAIN, conv_dx_args(n, w, i, B)));
  }
  return CPP_OK;
}

// the same for several networks with identical geometry, every layer's kernels batched into one launch
// The conv trunks (no batch norm) of nn same-shaped networks, one launch per layer: networks [0, first_target) will run a backward
// pass, networks [first_target, nn) are forward-only (target networks: base_network.py:35-49).
//  * conv1 of all of them is ONE launch: 16 tiles per persistent workgroup amortise the weight preload and the tail (measured
//    0.560 -> 0.526 ms per step for DDPG's four networks);
//  * a forward-only network whose conv2 reads the bf16 planes of pool1 never reads pool1's f32 copy or its arg-max codes (26 MB of
//    writes per minibatch at 64x64x18): null outputs, which the kernel skips;
//  * conv2 (bf16 pipes, 32x32 inputs) carries conv3 + pool3 as its tail when the geometry allows (conv23_fuse_ok): one launch
//    less, and the forward-only networks' pool2 never leaves LDS.
int nets_forward_trunk_fused(cpp_ctx* ctx, cpp_net* const* nets, int nn, const void* const* sts, const float* const* whs, int first_target,
                             int dt, int B) {
  if (nn < 1 || nn > CONV_BATCH_MAX) { cpp_set_error("conv trunks: batch of %d networks", nn); return CPP_ERR_ARG; }
  cpp_net* a = nets[0];
  for (int k = 0; k < nn; ++k) nets[k]->use_b16 = trunk_b16(nets[k], dt, B, 0);
  {
    ConvArgs cl[CONV_BATCH_MAX]; int mode = 0;
    for (int k = 0; k < nn; ++k) cl[k] = conv_fwd_args(nets[k], nets[k]->ws[0], 0, sts[k], dt, whs[k], B, &mode);
    for (int k = first_target; k < nn; ++k)
      if (nets[k]->use_b16 && cl[k].out_b16) { cl[k].out = nullptr; cl[k].out_amax = nullptr; }
    for (int k = 0; k < nn; ++k) nets[k]->wimg_key = nullptr;      // (consumed -- or not used: either way the next forward builds its own)
    RC(launch_conv_fwd_multi(ctx, kFwdKid[0], a->conv[0].Cin, a->conv[0].ks, mode, EPI_RELU_POOL, cl, nn));
  }
  bool fuse23 = conv23_fuse_ok(a->conv[1].H, a->conv[1].W, B, kConvOut);
  for (int k = 0; k < nn; ++k) fuse23 = fuse23 && nets[k]->use_b16;
  for (int i = 1; i < 3; ++i) {
    if (i == 2 && fuse23) break;
    ConvArgs cl[CONV_BATCH_MAX]; int mode = 0;
    for (int k = 0; k < nn; ++k) {
      cl[k] = conv_fwd_args(nets[k], nets[k]->ws[0], i, sts[k], dt, whs[k], B, &mode);
      if (i == 1 && fuse23) {
        int m3 = 0;
        const ConvArgs c3 = conv_fwd_args(nets[k], nets[k]->ws[0], 2, sts[k], dt, whs[k], B, &m3);
        cl[k].n3_w = c3.w; cl[k].n3_bias = c3.bias; cl[k].n3_out = c3.out; cl[k].n3_out_bstride = c3.out_bstride; cl[k].n3_amax = c3.out_amax;
        if (k >= first_target) { cl[k].out = nullptr; cl[k].out_amax = nullptr; }      // pool2 only feeds conv3, which reads it from LDS
      }
    }
    RC(launch_conv_fwd_multi(ctx, kFwdKid[i], a->conv[i].Cin, a->conv[i].ks, mode, EPI_RELU_POOL, cl, nn));
  }
  return CPP_OK;
}

int nets_backward_conv(cpp_ctx* ctx, cpp_net* const* nets, int nn, int B, const void* state, int dtype, const float* white) {
  if (nets[0]->spec.use_batch_norm) {                 // dense dz per layer, then dW / dX of all networks in one launch each
    for (int i = 2; i >= 0; --i) {
      const ConvL& L = nets[0]->conv[i];
      const long zbs = (long)L.H * L.W * kConvOut;
      RC(launch_bn_backward(ctx, bn_batch(nets, nn, i, B)));
      ConvArgs dl[CONV_BATCH_MAX], xl[CONV_BATCH_MAX]; float *gw[CONV_BATCH_MAX], *gb[CONV_BATCH_MAX];
      int mode = 0;
      for (int k = 0; k < nn; ++k) {
        cpp_net* n = nets[k];
        dl[k] = conv_dw_args(n, n->ws[0], i, state, dtype, white, B, &mode);
        dl[k].dy_dense = n->ws[0].z[i]; dl[k].dy_dense_bstride = zbs;
        gw[k] = n->grads + L.w_off; gb[k] = n->bn_scratch;
        if (i > 0) {
          ConvArgs& x = xl[k]; memset(&x, 0, sizeof(x));
          x.in = n->ws[0].z[i]; x.in_bstride = zbs; x.w = n->params + L.w_off; x.nout = L.Cin;
          x.out = n->ws[0].dpool[i - 1]; x.out_bstride = (long)L.H * L.W * L.Cin;
          x.B = B; x.H = L.H; x.W = L.W;
        }
      }
      RC(launch_conv_dw_multi(ctx, kDwKid[i], L.Cin, L.ks, mode, dl, nn, gw, gb));
      if (i > 0) RC(launch_conv_fwd_multi(ctx, kDxKid[i], kConvOut, L.ks, IN_F32_FLIP, EPI_PLAIN, xl, nn));
    }
    return CPP_OK;
  }
  for (int i = 2; i >= 0; --i) {
    const ConvL& L = nets[0]->conv[i];
    ConvArgs dl[CONV_BATCH_MAX], xl[CONV_BATCH_MAX]; float *gw[CONV_BATCH_MAX], *gb[CONV_BATCH_MAX];
    int mode = 0;
    for (int k = 0; k < nn; ++k) {
      dl[k] = conv_dw_args(nets[k], nets[k]->ws[0], i, state, dtype, white, B, &mode);
      gw[k] = nets[k]->grads + L.w_off; gb[k] = nets[k]->grads + L.b_off;
      if (i > 0) xl[k] = conv_dx_args(nets[k], nets[k]->ws[0], i, B);
    }
    // a layer's dW and dX leave in one launch (conv3_bwd_pair.hip, conv2_bwd_pair.hip; CPP_CONV3_PAIR=0 / CPP_CONV2_PAIR=0: two)
    static const bool no_pair3 = cpp_switch_off("CPP_CONV3_PAIR");
    static const bool no_pair2 = cpp_switch_off("CPP_CONV2_PAIR");
    const bool no_pair = i == 2 ? no_pair3 : (i == 1 ? no_pair2 : true);
    ConvPairSlot slot; slot.have_dw = slot.have_dx = false; slot.dx_rs = slot.dw_rs = false; slot.layer = i;
    if (!no_pair) ctx->pair = &slot;
    int rc = launch_conv_dw_multi(ctx, kDwKid[i], L.Cin, L.ks, mode, dl, nn, gw, gb);
    if (!rc && i > 0) rc = launch_conv_fwd_multi(ctx, kDxKid[i], kConvOut, L.ks, IN_DY, EPI_PLAIN, xl, nn);
    ctx->pair = nullptr;
    RC(rc);
    if (!no_pair) RC(i == 2 ? launch_conv3_bwd_pair(ctx, slot) : launch_conv2_bwd_pair(ctx, slot));
  }
  return CPP_OK;
}

// Backward from w.dz[last] (gradient w.r.t. the last layer's pre-activation).  want_params: write
// [dW; db] of every layer into n->grads, otherwise stop once d_action is known.  d_action: (B, A) out.
int net_backward(cpp_net* n, Workspace& w, int B, bool want_params, float* d_action,
                        const void* state, int dtype, const float* white, int start_layer) {
  cpp_ctx* ctx = n->ctx;
  const int nfc = (int)n->fc.size(), A = n->spec.action_dim;
  if (want_params && !n->grads) { cpp_set_error("network has no gradient buffer"); return CPP_ERR_STATE; }
  if (start_layer == -2) start_layer = nfc - 1;       // -1: only the conv trunk (w.dpool[2] already holds d flat)
  for (int l = start_layer; l >= 0; --l) {
    const FcL& L = n->fc[l];
    const float* dz = w.dz[l];
    const float* W = n->params + L.w_off;
    if (want_params)    // [dW; db] = [x, 1]^T dz
      RC(gemm(ctx, w.fcin[l], 1, L.n_in + 1, dz, L.n_out, 1, n->grads + L.w_off, L.n_out, L.n_in + 1, L.n_out, B, GE_NONE));
    if (L.cat) {
      if (d_action)     // dQ/da: the action columns of dz W^T (ddpg_cartpole.py:222)
        RC(gemm(ctx, dz, L.n_out, 1, W + (long)(L.n_in - A) * L.n_out, 1, L.n_out, d_action, A, B, A, L.n_out, GE_NONE));
      if (!want_params) return CPP_OK;
      if (l > 0)
        RC(gemm(ctx, dz, L.n_out, 1, W, 1, L.n_out, w.dz[l - 1], L.n_in - A, B, L.n_in - A, L.n_out,
                relu_grad_epi(n, l - 1), w.fcin[l], L.n_in + 1));
    } else if (l > 0) {
      RC(gemm(ctx, dz, L.n_out, 1, W, 1, L.n_out, w.dz[l - 1], L.n_in, B, L.n_in, L.n_out,
              relu_grad_epi(n, l - 1), w.fcin[l], L.n_in + 1));
    } else if (n->spec.pixel && want_params) {
      RC(gemm(ctx, dz, L.n_out, 1, W, 1, L.n_out, w.dpool[2], n->flat, B, n->flat, L.n_out, GE_NONE));
    }
  }
  if (!want_params || !n->spec.pixel) return CPP_OK;
  DwPendingGuard pending(ctx);      // (a failure below drops what was queued)
  RC(net_backward_conv(n, w, B, state, dtype, white));
  return flush_dw_reduce(ctx);
}


GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc,
                        int M, int N, int K, int epi, const float* Y, long ldy) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.sAm = sAm; g.sAk = sAk; g.B = Bm; g.sBk = sBk; g.sBn = sBn; g.C = C; g.ldc = ldc;
  g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.epi = epi;
  return g;
}
// --use-dropout: a training-mode forward draws its keep bits from (seed, layer, the network's forward count); inference
// mode is the plain ReLU.  The backward pass of such a layer doubles what it lets through (Y > 0 <=> kept and active).
void set_dropout(GemmArgs& g, cpp_net* n, int l) {
  if (g.epi != GE_RELU_DROPOUT) return;
  if (n->is_training && n->drop_counter) { g.drop_counter = n->drop_counter; g.drop_seed = n->spec.dropout_seed; g.drop_layer = (uint32_t)l; }
  else g.epi = GE_RELU;
}
int relu_grad_epi(const cpp_net* n, int producer_layer) {
  return (producer_layer >= 0 && n->fc[producer_layer].act == GE_RELU_DROPOUT) ? GE_MUL_RELU_GRAD_X2 : GE_MUL_RELU_GRAD;
}
int bump_dropout(cpp_net* n) {      // after every training-mode forward of the network's FC stack
  if (!n->drop_counter || !n->is_training) return CPP_OK;
  return launch_counter_add(n->ctx, n->drop_counter, 1);
}
// y = act([x, 1] [W; b]) of layer l into the next layer's input buffer (or w.out for the last layer)
GemmArgs fc_fwd_args(cpp_net* n, Workspace& w, int l, int B) {
  const FcL& L = n->fc[l];
  const int nfc = (int)n->fc.size();
  float* C = (l + 1 < nfc) ? w.fcin[l + 1] : w.out;
  const long ldc = (l + 1 < nfc) ? n->fc[l + 1].n_in + 1 : L.n_out;
  GemmArgs g = mk_gemm(w.fcin[l], L.n_in + 1, 1, n->params + L.w_off, L.n_out, 1, C, ldc, B, L.n_out, L.n_in + 1, L.act);
  set_dropout(g, n, l);
  return g;
}
// [dW; db] = [x, 1]^T dz
GemmArgs fc_dw_args(cpp_net* n, Workspace& w, int l, int B, const float* dz) {
  const FcL& L = n->fc[l];
  return mk_gemm(w.fcin[l], 1, L.n_in + 1, dz, L.n_out, 1, n->grads + L.w_off, L.n_out, L.n_in + 1, L.n_out, B, GE_NONE);
}
// columns [col0, col0+ncols) of dz W^T, optionally times relu'(Y)
GemmArgs fc_dx_args(cpp_net* n, int l, int B, const float* dz, long dz_ld, int col0, int ncols, float* C, long ldc,
                           int epi, const float* Y, long ldy) {
  const FcL& L = n->fc[l];
  return mk_gemm(dz, dz_ld, 1, n->params + L.w_off + (long)col0 * L.n_out, 1, L.n_out, C, ldc, B, ncols, L.n_out, epi, Y, ldy);
}

// whitening statistics of a device-resident (B, H*W*C) batch -> white[2][C]
int batch_stats(cpp_ctx* ctx, const void* s0, const void* s1, int dtype, long elems, int B, int C,
                       double* part, float* white) {
  int g = 8, c = C; while (c) { int t = g % c; g = c; c = t; }     // gcd(8, C)
  const bool vec = (elems % 8 == 0) && (C / g <= 16);
  const int nw = s1 ? 2 : 1;
  if (vec) {
    GatherArgs a; memset(&a, 0, sizeof(a));
    a.store[0] = s0; a.store[1] = s1 ? s1 : s0; a.part = part; a.elems = elems; a.B = B; a.C = C;
    RC(launch_gather_stats(ctx, a, dtype));     // grid (B,2): second column recomputes s0 when s1 == NULL (cheap, rare)
    RC(launch_stats_finalize(ctx, part, B, nw, C, (double)B * (double)(elems / C), white));
  } else {
    RC(launch_stats_generic(ctx, s0, dtype, (long)B * (elems / C), C, white));
    if (s1) RC(launch_stats_generic(ctx, s1, dtype, (long)B * (elems / C), C, white + 2 * C));
  }
  return CPP_OK;
}

extern "C" int cpp_net_forward(cpp_net* n, const void* state, int state_dtype, int B, const float* action, float* out) {
  ARG_CHECK(n && state && out, "cpp_net_forward: NULL argument");
  ARG_CHECK(B >= 1 && B <= n->maxB, "cpp_net_forward: batch %d outside [1,%d]", B, n->maxB);
  ARG_CHECK(state_dtype == CPP_F32 || state_dtype == CPP_F16, "cpp_net_forward: dtype %d", state_dtype);
  ARG_CHECK(n->spec.kind != CPP_CRITIC || action, "cpp_net_forward: critic needs an action batch");
  cpp_ctx* ctx = n->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  const int A = n->spec.action_dim, no = n->fc.back().n_out;
  if (!n->stage_state) {
    RC(n->arena.alloc(&n->stage_state, (size_t)n->maxB * n->state_elems * sizeof(float), false));
    RC(dalloc(n->arena, &n->stage_action, (size_t)n->maxB * A));
  }
  const size_t esz = state_dtype == CPP_F16 ? 2 : 4;
  HIP_CHECK(hipMemcpyAsync(n->stage_state, state, (size_t)B * n->state_elems * esz, hipMemcpyHostToDevice, ctx->stream));
  if (action) HIP_CHECK(hipMemcpyAsync(n->stage_action, action, (size_t)B * A * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (n->spec.pixel)
    RC(batch_stats(ctx, n->stage_state, nullptr, state_dtype, n->state_elems, B, n->spec.C, n->stats_part, n->white));
  n->is_training = false;                              // IS_TRAINING: False (ddpg_cartpole.py:125)
  int frc = net_forward_trunk(n, n->ws[0], n->stage_state, state_dtype, n->white, B);
  if (!frc) frc = net_forward_fc(n, n->ws[0], 0, B, action ? n->stage_action : nullptr);
  n->is_training = true;
  if (frc) return frc;
  HIP_CHECK(hipMemcpyAsync(out, n->ws[0].out, (size_t)B * no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx_sync_stream(ctx));
  return CPP_OK;
}

// B independent action_given calls in one pass (rollout-side inference for many env workers): every
// image is whitened with ITS OWN statistics, exactly as B separate batches of one would be (base_network.py:95-99
// at B = 1); everything after the whitening is row-local anyway.
extern "C" int cpp_net_forward_each(cpp_net* n, const void* state, int state_dtype, int B, const float* action, float* out) {
  ARG_CHECK(n && state && out, "cpp_net_forward_each: NULL argument");
  ARG_CHECK(B >= 1 && B <= n->maxB, "cpp_net_forward_each: batch %d outside [1,%d]", B, n->maxB);
  ARG_CHECK(state_dtype == CPP_F32 || state_dtype == CPP_F16, "cpp_net_forward_each: dtype %d", state_dtype);
  ARG_CHECK(n->spec.kind != CPP_CRITIC || action, "cpp_net_forward_each: critic needs an action batch");
  cpp_ctx* ctx = n->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  const int A = n->spec.action_dim, no = n->fc.back().n_out, C = n->spec.C;
  if (!n->stage_state) {
    RC(n->arena.alloc(&n->stage_state, (size_t)n->maxB * n->state_elems * sizeof(float), false));
    RC(dalloc(n->arena, &n->stage_action, (size_t)n->maxB * A));
  }
  const size_t esz = state_dtype == CPP_F16 ? 2 : 4;
  HIP_CHECK(hipMemcpyAsync(n->stage_state, state, (size_t)B * n->state_elems * esz, hipMemcpyHostToDevice, ctx->stream));
  if (action) HIP_CHECK(hipMemcpyAsync(n->stage_action, action, (size_t)B * A * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  long wbs = 0;
  if (n->spec.pixel) {
    int g = 8, c = C; while (c) { int t = g % c; g = c; c = t; }
    const long npix = n->state_elems / C;
    if (n->state_elems % 8 == 0 && C / g <= 16) {       // per-row partial sums, finalised row by row
      GatherArgs ga; memset(&ga, 0, sizeof(ga));
      ga.store[0] = n->stage_state; ga.store[1] = n->stage_state; ga.part = n->stats_part; ga.elems = n->state_elems; ga.B = B; ga.C = C;
      RC(launch_gather_stats(ctx, ga, state_dtype));
      RC(launch_stats_finalize(ctx, n->stats_part, 1, B, C, (double)npix, n->white_rows));
    } else {
      for (int b = 0; b < B; ++b)
        RC(launch_stats_generic(ctx, (const char*)n->stage_state + (size_t)b * n->state_elems * esz, state_dtype, npix, C,
                                n->white_rows + (long)b * 2 * C));
    }
    wbs = 2 * C;
  }
  n->is_training = false;
  int frc = net_forward_trunk(n, n->ws[0], n->stage_state, state_dtype, n->white_rows, B, wbs);
  if (!frc) frc = net_forward_fc(n, n->ws[0], 0, B, action ? n->stage_action : nullptr);
  n->is_training = true;
  if (frc) return frc;
  HIP_CHECK(hipMemcpyAsync(out, n->ws[0].out, (size_t)B * no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx_sync_stream(ctx));
  return CPP_OK;
}

extern "C" int cpp_net_get_pool(cpp_net* n, int which, int B, float* out) {
  ARG_CHECK(n && out, "cpp_net_get_pool: NULL argument");
  ARG_CHECK(B >= 1 && B <= n->maxB, "cpp_net_get_pool: batch %d", B);
  if (n->spec.pixel && which >= 11 && which <= 13) {   // debug: arg-max codes (0..3) of the 2x2 windows, as floats
    const ConvL& L = n->conv[which - 11];
    const size_t cnt = (size_t)B * L.Hp * L.Wp * kConvOut;
    std::vector<uint8_t> tmp(cnt);
    HIP_CHECK(hipMemcpyAsync(tmp.data(), n->ws[0].amax[which - 11], cnt, hipMemcpyDeviceToHost, n->ctx->stream));
    HIP_CHECK(ctx_sync_stream(n->ctx));
    for (size_t i = 0; i < cnt; ++i) out[i] = (float)(tmp[i] & 3);      // (bit 2 of the byte: "the pooled output is > 0", conv_kyo.h POOL_ACTIVE)
    return CPP_OK;
  }
  if (n->spec.pixel && which >= 21 && which <= 22) {   // debug: the pooled-gradient buffers of the last backward pass (dX of conv2 / conv3)
    const ConvL& L = n->conv[which - 21];
    ARG_CHECK(n->ws[0].dpool[which - 21], "cpp_net_get_pool: no gradient workspace");
    const size_t cnt = (size_t)B * L.Hp * L.Wp * kConvOut;
    HIP_CHECK(hipMemcpyAsync(out, n->ws[0].dpool[which - 21], cnt * sizeof(float), hipMemcpyDeviceToHost, n->ctx->stream));
    HIP_CHECK(ctx_sync_stream(n->ctx));
    return CPP_OK;
  }
  ARG_CHECK(n->spec.pixel && which >= 1 && which <= 3, "cpp_net_get_pool: which=%d (pixel nets, 1..3)", which);
  const ConvL& L = n->conv[which - 1];
  const size_t row = (size_t)L.Hp * L.Wp * kConvOut * sizeof(float);
  const size_t spitch = (which == 3) ? ((size_t)n->flat + 1) * sizeof(float) : row;
  HIP_CHECK(hipMemcpy2DAsync(out, row, n->ws[0].pool[which - 1], spitch, row, B, hipMemcpyDeviceToHost, n->ctx->stream));
  HIP_CHECK(ctx_sync_stream(n->ctx));
  return CPP_OK;
}

