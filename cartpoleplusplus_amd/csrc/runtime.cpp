// C-ABI implementation (include/cartpolepp_abi.h): handles, device memory, and the launch sequences
// of the DDPG-from-pixels hot path.  Host-side logic only; all arithmetic is in the HIP kernels.
#include "../../include/cartpolepp_abi.h"
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------
// errors / profiling brackets
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void cpp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* cpp_last_error(void) { return g_err; }
extern "C" int cpp_abi_version(void) { return CPP_ABI_VERSION; }

#define ARG_CHECK(cond, ...)                \
  do {                                      \
    if (!(cond)) {                          \
      cpp_set_error(__VA_ARGS__);           \
      return CPP_ERR_ARG;                   \
    }                                       \
  } while (0)
#define RC(expr)                  \
  do {                            \
    int _rc = (expr);             \
    if (_rc) return _rc;          \
  } while (0)

void prof_begin(cpp_ctx* ctx) {
  if (ctx->prof) (void)hipEventRecord(ctx->pe0, ctx->stream);
}
void prof_end(cpp_ctx* ctx, int kid) {
  if (!ctx->prof) return;
  if (ctx->pair && ((ctx->pair->layer == 2 && (kid == K_CONV3_DW || kid == K_CONV3_DX)) ||
                    (ctx->pair->layer == 1 && (kid == K_CONV2_DW || kid == K_CONV2_DX)))) return;      // parked, not launched (conv*_bwd_pair.hip)
  (void)hipEventRecord(ctx->pe1, ctx->stream);
  (void)hipEventSynchronize(ctx->pe1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ctx->pe0, ctx->pe1);
  ctx->prof_ms[kid] += ms;
  ctx->prof_n[kid] += 1;
}

static const char* kKernelNames[K_NUM_KERNELS] = {
    "gather_stats", "stats_finalize", "stats_generic", "conv1_fwd", "conv2_fwd", "conv3_fwd",
    "conv1_dw", "conv2_dw", "conv3_dw", "conv2_dx", "conv3_dx", "dw_reduce", "gemm", "elementwise",
    "td", "sumsq", "clip_sgd", "soft_update", "replay_fill", "naf_head", "conv1_fwd_f16x3", "conv1_dw_f16x3", "heads", "conv3_bwd", "conv2_bwd", "reduce_gather", "conv1_dw_gather"};

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int cpp_ctx_create(int device_id, void* hip_stream, cpp_ctx** out) {
  ARG_CHECK(out, "cpp_ctx_create: out is NULL");
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  ARG_CHECK(device_id >= 0 && device_id < ndev, "cpp_ctx_create: device %d not in [0,%d)", device_id, ndev);
  HIP_CHECK(hipSetDevice(device_id));
  cpp_ctx* c = new cpp_ctx();
  memset(c, 0, sizeof(*c));
  c->device = device_id;
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else { HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  HIP_CHECK(hipEventCreate(&c->t0));
  HIP_CHECK(hipEventCreate(&c->t1));
  HIP_CHECK(hipEventCreate(&c->pe0));
  HIP_CHECK(hipEventCreate(&c->pe1));
  HIP_CHECK(hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device_id));
  *out = c;
  return CPP_OK;
}

extern "C" int cpp_ctx_destroy(cpp_ctx* c) {
  if (!c) return CPP_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipEventDestroy(c->t0); (void)hipEventDestroy(c->t1);
  (void)hipEventDestroy(c->pe0); (void)hipEventDestroy(c->pe1);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return CPP_OK;
}

extern "C" int cpp_sync(cpp_ctx* c) {
  ARG_CHECK(c, "cpp_sync: ctx is NULL");
  HIP_CHECK(hipStreamSynchronize(c->stream));
  return CPP_OK;
}

extern "C" int cpp_timer_begin(cpp_ctx* c) {
  ARG_CHECK(c, "ctx is NULL");
  HIP_CHECK(hipEventRecord(c->t0, c->stream));
  return CPP_OK;
}
extern "C" int cpp_timer_end(cpp_ctx* c, float* ms) {
  ARG_CHECK(c && ms, "ctx/ms is NULL");
  HIP_CHECK(hipEventRecord(c->t1, c->stream));
  HIP_CHECK(hipEventSynchronize(c->t1));
  HIP_CHECK(hipEventElapsedTime(ms, c->t0, c->t1));
  return CPP_OK;
}
extern "C" int cpp_prof_enable(cpp_ctx* c, int on) { ARG_CHECK(c, "ctx is NULL"); c->prof = on != 0; return CPP_OK; }
extern "C" int cpp_prof_reset(cpp_ctx* c) {
  ARG_CHECK(c, "ctx is NULL");
  memset(c->prof_ms, 0, sizeof(c->prof_ms)); memset(c->prof_n, 0, sizeof(c->prof_n));
  return CPP_OK;
}
extern "C" int cpp_prof_num_kernels(void) { return K_NUM_KERNELS; }
extern "C" const char* cpp_prof_kernel_name(int k) { return (k >= 0 && k < K_NUM_KERNELS) ? kKernelNames[k] : ""; }
extern "C" int cpp_prof_read(cpp_ctx* c, int k, double* total_ms, int64_t* launches) {
  ARG_CHECK(c && k >= 0 && k < K_NUM_KERNELS, "cpp_prof_read: bad kernel id %d", k);
  if (total_ms) *total_ms = c->prof_ms[k];
  if (launches) *launches = c->prof_n[k];
  return CPP_OK;
}

// ---------------------------------------------------------------------------------------------
// device memory helper
// ---------------------------------------------------------------------------------------------
struct Arena {
  std::vector<void*> ptrs;
  hipStream_t stream = nullptr;     // zero-fills are ordered on the owning ctx's stream (never the null stream)
  // every allocation sits between two 256-byte guard bands: the conv1 operand loads (conv_k16.h) read up to 128 bytes
  // before and 256 after an image batch (masked out, but the addresses must be mapped)
  static constexpr size_t GUARD = 256;
  int alloc(void** p, size_t bytes, bool zero = true) {
    if (bytes == 0) bytes = 16;
    void* raw = nullptr;
    HIP_CHECK(hipMalloc(&raw, bytes + 2 * GUARD));
    ptrs.push_back(raw);
    *p = (char*)raw + GUARD;
    if (zero) HIP_CHECK(hipMemsetAsync(raw, 0, bytes + 2 * GUARD, stream));
    return 0;
  }
  void release() { for (void* p : ptrs) (void)hipFree(p); ptrs.clear(); }
};
template <typename T> static int dalloc(Arena& a, T** p, size_t count, bool zero = true) {
  return a.alloc((void**)p, count * sizeof(T), zero);
}

// ---------------------------------------------------------------------------------------------
// networks
// ---------------------------------------------------------------------------------------------
static const int kConvKs[3] = {5, 5, 3};          // base_network.py:103,111,119
static const int kConvOut = 10;
static const char* kConvNames[3] = {"conv1", "conv2", "conv3"};

struct ConvL { int H, W, Cin, ks, Hp, Wp; long w_off, b_off; };
struct FcL { int n_in, n_out, act, cat; long w_off; std::string name; };   // bias row at w_off + n_in*n_out
struct VarInfo { std::string name; int rank; int shape[4]; long offset; };

struct Workspace {
  float* pool[3] = {nullptr, nullptr, nullptr};
  unsigned short* pool_b16 = nullptr;      // pool[0] once more as three bf16 planes (conv2 forward on the bf16 pipes)
  uint8_t* amax[3] = {nullptr, nullptr, nullptr};
  float* dpool[3] = {nullptr, nullptr, nullptr};
  // batch norm (training mode): plain conv output (overwritten by its gradient in the backward pass), (inv, -mean*inv)
  float* z[3] = {nullptr, nullptr, nullptr};
  float* bn_stat[3] = {nullptr, nullptr, nullptr};
  std::vector<float*> fcin, dz;
  float* out = nullptr;
};

struct cpp_net {
  cpp_ctx* ctx; cpp_net_spec spec; int maxB;
  std::vector<ConvL> conv; std::vector<FcL> fc; std::vector<VarInfo> vars;
  long nparams; int flat; int cat_layer; long state_elems;
  float* params; float* grads; float* own_grads;
  Workspace ws[2];
  bool use_b16;             // this forward: conv1 (f16 pipes) leaves bf16 planes of pool1, conv2 forward reads them
  const int32_t* img_slot;  // conv1 reads image b from row img_slot[b] of the state pointer (the replay store); nullptr: b
  float* white;            // [2][C] statistics for cpp_net_forward
  float* white_rows;       // [maxB][2][C]: per-image statistics for cpp_net_forward_each
  double* stats_part;      // [maxB][2C]
  float* dw_partial[3];     // one per conv layer: their reductions are deferred and batched
  bool is_training;         // base_network.IS_TRAINING for the next forward (batch norm and dropout look at it)
  uint64_t* drop_counter;   // dropout: number of training-mode forwards so far (device; part of the Philox counter)
  double* bn_part; float* bn_means; float* bn_scratch;   // batch norm: reduction partials, (mean dy, mean dy*zhat), dW bias-slot dump
  void* stage_state; float* stage_action; float* stage_out;
  Arena arena;
};

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
  n->img_slot = nullptr; n->use_b16 = false;
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
    for (int i = 0; i < 3; ++i) { n->ws[1].pool[i] = n->ws[0].pool[i]; n->ws[1].amax[i] = n->ws[0].amax[i]; n->ws[1].dpool[i] = n->ws[0].dpool[i];
                                  n->ws[1].z[i] = n->ws[0].z[i]; n->ws[1].bn_stat[i] = n->ws[0].bn_stat[i]; }
  }
  if (spec->pixel) {
    for (int i = 0; i < 3; ++i)
      if ((rc = dalloc(n->arena, &n->dw_partial[i], conv_dw_partial_floats(ctx, n->conv[i].Cin, n->conv[i].ks, kConvOut)))) return fail(rc);
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
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  *out = n;
  return CPP_OK;
}

extern "C" int cpp_net_destroy(cpp_net* n) {
  if (!n) return CPP_OK;
  (void)hipSetDevice(n->ctx->device);
  (void)hipStreamSynchronize(n->ctx->stream);
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
  HIP_CHECK(hipMemcpyAsync(n->params, host, cnt * sizeof(float), hipMemcpyHostToDevice, n->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  return CPP_OK;
}
extern "C" int cpp_net_get_params(cpp_net* n, float* host, int64_t cnt) {
  ARG_CHECK(n && host, "cpp_net_get_params: NULL argument");
  ARG_CHECK(cnt == n->nparams, "cpp_net_get_params: asked %ld values, network has %ld", (long)cnt, n->nparams);
  HIP_CHECK(hipMemcpyAsync(host, n->params, cnt * sizeof(float), hipMemcpyDeviceToHost, n->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  return CPP_OK;
}
extern "C" int cpp_net_get_grads(cpp_net* n, float* host, int64_t cnt) {
  ARG_CHECK(n && host, "cpp_net_get_grads: NULL argument");
  ARG_CHECK(cnt == n->nparams, "cpp_net_get_grads: asked %ld values, network has %ld", (long)cnt, n->nparams);
  if (!n->grads) { cpp_set_error("cpp_net_get_grads: network has no train op (init_ops_for_training not called)"); return CPP_ERR_STATE; }
  HIP_CHECK(hipMemcpyAsync(host, n->grads, cnt * sizeof(float), hipMemcpyDeviceToHost, n->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  return CPP_OK;
}

extern "C" int cpp_net_soft_update(cpp_net* target, const cpp_net* source, float coeff) {
  ARG_CHECK(target && source, "cpp_net_soft_update: NULL argument");
  ARG_CHECK(coeff >= 0.f && coeff <= 1.f, "affine_combo_coeff %g outside [0,1]", coeff);    // base_network.py:22
  ARG_CHECK(target->nparams == source->nparams, "cpp_net_soft_update: shapes differ (%ld vs %ld)",
            target->nparams, source->nparams);                                             // base_network.py:30
  return launch_soft_update(target->ctx, target->params, source->params, target->nparams, nullptr, nullptr, 0, coeff);
}

// --- launch sequences -------------------------------------------------------------------------
static int gemm(cpp_ctx* ctx, const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn,
                float* C, long ldc, int M, int N, int K, int epi, const float* Y = nullptr, long ldy = 0,
                int accumulate = 0) {
  GemmArgs g; memset(&g, 0, sizeof(g)); g.accumulate = accumulate; g.A = A; g.sAm = sAm; g.sAk = sAk; g.B = Bm; g.sBk = sBk; g.sBn = sBn; g.C = C; g.ldc = ldc;
  g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.epi = epi;
  return launch_gemm(ctx, g);
}

static const int kFwdKid[3] = {K_CONV1_FWD, K_CONV2_FWD, K_CONV3_FWD};
static const int kDwKid[3] = {K_CONV1_DW, K_CONV2_DW, K_CONV3_DW};
static const int kDxKid[3] = {-1, K_CONV2_DX, K_CONV3_DX};

// launch descriptors of conv layer i of a network (forward, dW, dX)
static ConvArgs conv_fwd_args(cpp_net* n, Workspace& w, int i, const void* state, int dtype, const float* white, int B, int* mode,
                              long white_bstride = 0) {
  const ConvL& L = n->conv[i];
  ConvArgs a; memset(&a, 0, sizeof(a));
  if (i == 0) { a.in = state; a.in_bstride = n->state_elems; a.scale = white; a.shift = white + n->spec.C; a.white_bstride = white_bstride;
                a.img_slot = n->img_slot;
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
static void conv_dy_desc(cpp_net* n, Workspace& w, int i, ConvArgs& a, int B) {
  const ConvL& L = n->conv[i];
  a.dy.dpool = w.dpool[i]; a.dy.pool = w.pool[i]; a.dy.amax = w.amax[i];
  a.dy.dpool_bstride = (i == 2) ? (long)n->flat : (long)L.Hp * L.Wp * kConvOut;
  a.dy.pool_bstride = (i == 2) ? (long)n->flat + 1 : (long)L.Hp * L.Wp * kConvOut;
  a.dy.Hp = L.Hp; a.dy.Wp = L.Wp;
  a.B = B; a.H = L.H; a.W = L.W;
}
static ConvArgs conv_dw_args(cpp_net* n, Workspace& w, int i, const void* state, int dtype, const float* white, int B, int* mode) {
  const ConvL& L = n->conv[i];
  ConvArgs d; memset(&d, 0, sizeof(d));
  conv_dy_desc(n, w, i, d, B);
  if (i == 0) { d.in = state; d.in_bstride = n->state_elems; d.scale = white; d.shift = white + n->spec.C; d.img_slot = n->img_slot;
                *mode = dtype == CPP_F16 ? IN_F16_WHITEN : IN_F32_WHITEN; }
  else { d.in = w.pool[i - 1]; d.in_bstride = (long)L.H * L.W * L.Cin; *mode = IN_F32_PLAIN; }
  d.nout = kConvOut; d.partial = n->dw_partial[i];
  return d;
}
static ConvArgs conv_dx_args(cpp_net* n, Workspace& w, int i, int B) {
  const ConvL& L = n->conv[i];
  ConvArgs x; memset(&x, 0, sizeof(x));
  conv_dy_desc(n, w, i, x, B);
  x.w = n->params + L.w_off; x.nout = L.Cin;
  x.out = w.dpool[i - 1]; x.out_bstride = (long)L.H * L.W * L.Cin;
  return x;
}

// slim.batch_norm's epsilon; the moving variance stays at its initial 1 (never updated by the reference's train ops)
static const double kBnEps = 1e-3;

// descriptor of layer i of a batch-norm network for the bn.hip launches
static BnNet bn_net_desc(cpp_net* n, Workspace& w, int i) {
  const ConvL& L = n->conv[i];
  BnNet d; memset(&d, 0, sizeof(d));
  d.z = w.z[i]; d.stat = w.bn_stat[i]; d.beta = n->params + L.b_off;
  d.pool = w.pool[i]; d.pool_bstride = (i == 2) ? (long)n->flat + 1 : (long)L.Hp * L.Wp * kConvOut; d.amax = w.amax[i];
  d.dpool = w.dpool[i]; d.dpool_bstride = (i == 2) ? (long)n->flat : (long)L.Hp * L.Wp * kConvOut;
  d.part = n->bn_part; d.means = n->bn_means; d.dbeta = n->grads ? n->grads + L.b_off : nullptr;
  return d;
}
static BnBatch bn_batch(cpp_net* const* nets, int nn, int i, int B) {
  BnBatch bb; memset(&bb, 0, sizeof(bb));
  const ConvL& L = nets[0]->conv[i];
  bb.count = nn; bb.B = B; bb.H = L.H; bb.W = L.W; bb.C = kConvOut;
  for (int k = 0; k < nn; ++k) bb.n[k] = bn_net_desc(nets[k], nets[k]->ws[0], i);
  return bb;
}

// conv trunk (pixel) or state conversion (low-dim) into ws.fcin[0]
// conv1 on the f16 pipes can leave bf16 planes of pool1 for a conv2 forward on the bf16 pipes (same launch sequence only)
static bool trunk_b16(const cpp_net* n, int dtype, int B, long white_bstride) {
  return n->spec.pixel && !n->spec.use_batch_norm && dtype == CPP_F16 && white_bstride == 0 &&
         conv12_b16_ok(n->conv[0].Cin, n->conv[0].H, n->conv[0].W, B);
}

static int net_forward_trunk(cpp_net* n, Workspace& w, const void* state, int dtype, const float* white, int B,
                             long white_bstride = 0) {
  cpp_ctx* ctx = n->ctx;
  n->use_b16 = trunk_b16(n, dtype, B, white_bstride);
  if (!n->spec.pixel)
    return launch_state_to_f32(ctx, w.fcin[0], n->fc[0].n_in + 1, state, dtype, n->state_elems, B);
  for (int i = 0; i < 3; ++i) {
    int mode;
    ConvArgs a = conv_fwd_args(n, w, i, state, dtype, white, B, &mode, white_bstride);
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
static int nets_forward_trunk_bn(cpp_ctx* ctx, cpp_net* const* nets, int nn, const void* const* states, const float* const* whites,
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
static GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc,
                        int M, int N, int K, int epi, const float* Y, long ldy);
static GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc,
                        int M, int N, int K, int epi) { return mk_gemm(A, sAm, sAk, Bm, sBk, sBn, C, ldc, M, N, K, epi, nullptr, 0); }
static void set_dropout(GemmArgs& g, cpp_net* n, int l);
static int relu_grad_epi(const cpp_net* n, int producer_layer);
static int bump_dropout(cpp_net* n);

static int net_forward_fc(cpp_net* n, Workspace& w, int from, int B, const float* action) {
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

static int net_backward_conv(cpp_net* n, Workspace& w, int B, const void* state, int dtype, const float* white) {
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
static int nets_backward_conv(cpp_ctx* ctx, cpp_net* const* nets, int nn, int B, const void* state, int dtype, const float* white) {
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
    static const bool no_pair3 = getenv("CPP_CONV3_PAIR") != nullptr && atoi(getenv("CPP_CONV3_PAIR")) == 0;
    static const bool no_pair2 = getenv("CPP_CONV2_PAIR") != nullptr && atoi(getenv("CPP_CONV2_PAIR")) == 0;
    const bool no_pair = i == 2 ? no_pair3 : (i == 1 ? no_pair2 : true);
    ConvPairSlot slot; slot.have_dw = slot.have_dx = false; slot.layer = i;
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
static int net_backward(cpp_net* n, Workspace& w, int B, bool want_params, float* d_action,
                        const void* state, int dtype, const float* white, int start_layer = -2) {
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
  RC(net_backward_conv(n, w, B, state, dtype, white));
  return flush_dw_reduce(ctx);
}


// ---------------------------------------------------------------------------------------------
// Level-synchronous launch scheduler for the fused step.  The MLP heads are ~36 tiny, latency-bound
// GEMMs per minibatch; most of them are mutually independent (four networks' forwards, dW vs dX of one
// layer, the actor's and the critic's backward chains).  Ops declare their dependencies; each round
// launches every ready op, with all ready GEMMs sharing ONE launch (gemm_batch_kernel).  Everything stays
// on the ctx stream, so the order is also what a hipGraph capture records.
// ---------------------------------------------------------------------------------------------
struct OpGraph {
  struct Op { bool is_gemm; GemmArgs g; std::function<int()> fn; std::vector<int> deps; bool done; };
  std::vector<Op> ops;
  int gemm(const GemmArgs& g, std::initializer_list<int> deps) {
    Op o; o.is_gemm = true; o.g = g; o.done = false;
    for (int d : deps) if (d >= 0) o.deps.push_back(d);
    ops.push_back(o); return (int)ops.size() - 1;
  }
  int fn(std::function<int()> f, std::initializer_list<int> deps) {
    Op o; o.is_gemm = false; o.fn = f; o.done = false; memset(&o.g, 0, sizeof(o.g));
    for (int d : deps) if (d >= 0) o.deps.push_back(d);
    ops.push_back(o); return (int)ops.size() - 1;
  }
  int run(cpp_ctx* ctx) {
    size_t remaining = ops.size();
    std::vector<int> ready; std::vector<GemmArgs> batch;
    while (remaining) {
      ready.clear(); batch.clear();
      for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].done) continue;
        bool ok = true;
        for (int d : ops[i].deps) if (!ops[d].done) { ok = false; break; }
        if (ok) ready.push_back((int)i);
      }
      if (ready.empty()) { cpp_set_error("OpGraph: dependency cycle"); return CPP_ERR_STATE; }
      static const bool dbg = getenv("CPP_OPGRAPH_DEBUG") != nullptr;
      if (dbg) {
        fprintf(stderr, "[opgraph] level:");
        for (int i : ready) {
          if (ops[i].is_gemm) fprintf(stderr, " gemm#%d(M%d N%d K%d e%d)", i, ops[i].g.M, ops[i].g.N, ops[i].g.K, ops[i].g.epi);
          else fprintf(stderr, " fn#%d", i);
        }
        fprintf(stderr, "\n");
      }
      for (int i : ready) if (!ops[i].is_gemm) RC(ops[i].fn());
      for (int i : ready) if (ops[i].is_gemm) batch.push_back(ops[i].g);
      if (!batch.empty()) RC(launch_gemm_batch(ctx, batch.data(), (int)batch.size()));
      for (int i : ready) ops[i].done = true;
      remaining -= ready.size();
    }
    return CPP_OK;
  }
};

static GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc,
                        int M, int N, int K, int epi, const float* Y, long ldy) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.sAm = sAm; g.sAk = sAk; g.B = Bm; g.sBk = sBk; g.sBn = sBn; g.C = C; g.ldc = ldc;
  g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.epi = epi;
  return g;
}
// --use-dropout: a training-mode forward draws its keep bits from (seed, layer, the network's forward count); inference
// mode is the plain ReLU.  The backward pass of such a layer doubles what it lets through (Y > 0 <=> kept and active).
static void set_dropout(GemmArgs& g, cpp_net* n, int l) {
  if (g.epi != GE_RELU_DROPOUT) return;
  if (n->is_training && n->drop_counter) { g.drop_counter = n->drop_counter; g.drop_seed = n->spec.dropout_seed; g.drop_layer = (uint32_t)l; }
  else g.epi = GE_RELU;
}
static int relu_grad_epi(const cpp_net* n, int producer_layer) {
  return (producer_layer >= 0 && n->fc[producer_layer].act == GE_RELU_DROPOUT) ? GE_MUL_RELU_GRAD_X2 : GE_MUL_RELU_GRAD;
}
static int bump_dropout(cpp_net* n) {      // after every training-mode forward of the network's FC stack
  if (!n->drop_counter || !n->is_training) return CPP_OK;
  return launch_counter_add(n->ctx, n->drop_counter, 1);
}
// y = act([x, 1] [W; b]) of layer l into the next layer's input buffer (or w.out for the last layer)
static GemmArgs fc_fwd_args(cpp_net* n, Workspace& w, int l, int B) {
  const FcL& L = n->fc[l];
  const int nfc = (int)n->fc.size();
  float* C = (l + 1 < nfc) ? w.fcin[l + 1] : w.out;
  const long ldc = (l + 1 < nfc) ? n->fc[l + 1].n_in + 1 : L.n_out;
  GemmArgs g = mk_gemm(w.fcin[l], L.n_in + 1, 1, n->params + L.w_off, L.n_out, 1, C, ldc, B, L.n_out, L.n_in + 1, L.act);
  set_dropout(g, n, l);
  return g;
}
// [dW; db] = [x, 1]^T dz
static GemmArgs fc_dw_args(cpp_net* n, Workspace& w, int l, int B, const float* dz) {
  const FcL& L = n->fc[l];
  return mk_gemm(w.fcin[l], 1, L.n_in + 1, dz, L.n_out, 1, n->grads + L.w_off, L.n_out, L.n_in + 1, L.n_out, B, GE_NONE);
}
// columns [col0, col0+ncols) of dz W^T, optionally times relu'(Y)
static GemmArgs fc_dx_args(cpp_net* n, int l, int B, const float* dz, long dz_ld, int col0, int ncols, float* C, long ldc,
                           int epi, const float* Y, long ldy) {
  const FcL& L = n->fc[l];
  return mk_gemm(dz, dz_ld, 1, n->params + L.w_off + (long)col0 * L.n_out, 1, L.n_out, C, ldc, B, ncols, L.n_out, epi, Y, ldy);
}

// whitening statistics of a device-resident (B, H*W*C) batch -> white[2][C]
static int batch_stats(cpp_ctx* ctx, const void* s0, const void* s1, int dtype, long elems, int B, int C,
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
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
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
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
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
    HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
    for (size_t i = 0; i < cnt; ++i) out[i] = (float)tmp[i];
    return CPP_OK;
  }
  ARG_CHECK(n->spec.pixel && which >= 1 && which <= 3, "cpp_net_get_pool: which=%d (pixel nets, 1..3)", which);
  const ConvL& L = n->conv[which - 1];
  const size_t row = (size_t)L.Hp * L.Wp * kConvOut * sizeof(float);
  const size_t spitch = (which == 3) ? ((size_t)n->flat + 1) * sizeof(float) : row;
  HIP_CHECK(hipMemcpy2DAsync(out, row, n->ws[0].pool[which - 1], spitch, row, B, hipMemcpyDeviceToHost, n->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(n->ctx->stream));
  return CPP_OK;
}

// ---------------------------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------------------------
struct cpp_batch {
  cpp_ctx* ctx; int maxB, B; long elems; int A; int dtype;
  void* s[2]; float *a, *r, *m;
  float* white;        // [2 states][2][CPP_MAX_CHANNELS]-compatible: laid out [2][2*C] for the current C
  double* part;        // [2][maxB][2*CPP_MAX_CHANNELS]
  int stats_C;         // channels the statistics were computed for (0: none yet)
  // device-sampled minibatch that was NOT gathered: state k of row b is row slot[k][b] of direct_store (the replay store);
  // only the f16-pipe conv1 kernels can consume it (direct_store == nullptr: s[] holds the gathered copy)
  int32_t* slot[2]; const void* direct_store;
  int32_t* slot_alt[2];   // the set the NEXT minibatch's sample pass writes while conv1's dW still reads slot[] (step_body)
  Arena arena;
};

extern "C" int cpp_batch_create(cpp_ctx* ctx, int max_batch, int64_t state_elems, int action_dim, cpp_batch** out) {
  ARG_CHECK(ctx && out, "cpp_batch_create: NULL argument");
  ARG_CHECK(max_batch >= 1 && state_elems >= 1 && action_dim >= 1, "cpp_batch_create: bad sizes");
  HIP_CHECK(hipSetDevice(ctx->device));
  cpp_batch* b = new cpp_batch();
  b->arena.stream = ctx->stream;
  b->ctx = ctx; b->maxB = max_batch; b->B = 0; b->elems = state_elems; b->A = action_dim; b->dtype = CPP_F16; b->stats_C = 0;
  int rc = 0;
  for (int k = 0; k < 2 && !rc; ++k) rc = b->arena.alloc(&b->s[k], (size_t)max_batch * state_elems * sizeof(float), false);
  if (!rc) rc = dalloc(b->arena, &b->a, (size_t)max_batch * action_dim);
  if (!rc) rc = dalloc(b->arena, &b->r, (size_t)max_batch);
  if (!rc) rc = dalloc(b->arena, &b->m, (size_t)max_batch);
  if (!rc) rc = dalloc(b->arena, &b->white, (size_t)4 * CPP_MAX_CHANNELS);
  if (!rc) rc = dalloc(b->arena, &b->part, (size_t)2 * max_batch * 2 * CPP_MAX_CHANNELS);
  b->direct_store = nullptr;
  for (int k = 0; k < 2 && !rc; ++k) rc = dalloc(b->arena, &b->slot[k], (size_t)max_batch);
  for (int k = 0; k < 2 && !rc; ++k) rc = dalloc(b->arena, &b->slot_alt[k], (size_t)max_batch);
  if (rc) { b->arena.release(); delete b; return rc; }
  *out = b;
  return CPP_OK;
}
extern "C" int cpp_batch_destroy(cpp_batch* b) {
  if (!b) return CPP_OK;
  (void)hipSetDevice(b->ctx->device);
  (void)hipStreamSynchronize(b->ctx->stream);
  b->arena.release(); delete b; return CPP_OK;
}
extern "C" int cpp_batch_size(const cpp_batch* b) { return b ? b->B : -1; }
extern "C" int cpp_batch_state_dtype(const cpp_batch* b) { return b ? b->dtype : -1; }

extern "C" int cpp_batch_upload(cpp_batch* b, int B, const void* s1, const void* s2, int dtype,
                                const float* action, const float* reward, const float* mask) {
  if (b) b->direct_store = nullptr;
  ARG_CHECK(b && s1, "cpp_batch_upload: NULL argument");
  ARG_CHECK(B >= 1 && B <= b->maxB, "cpp_batch_upload: batch %d outside [1,%d]", B, b->maxB);
  ARG_CHECK(dtype == CPP_F32 || dtype == CPP_F16, "cpp_batch_upload: dtype %d", dtype);
  hipStream_t st = b->ctx->stream;
  HIP_CHECK(hipSetDevice(b->ctx->device));
  const size_t sb = (size_t)B * b->elems * (dtype == CPP_F16 ? 2 : 4);
  HIP_CHECK(hipMemcpyAsync(b->s[0], s1, sb, hipMemcpyHostToDevice, st));
  if (s2) HIP_CHECK(hipMemcpyAsync(b->s[1], s2, sb, hipMemcpyHostToDevice, st));
  if (action) HIP_CHECK(hipMemcpyAsync(b->a, action, (size_t)B * b->A * sizeof(float), hipMemcpyHostToDevice, st));
  if (reward) HIP_CHECK(hipMemcpyAsync(b->r, reward, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
  if (mask) HIP_CHECK(hipMemcpyAsync(b->m, mask, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));
  b->B = B; b->dtype = dtype; b->stats_C = 0;
  return CPP_OK;
}

extern "C" int cpp_batch_download(cpp_batch* b, void* s1, void* s2, float* action, float* reward, float* mask) {
  ARG_CHECK(b, "cpp_batch_download: NULL argument");
  ARG_CHECK(b->B >= 1, "cpp_batch_download: batch is empty");
  hipStream_t st = b->ctx->stream;
  const size_t sb = (size_t)b->B * b->elems * (b->dtype == CPP_F16 ? 2 : 4);
  if (s1) HIP_CHECK(hipMemcpyAsync(s1, b->s[0], sb, hipMemcpyDeviceToHost, st));
  if (s2) HIP_CHECK(hipMemcpyAsync(s2, b->s[1], sb, hipMemcpyDeviceToHost, st));
  if (action) HIP_CHECK(hipMemcpyAsync(action, b->a, (size_t)b->B * b->A * sizeof(float), hipMemcpyDeviceToHost, st));
  if (reward) HIP_CHECK(hipMemcpyAsync(reward, b->r, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost, st));
  if (mask) HIP_CHECK(hipMemcpyAsync(mask, b->m, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

static int batch_ensure_stats(cpp_batch* b, int C) {
  if (C <= 0 || b->stats_C == C) return CPP_OK;
  RC(batch_stats(b->ctx, b->s[0], b->s[1], b->dtype, b->elems, b->B, C, b->part, b->white));
  b->stats_C = C;
  return CPP_OK;
}

// ---------------------------------------------------------------------------------------------
// replay
// ---------------------------------------------------------------------------------------------
struct cpp_replay {
  cpp_ctx* ctx; int rows, slots, A, size; long elems;
  int store_dtype;         // CPP_F16 (replay_memory.py:32) or CPP_U8 (pixel codes k, read back as f16(k/255): half the HBM)
  void* store; int32_t *s1, *s2, *rows_in, *rows_out; float *action, *reward, *mask;
  uint64_t* counter;       // device-side Philox counter for graph replay
  __half* lut; int* bad; uint16_t lut_host[256];      // CPP_U8: f16(k/255) table, "not a pixel image" flag
  void* stage; size_t stage_cap;                      // device staging of incoming states (conversion source)
  void* pinned; size_t pinned_cap; hipEvent_t pinned_free; bool pinned_busy;   // host staging: writes return before the copy ends
  Arena arena;
};
static size_t replay_esz(const cpp_replay* r) { return r->store_dtype == CPP_U8 ? 1 : sizeof(__half); }

// f16(k / 255.0) rounded to nearest-even from the exact quotient -- numpy's float16(k / 255.0), which is what the
// reference's renders hold (bullet_cartpole.py:239-243)
static uint16_t f16_of_code(int k) {
  const double d = (double)k / 255.0;
  const uint16_t h0 = __half_as_ushort(__float2half((float)d));
  uint16_t best = h0; double berr = 1e9;
  for (int delta = -1; delta <= 1; ++delta) {
    const int hb = (int)h0 + delta;
    if (hb < 0 || hb > 0x7bff) continue;
    const double err = fabs((double)__half2float(__ushort_as_half((uint16_t)hb)) - d);
    if (err < berr || (err == berr && (hb & 1) == 0)) { berr = err; best = (uint16_t)hb; }
  }
  return best;
}

static int replay_create(cpp_ctx* ctx, int buffer_size, int state_slots, int64_t state_elems, int action_dim, int store_dtype,
                         cpp_replay** out) {
  ARG_CHECK(ctx && out, "cpp_replay_create: NULL argument");
  ARG_CHECK(buffer_size >= 1 && state_elems >= 1 && action_dim >= 1, "cpp_replay_create: bad sizes");
  ARG_CHECK(state_slots >= buffer_size + 1, "cpp_replay_create: %d state slots for %d rows", state_slots, buffer_size);
  ARG_CHECK(store_dtype == CPP_F16 || store_dtype == CPP_U8, "cpp_replay_create: store dtype %d", store_dtype);
  ARG_CHECK(store_dtype != CPP_U8 || state_elems % 8 == 0, "cpp_replay_create: the 8-bit store needs state_elems %% 8 == 0");
  HIP_CHECK(hipSetDevice(ctx->device));
  cpp_replay* r = new cpp_replay();
  r->arena.stream = ctx->stream;
  r->ctx = ctx; r->rows = buffer_size; r->slots = state_slots; r->A = action_dim; r->size = 0; r->elems = state_elems;
  r->store_dtype = store_dtype;
  r->stage = nullptr; r->stage_cap = 0; r->pinned = nullptr; r->pinned_cap = 0; r->pinned_busy = false; r->lut = nullptr; r->bad = nullptr;
  HIP_CHECK(hipEventCreateWithFlags(&r->pinned_free, hipEventDisableTiming));
  int rc = r->arena.alloc(&r->store, (size_t)state_slots * state_elems * replay_esz(r), false);
  if (!rc) rc = dalloc(r->arena, &r->s1, (size_t)buffer_size);
  if (!rc) rc = dalloc(r->arena, &r->s2, (size_t)buffer_size);
  if (!rc) rc = dalloc(r->arena, &r->rows_in, (size_t)65536);
  if (!rc) rc = dalloc(r->arena, &r->rows_out, (size_t)65536);
  if (!rc) rc = dalloc(r->arena, &r->action, (size_t)buffer_size * action_dim);
  if (!rc) rc = dalloc(r->arena, &r->reward, (size_t)buffer_size);
  if (!rc) rc = dalloc(r->arena, &r->mask, (size_t)buffer_size);
  if (!rc) rc = dalloc(r->arena, &r->counter, (size_t)1);
  if (!rc) {
    rc = r->arena.alloc((void**)&r->lut, 256 * sizeof(__half), false);
    if (!rc) rc = r->arena.alloc((void**)&r->bad, sizeof(int), true);
    if (!rc) {
      for (int k = 0; k < 256; ++k) r->lut_host[k] = f16_of_code(k);
      if (hipMemcpy(r->lut, r->lut_host, sizeof(r->lut_host), hipMemcpyHostToDevice) != hipSuccess) rc = CPP_ERR_HIP;
    }
  }
  if (rc) { r->arena.release(); delete r; return rc; }
  *out = r;
  return CPP_OK;
}
extern "C" int cpp_replay_create(cpp_ctx* ctx, int buffer_size, int state_slots, int64_t state_elems,
                                 int action_dim, cpp_replay** out) {
  return replay_create(ctx, buffer_size, state_slots, state_elems, action_dim, CPP_F16, out);
}
extern "C" int cpp_replay_create_ex(cpp_ctx* ctx, int buffer_size, int state_slots, int64_t state_elems,
                                    int action_dim, int store_dtype, cpp_replay** out) {
  return replay_create(ctx, buffer_size, state_slots, state_elems, action_dim, store_dtype, out);
}
extern "C" int cpp_replay_destroy(cpp_replay* r) {
  if (!r) return CPP_OK;
  (void)hipSetDevice(r->ctx->device);
  (void)hipStreamSynchronize(r->ctx->stream);
  if (r->stage) (void)hipFree(r->stage);
  if (r->pinned) (void)hipHostFree(r->pinned);
  (void)hipEventDestroy(r->pinned_free);
  r->arena.release(); delete r; return CPP_OK;
}

// self.state[idx] = s (replay_memory.py:67,106).  The host rows go through a pinned staging buffer, so the call returns
// as soon as they are copied there: the H2D transfer and the conversions run on the context's stream, in order with
// everything launched later (rendered frames straight into replay slots, no stall of the rollout loop).
extern "C" int cpp_replay_write_states(cpp_replay* r, const int32_t* slots, int n, const void* states, int dtype) {
  ARG_CHECK(r && slots && states, "cpp_replay_write_states: NULL argument");
  ARG_CHECK(dtype == CPP_F32 || dtype == CPP_F16 || dtype == CPP_U8, "cpp_replay_write_states: dtype %d", dtype);
  hipStream_t st = r->ctx->stream;
  HIP_CHECK(hipSetDevice(r->ctx->device));
  for (int i = 0; i < n; ++i)
    ARG_CHECK(slots[i] >= 0 && slots[i] < r->slots, "cpp_replay_write_states: slot %d outside [0,%d)", slots[i], r->slots);
  const size_t esz = dtype == CPP_U8 ? 1 : dtype == CPP_F16 ? sizeof(__half) : sizeof(float);
  const size_t need = (size_t)n * r->elems * esz;
  if (r->pinned_busy) { HIP_CHECK(hipEventSynchronize(r->pinned_free)); r->pinned_busy = false; }   // previous transfer done
  if (need > r->pinned_cap) {
    if (r->pinned) HIP_CHECK(hipHostFree(r->pinned));
    HIP_CHECK(hipHostMalloc(&r->pinned, need, hipHostMallocDefault)); r->pinned_cap = need;
  }
  memcpy(r->pinned, states, need);
  const bool direct = dtype == r->store_dtype;       // no conversion (f16 -> f16, camera bytes -> 8-bit store): copy into the slots
  bool checked = false;
  if (direct) {
    for (int i = 0; i < n; ++i)
      HIP_CHECK(hipMemcpyAsync((char*)r->store + (size_t)slots[i] * r->elems * esz, (const char*)r->pinned + (size_t)i * r->elems * esz,
                               (size_t)r->elems * esz, hipMemcpyHostToDevice, st));
  } else {
    if (need > r->stage_cap) {
      HIP_CHECK(hipStreamSynchronize(st));
      if (r->stage) HIP_CHECK(hipFree(r->stage));
      HIP_CHECK(hipMalloc(&r->stage, need)); r->stage_cap = need;
    }
    HIP_CHECK(hipMemcpyAsync(r->stage, r->pinned, need, hipMemcpyHostToDevice, st));
    for (int i = 0; i < n; ++i) {
      const char* src = (const char*)r->stage + (size_t)i * r->elems * esz;
      if (r->store_dtype == CPP_U8) {
        RC(launch_to_u8(r->ctx, (uint8_t*)r->store + (size_t)slots[i] * r->elems, src, dtype, r->elems, r->lut, r->bad));
        checked = true;
      } else if (dtype == CPP_U8)
        RC(launch_u8_to_f16(r->ctx, (__half*)r->store + (size_t)slots[i] * r->elems, (const uint8_t*)src, r->elems, r->lut));
      else
        RC(launch_f32_to_f16(r->ctx, (__half*)r->store + (size_t)slots[i] * r->elems, (const float*)src, r->elems));
    }
  }
  HIP_CHECK(hipEventRecord(r->pinned_free, st));
  r->pinned_busy = true;
  if (checked) {      // the exactness check is part of the contract: report it with this call
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, r->bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (bad) {
      HIP_CHECK(hipMemsetAsync(r->bad, 0, sizeof(int), st));
      cpp_set_error("cpp_replay_write_states: the 8-bit store holds pixel images only (every value must be f16(k/255)); "
                    "create the memory with the f16 store for other states");
      return CPP_ERR_ARG;
    }
  }
  return CPP_OK;
}

extern "C" int cpp_replay_write_rows(cpp_replay* r, const int32_t* rows, int n, const int32_t* s1, const int32_t* s2,
                                     const float* action, const float* reward, const float* mask) {
  ARG_CHECK(r && rows && s1 && s2 && action && reward && mask, "cpp_replay_write_rows: NULL argument");
  hipStream_t st = r->ctx->stream;
  HIP_CHECK(hipSetDevice(r->ctx->device));
  int i = 0;
  while (i < n) {          // contiguous runs of rows go out as one copy per column
    ARG_CHECK(rows[i] >= 0 && rows[i] < r->rows, "cpp_replay_write_rows: row %d outside [0,%d)", rows[i], r->rows);
    int j = i + 1;
    while (j < n && rows[j] == rows[j - 1] + 1) ++j;
    const int cnt = j - i, r0 = rows[i];
    HIP_CHECK(hipMemcpyAsync(r->s1 + r0, s1 + i, cnt * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(r->s2 + r0, s2 + i, cnt * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(r->action + (size_t)r0 * r->A, action + (size_t)i * r->A, (size_t)cnt * r->A * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(r->reward + r0, reward + i, cnt * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(r->mask + r0, mask + i, cnt * sizeof(float), hipMemcpyHostToDevice, st));
    i = j;
  }
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

extern "C" int cpp_replay_set_size(cpp_replay* r, int size) {
  ARG_CHECK(r && size >= 0 && size <= r->rows, "cpp_replay_set_size: size %d", size);
  r->size = size;
  return CPP_OK;
}

extern "C" int cpp_replay_read_states(cpp_replay* r, const int32_t* slots, int n, void* out_f16) {
  ARG_CHECK(r && slots && out_f16, "cpp_replay_read_states: NULL argument");
  hipStream_t st = r->ctx->stream;
  std::vector<uint8_t> codes(r->store_dtype == CPP_U8 ? (size_t)n * r->elems : 0);
  for (int i = 0; i < n; ++i) {
    ARG_CHECK(slots[i] >= 0 && slots[i] < r->slots, "cpp_replay_read_states: slot %d", slots[i]);
    if (r->store_dtype == CPP_U8)
      HIP_CHECK(hipMemcpyAsync(codes.data() + (size_t)i * r->elems, (const uint8_t*)r->store + (size_t)slots[i] * r->elems,
                               (size_t)r->elems, hipMemcpyDeviceToHost, st));
    else
      HIP_CHECK(hipMemcpyAsync((__half*)out_f16 + (size_t)i * r->elems, (const __half*)r->store + (size_t)slots[i] * r->elems,
                               (size_t)r->elems * sizeof(__half), hipMemcpyDeviceToHost, st));
  }
  HIP_CHECK(hipStreamSynchronize(st));
  if (r->store_dtype == CPP_U8) {
    uint16_t* o = (uint16_t*)out_f16;
    for (size_t i = 0; i < codes.size(); ++i) o[i] = r->lut_host[codes[i]];
  }
  return CPP_OK;
}

// device-only part of sampling (graph-capturable when rows_dev == nullptr or already resident)
// descriptor of the fused sample + gather + statistics pass into `out` (C_out: channels of the vector statistics path, 0: none)
static GatherArgs replay_gather_args(cpp_replay* r, int B, const int32_t* rows_dev, uint64_t seed, const uint64_t* counter_dev,
                                     int channels, cpp_batch* out, bool direct, int* C_out) {
  int C = channels;
  if (C > 0) {
    int g = 8, c = C; while (c) { int t = g % c; g = c; c = t; }
    if (r->elems % 8 != 0 || C / g > 16 || r->elems % C != 0) C = 0;    // statistics via the generic path below
  }
  GatherArgs a; memset(&a, 0, sizeof(a));
  a.store[0] = r->store; a.store[1] = r->store; a.s_idx[0] = r->s1; a.s_idx[1] = r->s2; a.lut = r->lut;
  a.rows = rows_dev; a.rows_out = r->rows_out;
  a.action = r->action; a.reward = r->reward; a.mask = r->mask;
  // direct: no gathered copy -- statistics + the store rows of the sampled states; conv1 reads the store (caller checked)
  a.out_state[0] = direct ? nullptr : out->s[0]; a.out_state[1] = direct ? nullptr : out->s[1];
  a.out_slot[0] = direct ? out->slot[0] : nullptr; a.out_slot[1] = direct ? out->slot[1] : nullptr;
  out->direct_store = direct ? r->store : nullptr;
  a.out_action = out->a; a.out_reward = out->r; a.out_mask = out->m;
  a.part = out->part; a.seed = seed; a.counter = counter_dev;
  a.elems = r->elems; a.B = B; a.size = r->size; a.action_dim = r->A; a.C = C;
  *C_out = C;
  return a;
}
// what follows the gather kernel: the batch's bookkeeping and the whitening tables
static int replay_sample_finish(cpp_replay* r, int B, int C, int channels, cpp_batch* out) {
  out->B = B; out->dtype = CPP_F16; out->stats_C = 0;
  if (C > 0) {
    RC(launch_stats_finalize(r->ctx, out->part, B, 2, C, (double)B * (double)(r->elems / C), out->white));
    out->stats_C = C;
  } else if (channels > 0) {
    RC(batch_ensure_stats(out, channels));
  }
  return CPP_OK;
}
static int replay_sample_device(cpp_replay* r, int B, const int32_t* rows_dev, uint64_t seed, const uint64_t* counter_dev,
                                int channels, cpp_batch* out, bool direct = false) {
  int C = 0;
  const GatherArgs a = replay_gather_args(r, B, rows_dev, seed, counter_dev, channels, out, direct, &C);
  RC(launch_gather_stats(r->ctx, a, r->store_dtype));      // a CPP_U8 store gathers to f16 as well
  return replay_sample_finish(r, B, C, channels, out);
}

extern "C" int cpp_replay_sample(cpp_replay* r, int B, const int32_t* idxs, uint64_t seed, uint64_t counter,
                                 int channels, cpp_batch* out) {
  ARG_CHECK(r && out, "cpp_replay_sample: NULL argument");
  ARG_CHECK(B >= 1 && B <= out->maxB && B <= 65536, "cpp_replay_sample: batch %d outside [1,%d]", B, out->maxB);
  ARG_CHECK(out->elems == r->elems && out->A == r->A, "cpp_replay_sample: batch/replay shapes differ");
  ARG_CHECK(channels >= 0 && channels <= CPP_MAX_CHANNELS, "cpp_replay_sample: channels %d", channels);
  if (r->size <= 0) { cpp_set_error("cpp_replay_sample: replay memory is empty"); return CPP_ERR_STATE; }
  hipStream_t st = r->ctx->stream;
  HIP_CHECK(hipSetDevice(r->ctx->device));
  const int32_t* rows_dev = nullptr;
  if (idxs) {
    for (int i = 0; i < B; ++i)
      ARG_CHECK(idxs[i] >= 0 && idxs[i] < r->size, "cpp_replay_sample: index %d outside [0,%d)", idxs[i], r->size);
    HIP_CHECK(hipMemcpyAsync(r->rows_in, idxs, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, st));
    rows_dev = r->rows_in;
  } else {
    HIP_CHECK(hipMemcpyAsync(r->counter, &counter, sizeof(uint64_t), hipMemcpyHostToDevice, st));
  }
  RC(replay_sample_device(r, B, rows_dev, seed, idxs ? nullptr : r->counter, channels, out));
  if (idxs) HIP_CHECK(hipStreamSynchronize(st));     // the caller's index array may go away after return
  return CPP_OK;
}

extern "C" int cpp_replay_last_indexes(cpp_replay* r, int B, int32_t* out) {
  ARG_CHECK(r && out && B >= 1 && B <= 65536, "cpp_replay_last_indexes: bad argument");
  HIP_CHECK(hipMemcpyAsync(out, r->rows_out, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, r->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(r->ctx->stream));
  return CPP_OK;
}

extern "C" int cpp_replay_fill_synthetic(cpp_replay* r, int n_rows, uint64_t seed) {
  ARG_CHECK(r && n_rows >= 1 && n_rows <= r->rows, "cpp_replay_fill_synthetic: rows %d", n_rows);
  ARG_CHECK(n_rows + n_rows / 50 + 1 <= r->slots, "cpp_replay_fill_synthetic: not enough state slots");
  HIP_CHECK(hipSetDevice(r->ctx->device));
  RC(launch_replay_fill(r->ctx, r->store_dtype == CPP_U8 ? nullptr : (__half*)r->store, r->elems, r->slots, r->s1, r->s2,
                        r->action, r->reward, r->mask, n_rows, r->A, seed));
  if (r->store_dtype == CPP_U8) RC(launch_replay_fill_u8(r->ctx, (uint8_t*)r->store, r->elems * (long)r->slots, seed));
  HIP_CHECK(hipStreamSynchronize(r->ctx->stream));
  r->size = n_rows;
  return CPP_OK;
}

// ---------------------------------------------------------------------------------------------
// DDPG
// ---------------------------------------------------------------------------------------------
constexpr int NORM_PARTS = 64;

struct cpp_ddpg {
  cpp_ctx* ctx; cpp_net *actor, *critic, *tactor, *tcritic; cpp_ddpg_hyper hp;
  int maxB; long nA, nC;
  float* gradbuf; float *dq_da, *td, *dq, *loss_norms /* [0] loss [1] actor norm [2] critic norm */, *ones;
  double* norm_part;
  double* heads_part;                              // fused heads kernel: per-workgroup partial sums of td^2
  int heads_grid, heads_B;                         // ... of the last graph built by compute_gradients (0: GEMM levels + td_kernel)
  int loss_parts, loss_B;                          // how cpp_ddpg_last_stats finds the loss of the last call: partials to add, or loss_norms[0]
  // graph replay of the full inner step
  hipGraph_t graph; hipGraphExec_t gexec; bool graph_ok; int g_B, g_nb, g_size; uint64_t g_seed; cpp_replay* g_replay;   // g_size: rows in the replay when captured (the sampler's range is a kernel argument)
  cpp_batch* step_batch;
  // graph replay of the data-parallel half step (sample + both gradient sets)
  // three variants: 0 samples its own minibatch; 1 / 2 find it presampled (by the previous call's rider, conv1_dw_gather.hip)
  // in the second / first set of slot arrays.  One key for all three.
  hipGraph_t hgraph[3]; hipGraphExec_t hexec[3]; bool hgraph_ok[3]; int h_B, h_size; uint64_t h_seed; cpp_replay* h_replay;
  int h_next[3];           // variant the call after variant v must use (0: the rider did not leave)
  int pre_variant;         // variant of the next cpp_ddpg_sample_and_compute call if its key still matches (0: sample)
  int32_t* slot_set[2][2]; // the two sets of slot arrays of step_batch
  Arena arena;
};

extern "C" int cpp_ddpg_create(cpp_ctx* ctx, cpp_net* actor, cpp_net* critic, cpp_net* tactor, cpp_net* tcritic,
                               const cpp_ddpg_hyper* hp, cpp_ddpg** out) {
  ARG_CHECK(ctx && actor && critic && tactor && tcritic && hp && out, "cpp_ddpg_create: NULL argument");
  ARG_CHECK(actor->spec.kind == CPP_ACTOR && tactor->spec.kind == CPP_ACTOR, "cpp_ddpg_create: actor kinds");
  ARG_CHECK(critic->spec.kind == CPP_CRITIC && tcritic->spec.kind == CPP_CRITIC, "cpp_ddpg_create: critic kinds");
  ARG_CHECK(actor->nparams == tactor->nparams && critic->nparams == tcritic->nparams, "cpp_ddpg_create: target shapes differ");
  ARG_CHECK(actor->state_elems == critic->state_elems && actor->spec.action_dim == critic->spec.action_dim,
            "cpp_ddpg_create: actor/critic input shapes differ");
  HIP_CHECK(hipSetDevice(ctx->device));
  cpp_ddpg* d = new cpp_ddpg();
  d->arena.stream = ctx->stream;
  d->ctx = ctx; d->actor = actor; d->critic = critic; d->tactor = tactor; d->tcritic = tcritic; d->hp = *hp;
  d->maxB = actor->maxB < critic->maxB ? actor->maxB : critic->maxB;
  d->nA = actor->nparams; d->nC = critic->nparams;
  d->graph = nullptr; d->gexec = nullptr; d->graph_ok = false; d->step_batch = nullptr; d->g_replay = nullptr;
  for (int v = 0; v < 3; ++v) { d->hgraph[v] = nullptr; d->hexec[v] = nullptr; d->hgraph_ok[v] = false; d->h_next[v] = 0; }
  d->h_replay = nullptr; d->pre_variant = 0; d->h_B = 0; d->h_size = 0; d->h_seed = 0;
  memset(d->slot_set, 0, sizeof(d->slot_set));
  d->heads_grid = d->heads_B = d->loss_parts = d->loss_B = 0;
  const int A = actor->spec.action_dim;
  int rc = dalloc(d->arena, &d->gradbuf, (size_t)(d->nA + d->nC));
  if (!rc) rc = dalloc(d->arena, &d->dq_da, (size_t)d->maxB * A);
  if (!rc) rc = dalloc(d->arena, &d->td, (size_t)d->maxB);
  if (!rc) rc = dalloc(d->arena, &d->dq, (size_t)d->maxB);
  if (!rc) rc = dalloc(d->arena, &d->ones, (size_t)d->maxB);
  if (!rc) rc = dalloc(d->arena, &d->loss_norms, (size_t)4);
  if (!rc) rc = dalloc(d->arena, &d->norm_part, (size_t)OPT_MAX_SEGS * NORM_PARTS);
  if (!rc) rc = dalloc(d->arena, &d->heads_part, (size_t)DDPG_HEADS_MAX_WGS);
  if (!rc) rc = launch_fill(ctx, d->ones, 1, 0, 1, d->maxB, 1.0f);
  if (rc) { d->arena.release(); delete d; return rc; }
  actor->grads = d->gradbuf; critic->grads = d->gradbuf + d->nA;
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  *out = d;
  return CPP_OK;
}

extern "C" int cpp_ddpg_destroy(cpp_ddpg* d) {
  if (!d) return CPP_OK;
  (void)hipSetDevice(d->ctx->device);
  (void)hipStreamSynchronize(d->ctx->stream);
  if (d->gexec) (void)hipGraphExecDestroy(d->gexec);
  if (d->graph) (void)hipGraphDestroy(d->graph);
  for (int v = 0; v < 3; ++v) {
    if (d->hexec[v]) (void)hipGraphExecDestroy(d->hexec[v]);
    if (d->hgraph[v]) (void)hipGraphDestroy(d->hgraph[v]);
  }
  if (d->step_batch) cpp_batch_destroy(d->step_batch);
  d->actor->grads = nullptr; d->critic->grads = nullptr;
  d->arena.release(); delete d; return CPP_OK;
}

static int check_batch(cpp_ddpg* d, cpp_batch* b, const char* who) {
  ARG_CHECK(d && b, "%s: NULL argument", who);
  ARG_CHECK(b->B >= 1 && b->B <= d->maxB, "%s: batch size %d outside [1,%d]", who, b->B, d->maxB);
  ARG_CHECK(b->elems == d->actor->state_elems && b->A == d->actor->spec.action_dim, "%s: batch shape does not match the networks", who);
  return CPP_OK;
}

static const float* white_of(cpp_batch* b, int which, int C) { return b->white + (long)which * 2 * C; }

// critic "prefix": conv trunk + the fully connected layers in front of the action splice
static int critic_prefix(cpp_net* c, const void* state, int dtype, const float* white, int B) {
  RC(net_forward_trunk(c, c->ws[0], state, dtype, white, B));
  if (c->cat_layer > 0) {
    // run layers [0, cat) only
    for (int l = 0; l < c->cat_layer; ++l) {
      const FcL& L = c->fc[l];
      RC(gemm(c->ctx, c->ws[0].fcin[l], L.n_in + 1, 1, c->params + L.w_off, L.n_out, 1, c->ws[0].fcin[l + 1],
              c->fc[l + 1].n_in + 1, B, L.n_out, L.n_in + 1, L.act));
    }
  }
  return CPP_OK;
}

// evaluate the critic head from the splice on, in workspace `wi`, with the given device action batch
static int critic_head(cpp_net* c, int wi, const float* action, int B) {
  const int cl = c->cat_layer, A = c->spec.action_dim;
  if (wi == 1) {
    const FcL& L = c->fc[cl];
    RC(launch_copy_cols(c->ctx, c->ws[1].fcin[cl], L.n_in + 1, 0, c->ws[0].fcin[cl], L.n_in + 1, 0, L.n_in - A, B));
  }
  return net_forward_fc(c, c->ws[wi], cl, B, action);
}

// ddpg_cartpole.py:111-113 + :220-222.  critic_prefix_done: the critic prefix for batch.state_1 is
// already in critic->ws[0] (fused step computes it once for both updates).
static int actor_gradients(cpp_ddpg* d, cpp_batch* b, bool critic_prefix_done) {
  cpp_net *a = d->actor, *c = d->critic;
  const int B = b->B, C = a->spec.pixel ? a->spec.C : 0;
  const float* w1 = white_of(b, 0, C);
  RC(net_forward_trunk(a, a->ws[0], b->s[0], b->dtype, w1, B));
  RC(net_forward_fc(a, a->ws[0], 0, B, nullptr));
  if (!critic_prefix_done) RC(critic_prefix(c, b->s[0], b->dtype, w1, B));
  RC(critic_head(c, 1, a->ws[0].out, B));
  // d(sum_b Q)/da: dz of the linear q layer is 1
  const int last = (int)c->fc.size() - 1;
  RC(launch_copy_cols(d->ctx, c->ws[1].dz[last], 1, 0, d->ones, 1, 0, 1, B));
  // walk back to the splice (hidden layers after the splice are ReLU)
  for (int l = last; l > c->cat_layer; --l) {
    const FcL& L = c->fc[l];
    RC(gemm(d->ctx, c->ws[1].dz[l], L.n_out, 1, c->params + L.w_off, 1, L.n_out, c->ws[1].dz[l - 1], L.n_in, B, L.n_in,
            L.n_out, GE_MUL_RELU_GRAD, c->ws[1].fcin[l], L.n_in + 1));
  }
  {
    const FcL& L = c->fc[c->cat_layer];
    const int A = c->spec.action_dim;
    RC(gemm(d->ctx, c->ws[1].dz[c->cat_layer], L.n_out, 1, c->params + L.w_off + (long)(L.n_in - A) * L.n_out, 1, L.n_out,
            d->dq_da, A, B, A, L.n_out, GE_NONE));
  }
  // grad_ys = -dQ/da through the tanh head, then the whole actor backward
  const int alast = (int)a->fc.size() - 1;
  RC(launch_actor_head_grad(d->ctx, a->ws[0].dz[alast], d->dq_da, a->ws[0].out, B * a->spec.action_dim));
  RC(net_backward(a, a->ws[0], B, true, nullptr, b->s[0], b->dtype, w1));
  return CPP_OK;
}

// ddpg_cartpole.py:199-214
static int critic_gradients_impl(cpp_ddpg* d, cpp_batch* b, bool critic_prefix_done, bool backward);
// backward == false is check_loss (ddpg_cartpole.py:239-248), which feeds IS_TRAINING False; the train op (:237) feeds
// True for the whole graph, target networks included
static int critic_gradients(cpp_ddpg* d, cpp_batch* b, bool critic_prefix_done, bool backward) {
  cpp_net* nets[3] = {d->critic, d->tactor, d->tcritic};
  for (cpp_net* n : nets) n->is_training = backward;
  const int rc = critic_gradients_impl(d, b, critic_prefix_done, backward);
  for (cpp_net* n : nets) n->is_training = true;
  return rc;
}
static int critic_gradients_impl(cpp_ddpg* d, cpp_batch* b, bool critic_prefix_done, bool backward) {
  cpp_net *c = d->critic, *ta = d->tactor, *tc = d->tcritic;
  const int B = b->B, C = c->spec.pixel ? c->spec.C : 0;
  const float *w1 = white_of(b, 0, C), *w2 = white_of(b, 1, C);
  RC(net_forward_trunk(ta, ta->ws[0], b->s[1], b->dtype, w2, B));
  RC(net_forward_fc(ta, ta->ws[0], 0, B, nullptr));
  RC(critic_prefix(tc, b->s[1], b->dtype, w2, B));
  RC(critic_head(tc, 0, ta->ws[0].out, B));
  if (!critic_prefix_done) RC(critic_prefix(c, b->s[0], b->dtype, w1, B));
  RC(critic_head(c, 0, b->a, B));
  const int last = (int)c->fc.size() - 1;
  RC(launch_td(d->ctx, c->ws[0].out, tc->ws[0].out, b->r, b->m, d->hp.discount, B, d->td,
               backward ? c->ws[0].dz[last] : nullptr, d->loss_norms));
  d->loss_parts = 0;
  if (backward) RC(net_backward(c, c->ws[0], B, true, nullptr, b->s[0], b->dtype, w1));
  return CPP_OK;
}

static int apply(cpp_ddpg* d, bool do_actor, bool do_critic, float grad_scale, uint64_t* bump = nullptr) {
  OptSegs s; memset(&s, 0, sizeof(s));
  s.bump = bump;
  s.nseg = 2; s.kind = OPT_SGD;
  s.p[0] = d->actor->params; s.g[0] = d->gradbuf; s.n[0] = do_actor ? d->nA : 0; s.lr[0] = d->hp.actor_learning_rate; s.group[0] = 0;
  s.p[1] = d->critic->params; s.g[1] = d->gradbuf + d->nA; s.n[1] = do_critic ? d->nC : 0; s.lr[1] = d->hp.critic_learning_rate; s.group[1] = 1;
  RC(launch_sumsq(d->ctx, s, grad_scale, d->norm_part, NORM_PARTS));
  // norms_out[group] is only written for lists that were applied (n > 0)
  RC(launch_opt_apply(d->ctx, s, grad_scale, d->hp.gradient_clip, d->norm_part, NORM_PARTS, d->loss_norms + 1));
  return CPP_OK;
}

static int prep_batch(cpp_ddpg* d, cpp_batch* b) {
  if (d->actor->spec.pixel) RC(batch_ensure_stats(b, d->actor->spec.C));
  return CPP_OK;
}

extern "C" int cpp_ddpg_train_actor(cpp_ddpg* d, cpp_batch* b) {
  RC(check_batch(d, b, "cpp_ddpg_train_actor"));
  HIP_CHECK(hipSetDevice(d->ctx->device));
  RC(prep_batch(d, b));
  RC(actor_gradients(d, b, false));
  RC(apply(d, true, false, 1.0f));
  return CPP_OK;
}

extern "C" int cpp_ddpg_train_critic(cpp_ddpg* d, cpp_batch* b) {
  RC(check_batch(d, b, "cpp_ddpg_train_critic"));
  HIP_CHECK(hipSetDevice(d->ctx->device));
  RC(prep_batch(d, b));
  RC(critic_gradients(d, b, false, true));
  RC(apply(d, false, true, 1.0f));
  return CPP_OK;
}

extern "C" int cpp_ddpg_check_loss(cpp_ddpg* d, cpp_batch* b, float* loss, float* td, float* q) {
  RC(check_batch(d, b, "cpp_ddpg_check_loss"));
  HIP_CHECK(hipSetDevice(d->ctx->device));
  RC(prep_batch(d, b));
  RC(critic_gradients(d, b, false, false));
  hipStream_t st = d->ctx->stream;
  if (loss) HIP_CHECK(hipMemcpyAsync(loss, d->loss_norms, sizeof(float), hipMemcpyDeviceToHost, st));
  if (td) HIP_CHECK(hipMemcpyAsync(td, d->td, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost, st));
  if (q) HIP_CHECK(hipMemcpyAsync(q, d->critic->ws[0].out, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

extern "C" int cpp_ddpg_q_gradients_wrt_actions(cpp_ddpg* d, cpp_batch* b, float* dq_da, float* actions, float* q) {
  RC(check_batch(d, b, "cpp_ddpg_q_gradients_wrt_actions"));
  HIP_CHECK(hipSetDevice(d->ctx->device));
  RC(prep_batch(d, b));
  RC(actor_gradients(d, b, false));
  hipStream_t st = d->ctx->stream;
  const int A = d->actor->spec.action_dim;
  if (dq_da) HIP_CHECK(hipMemcpyAsync(dq_da, d->dq_da, (size_t)b->B * A * sizeof(float), hipMemcpyDeviceToHost, st));
  if (actions) HIP_CHECK(hipMemcpyAsync(actions, d->actor->ws[0].out, (size_t)b->B * A * sizeof(float), hipMemcpyDeviceToHost, st));
  if (q) HIP_CHECK(hipMemcpyAsync(q, d->critic->ws[1].out, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

// Both gradient sets of one minibatch (ddpg_cartpole.py:331-334) as one dependency graph: 4 conv trunk
// forwards, the MLP GEMMs batched level by level, 2 conv trunk backwards.  The critic trunk + the layers in
// front of the action splice run once for both uses of critic(s1, .).
static int compute_gradients(cpp_ddpg* d, cpp_batch* b) {
  cpp_ctx* ctx = d->ctx;
  cpp_net *a = d->actor, *c = d->critic, *ta = d->tactor, *tc = d->tcritic;
  const int B = b->B, A = a->spec.action_dim, C = a->spec.pixel ? a->spec.C : 0;
  const float *w1 = white_of(b, 0, C), *w2 = white_of(b, 1, C);
  const void *s1 = b->direct_store ? b->direct_store : b->s[0], *s2 = b->direct_store ? b->direct_store : b->s[1];
  struct SlotScope {      // conv1 of the four networks addresses its images through the sampled slots while this graph runs
    cpp_net* n[4];
    SlotScope(cpp_net* a_, cpp_net* c_, cpp_net* ta_, cpp_net* tc_, cpp_batch* b_) : n{a_, c_, ta_, tc_} {
      if (b_->direct_store) { a_->img_slot = c_->img_slot = b_->slot[0]; ta_->img_slot = tc_->img_slot = b_->slot[1]; }
    }
    ~SlotScope() { for (cpp_net* x : n) x->img_slot = nullptr; }
  } slot_scope(a, c, ta, tc, b);
  const int dt = b->dtype;
  const int na = (int)a->fc.size(), nc = (int)c->fc.size(), cat = c->cat_layer;
  const FcL& Lcat = c->fc[cat];
  const long ldcat = Lcat.n_in + 1;
  OpGraph G;

  // ---- forward: the four conv trunks.  conv1 saturates the chip per network; the narrow conv2 / conv3 layers
  // of all four networks share one launch each.
  int tA, tC, tTA, tTC;
  if (a->spec.pixel && !a->spec.use_batch_norm) {
    cpp_net* nets[4] = {a, c, ta, tc};
    const void* sts[4] = {s1, s1, s2, s2};
    const float* whs[4] = {w1, w1, w2, w2};
    const int t1 = G.fn([=] {
      for (int k = 0; k < 4; ++k) nets[k]->use_b16 = trunk_b16(nets[k], dt, B, 0);
      {
        ConvArgs cl[4]; int mode = 0;
        for (int k = 0; k < 4; ++k) cl[k] = conv_fwd_args(nets[k], nets[k]->ws[0], 0, sts[k], dt, whs[k], B, &mode);
        // all four conv1 forwards in one launch as well: 16 tiles per persistent workgroup amortise the weight
        // preload and the tail (measured 0.560 -> 0.526 ms per step for the four networks)
        RC(launch_conv_fwd_multi(ctx, kFwdKid[0], a->conv[0].Cin, a->conv[0].ks, mode, EPI_RELU_POOL, cl, 4));
      }
      for (int i = 1; i < 3; ++i) {
        ConvArgs cl[4]; int mode = 0;
        for (int k = 0; k < 4; ++k) cl[k] = conv_fwd_args(nets[k], nets[k]->ws[0], i, sts[k], dt, whs[k], B, &mode);
        RC(launch_conv_fwd_multi(ctx, kFwdKid[i], a->conv[i].Cin, a->conv[i].ks, mode, EPI_RELU_POOL, cl, 4));
      }
      return (int)CPP_OK; }, {});
    tA = tC = tTA = tTC = t1;
  } else if (a->spec.pixel) {       // batch norm (training mode for the whole graph, ddpg_cartpole.py:145,237)
    cpp_net* nets[4] = {a, c, ta, tc};
    const void* sts[4] = {s1, s1, s2, s2};
    const float* whs[4] = {w1, w1, w2, w2};
    const int t1 = G.fn([=] { return nets_forward_trunk_bn(ctx, nets, 4, sts, whs, dt, B); }, {});
    tA = tC = tTA = tTC = t1;
  } else {
    tA = G.fn([=] { return net_forward_trunk(a, a->ws[0], s1, dt, w1, B); }, {});
    tC = G.fn([=] { return net_forward_trunk(c, c->ws[0], s1, dt, w1, B); }, {});
    tTA = G.fn([=] { return net_forward_trunk(ta, ta->ws[0], s2, dt, w2, B); }, {});
    tTC = G.fn([=] { return net_forward_trunk(tc, tc->ws[0], s2, dt, w2, B); }, {});
  }
  // ---- fused heads (heads.hip): when the critic is "[prefix, action] -> relu layer -> linear q" and the actor ends in a tanh
  // layer (the reference's networks, ddpg_cartpole.py:95-100, :166-171), everything from the actors' / critics' last hidden
  // activations to the first backward layer is one row-local kernel instead of five dependent GEMM levels + TD + copies.
  // CPP_FUSED_HEADS=0 keeps the GEMM levels.
  static const bool no_heads = getenv("CPP_FUSED_HEADS") != nullptr && atoi(getenv("CPP_FUSED_HEADS")) == 0;
  DdpgHeadsArgs hd; memset(&hd, 0, sizeof(hd));
  bool fused = !no_heads && na >= 2 && cat >= 1 && nc - cat == 2 && a->fc[na - 1].act == GE_TANH && Lcat.act == GE_RELU &&
               c->fc[nc - 1].n_out == 1 && c->fc[nc - 1].act == GE_NONE && a->fc[na - 1].n_out == A;
  if (fused) {
    const FcL& Lo = a->fc[na - 1];
    hd.B = B; hd.A = A; hd.discount = d->hp.discount;
    hd.h2a = a->ws[0].fcin[na - 1]; hd.h2ta = ta->ws[0].fcin[na - 1]; hd.ld_h2a = Lo.n_in + 1; hd.n2a = Lo.n_in;
    hd.Wo = a->params + Lo.w_off; hd.Wo_t = ta->params + Lo.w_off;
    hd.h2c = c->ws[0].fcin[cat]; hd.h2tc = tc->ws[0].fcin[cat]; hd.ld_h2c = (int)ldcat; hd.n2c = Lcat.n_in - A;
    hd.W3 = c->params + Lcat.w_off; hd.W3_t = tc->params + Lcat.w_off; hd.n3 = Lcat.n_out;
    hd.wq = c->params + c->fc[nc - 1].w_off; hd.wq_t = tc->params + c->fc[nc - 1].w_off;
    hd.act = b->a; hd.r = b->r; hd.mask = b->m;
    hd.a_out = a->ws[0].out; hd.dq_da = d->dq_da; hd.adz = a->ws[0].dz[na - 1]; hd.dz_h2a = a->ws[0].dz[na - 2];
    hd.relu_x2 = relu_grad_epi(a, na - 2) == GE_MUL_RELU_GRAD_X2;
    hd.cat_splice = c->ws[0].fcin[cat] + (Lcat.n_in - A);
    hd.h3_out = c->ws[0].fcin[nc - 1]; hd.ld_h3 = Lcat.n_out + 1;
    hd.q_out = c->ws[0].out; hd.tq_out = tc->ws[0].out; hd.td = d->td; hd.dzq = c->ws[0].dz[nc - 1];
    hd.dz3 = c->ws[0].dz[cat]; hd.dz2c = c->ws[0].dz[cat - 1];
    hd.loss_part = d->heads_part;
    fused = ddpg_heads_supported(hd);
    // the actors are one layer deeper than the critics' prefix (100-100-50 against 200-50): their last hidden layer joins the
    // heads kernel so that both stacks reach it, and leave it, in the same number of GEMM levels.  CPP_HEADS_PRE=0: GEMMs.
    static const bool no_pre = getenv("CPP_HEADS_PRE") != nullptr && atoi(getenv("CPP_HEADS_PRE")) == 0;
    if (fused && !no_pre && na >= 3 && !a->drop_counter && a->fc[na - 2].act == GE_RELU && a->fc[na - 3].act == GE_RELU) {
      DdpgHeadsArgs hp = hd;
      const FcL& L2 = a->fc[na - 2];
      hp.h1a = a->ws[0].fcin[na - 2]; hp.h1ta = ta->ws[0].fcin[na - 2]; hp.ld_h1a = L2.n_in + 1; hp.n1a = L2.n_in;
      hp.W2 = a->params + L2.w_off; hp.W2_t = ta->params + L2.w_off;
      hp.h2a_out = a->ws[0].fcin[na - 1]; hp.dz_h1a = a->ws[0].dz[na - 3];
      if (ddpg_heads_supported(hp)) hd = hp;
    }
  }
  const int pre = (fused && hd.n1a > 0) ? 1 : 0;
  d->heads_grid = fused ? (B + 3) / 4 : 0; d->heads_B = B;
  d->loss_parts = d->heads_grid; d->loss_B = B;
  int adz, cdz;
  if (fused) {
    int aF = tA, taF = tTA;
    for (int l = 0; l < na - 1 - pre; ++l) {
      aF = G.gemm(fc_fwd_args(a, a->ws[0], l, B), {aF});
      taF = G.gemm(fc_fwd_args(ta, ta->ws[0], l, B), {taF});
    }
    if (a->drop_counter) {     // --use-dropout: this forward is counted once its layers have read the counter
      G.fn([=] { return bump_dropout(a); }, {aF});
      G.fn([=] { return bump_dropout(ta); }, {taF});
    }
    int cP = tC, tcP = tTC;
    for (int l = 0; l < cat; ++l) {
      cP = G.gemm(fc_fwd_args(c, c->ws[0], l, B), {cP});
      tcP = G.gemm(fc_fwd_args(tc, tc->ws[0], l, B), {tcP});
    }
    const int hk = G.fn([=] { return launch_ddpg_heads(ctx, hd); }, {aF, taF, cP, tcP});
    // ---- actor backward below its head (the head's dX is part of the fused kernel)
    G.gemm(fc_dw_args(a, a->ws[0], na - 1, B, a->ws[0].dz[na - 1]), {hk});
    adz = hk;
    for (int l = na - 2; l >= 0; --l) {
      const FcL& L = a->fc[l];
      G.gemm(fc_dw_args(a, a->ws[0], l, B, a->ws[0].dz[l]), {adz});
      if (pre && l == na - 2) continue;       // dz[l - 1] came out of the heads kernel
      if (l > 0)
        adz = G.gemm(fc_dx_args(a, l, B, a->ws[0].dz[l], L.n_out, 0, L.n_in, a->ws[0].dz[l - 1], L.n_in, relu_grad_epi(a, l - 1),
                                a->ws[0].fcin[l], L.n_in + 1), {adz});
      else if (a->spec.pixel)
        adz = G.gemm(fc_dx_args(a, 0, B, a->ws[0].dz[0], L.n_out, 0, a->flat, a->ws[0].dpool[2], a->flat, GE_NONE, nullptr, 0), {adz});
    }
    // ---- critic backward below its concat layer
    G.gemm(fc_dw_args(c, c->ws[0], nc - 1, B, c->ws[0].dz[nc - 1]), {hk});
    G.gemm(fc_dw_args(c, c->ws[0], cat, B, c->ws[0].dz[cat]), {hk});
    cdz = hk;
    for (int l = cat - 1; l >= 0; --l) {
      const FcL& L = c->fc[l];
      G.gemm(fc_dw_args(c, c->ws[0], l, B, c->ws[0].dz[l]), {cdz});
      if (l > 0)
        cdz = G.gemm(fc_dx_args(c, l, B, c->ws[0].dz[l], L.n_out, 0, L.n_in, c->ws[0].dz[l - 1], L.n_in, GE_MUL_RELU_GRAD,
                                c->ws[0].fcin[l], L.n_in + 1), {cdz});
      else if (c->spec.pixel)
        cdz = G.gemm(fc_dx_args(c, 0, B, c->ws[0].dz[0], L.n_out, 0, c->flat, c->ws[0].dpool[2], c->flat, GE_NONE, nullptr, 0), {cdz});
    }
  } else {
  const int cb = G.fn([=] { return launch_copy_cols(ctx, c->ws[0].fcin[cat], ldcat, Lcat.n_in - A, b->a, A, 0, A, B); }, {});
  int aF = tA, taF = tTA;
  for (int l = 0; l < na; ++l) {
    GemmArgs g = fc_fwd_args(a, a->ws[0], l, B), t = fc_fwd_args(ta, ta->ws[0], l, B);
    if (l == na - 1) {      // actions land directly in the critics' splice columns as well
      g.C2 = c->ws[1].fcin[cat] + (Lcat.n_in - A); g.ldc2 = ldcat;
      t.C2 = tc->ws[0].fcin[cat] + (Lcat.n_in - A); t.ldc2 = ldcat;
    }
    aF = G.gemm(g, {aF}); taF = G.gemm(t, {taF});
  }
  if (a->drop_counter) {     // --use-dropout: this forward is counted once its layers have read the counter
    G.fn([=] { return bump_dropout(a); }, {aF});
    G.fn([=] { return bump_dropout(ta); }, {taF});
  }
  int cP = tC, tcP = tTC;
  for (int l = 0; l < cat; ++l) {
    GemmArgs g = fc_fwd_args(c, c->ws[0], l, B);
    if (l == cat - 1) { g.C2 = c->ws[1].fcin[cat]; g.ldc2 = ldcat; }   // same prefix for the second evaluation
    cP = G.gemm(g, {cP});
    tcP = G.gemm(fc_fwd_args(tc, tc->ws[0], l, B), {tcP});
  }
  if (cat == 0)     // low-dim critic: the "prefix" is the converted state itself
    cP = G.fn([=] { return launch_copy_cols(ctx, c->ws[1].fcin[0], ldcat, 0, c->ws[0].fcin[0], ldcat, 0, Lcat.n_in - A, B); }, {tC});
  int c1 = -1, c0 = -1, tcH = -1;
  for (int l = cat; l < nc; ++l) {
    c1 = G.gemm(fc_fwd_args(c, c->ws[1], l, B), {l == cat ? cP : c1, l == cat ? aF : -1});
    c0 = G.gemm(fc_fwd_args(c, c->ws[0], l, B), {l == cat ? cP : c0, l == cat ? cb : -1, l == cat ? tC : -1});
    tcH = G.gemm(fc_fwd_args(tc, tc->ws[0], l, B), {l == cat ? tcP : tcH, l == cat ? taF : -1});
  }

  // ---- dQ/da at a = actor(s1): back through q_value .. splice on the second evaluation (dz of q is 1)
  int g = c1;
  for (int l = nc - 1; l > cat; --l) {
    const FcL& L = c->fc[l];
    const float* dz = (l == nc - 1) ? d->ones : c->ws[1].dz[l];
    g = G.gemm(fc_dx_args(c, l, B, dz, L.n_out, 0, L.n_in, c->ws[1].dz[l - 1], L.n_in, GE_MUL_RELU_GRAD,
                          c->ws[1].fcin[l], L.n_in + 1), {g});
  }
  {   // dQ/da (kept for cpp_ddpg_q_gradients_wrt_actions) and, in the same epilogue, the actor's head gradient
    const float* dz = (cat == nc - 1) ? d->ones : c->ws[1].dz[cat];
    GemmArgs ga = fc_dx_args(c, cat, B, dz, Lcat.n_out, Lcat.n_in - A, A, d->dq_da, A, GE_ACTOR_HEAD, a->ws[0].out, A);
    ga.C2 = a->ws[0].dz[na - 1]; ga.ldc2 = A;
    adz = G.gemm(ga, {g, aF});
  }

  // ---- actor backward
  for (int l = na - 1; l >= 0; --l) {
    const FcL& L = a->fc[l];
    G.gemm(fc_dw_args(a, a->ws[0], l, B, a->ws[0].dz[l]), {adz});
    if (l > 0)
      adz = G.gemm(fc_dx_args(a, l, B, a->ws[0].dz[l], L.n_out, 0, L.n_in, a->ws[0].dz[l - 1], L.n_in, relu_grad_epi(a, l - 1),
                              a->ws[0].fcin[l], L.n_in + 1), {adz});
    else if (a->spec.pixel)
      adz = G.gemm(fc_dx_args(a, 0, B, a->ws[0].dz[0], L.n_out, 0, a->flat, a->ws[0].dpool[2], a->flat, GE_NONE, nullptr, 0), {adz});
  }

  // ---- TD target + critic backward on the first evaluation (fed actions)
  cdz = G.fn([=] { return launch_td(ctx, c->ws[0].out, tc->ws[0].out, b->r, b->m, d->hp.discount, B, d->td,
                                        c->ws[0].dz[nc - 1], d->loss_norms); }, {c0, tcH});
  for (int l = nc - 1; l >= 0; --l) {
    const FcL& L = c->fc[l];
    G.gemm(fc_dw_args(c, c->ws[0], l, B, c->ws[0].dz[l]), {cdz});
    const int ncols = L.cat ? L.n_in - A : L.n_in;
    if (l > 0)
      cdz = G.gemm(fc_dx_args(c, l, B, c->ws[0].dz[l], L.n_out, 0, ncols, c->ws[0].dz[l - 1], ncols, GE_MUL_RELU_GRAD,
                              c->ws[0].fcin[l], L.n_in + 1), {cdz});
    else if (c->spec.pixel)
      cdz = G.gemm(fc_dx_args(c, 0, B, c->ws[0].dz[0], L.n_out, 0, c->flat, c->ws[0].dpool[2], c->flat, GE_NONE, nullptr, 0), {cdz});
  }
  }
  if (c->spec.pixel) {     // both conv backward passes, layer by layer, two networks per launch
    cpp_net* bn[2] = {a, c};
    G.fn([=] { return nets_backward_conv(ctx, bn, 2, B, s1, dt, w1); }, {adz, cdz});
  }
  RC(G.run(ctx));
  return flush_dw_reduce(ctx);      // all six dW reductions (3 layers x 2 networks) in one launch
}

extern "C" int cpp_ddpg_compute_gradients(cpp_ddpg* d, cpp_batch* b) {
  RC(check_batch(d, b, "cpp_ddpg_compute_gradients"));
  HIP_CHECK(hipSetDevice(d->ctx->device));
  RC(prep_batch(d, b));
  return compute_gradients(d, b);
}

extern "C" int cpp_ddpg_grad_buffer(cpp_ddpg* d, void** p, int64_t* n) {
  ARG_CHECK(d && p && n, "cpp_ddpg_grad_buffer: NULL argument");
  *p = d->gradbuf; *n = d->nA + d->nC;
  return CPP_OK;
}

extern "C" int cpp_ddpg_apply_gradients(cpp_ddpg* d, float grad_scale) {
  ARG_CHECK(d, "cpp_ddpg_apply_gradients: NULL argument");
  HIP_CHECK(hipSetDevice(d->ctx->device));
  return apply(d, true, true, grad_scale);
}

extern "C" int cpp_ddpg_update_targets(cpp_ddpg* d) {
  ARG_CHECK(d, "cpp_ddpg_update_targets: NULL argument");
  HIP_CHECK(hipSetDevice(d->ctx->device));
  return launch_soft_update(d->ctx, d->tactor->params, d->actor->params, d->nA, d->tcritic->params, d->critic->params,
                            d->nC, d->hp.target_update_rate);
}

// The fused step does not need a gathered copy of the minibatch when conv1 runs on the f16-pipe kernels: they take the
// replay store plus the sampled slots (the gather kernel then only reads -- statistics -- and writes 2 B ints).
// CPP_DIRECT_REPLAY=0 keeps the copy.
static bool direct_replay_ok(cpp_net* a, cpp_replay* r, int B) {
  static const bool off = getenv("CPP_DIRECT_REPLAY") != nullptr && atoi(getenv("CPP_DIRECT_REPLAY")) == 0;
  if (off || !a->spec.pixel || r->store_dtype != CPP_F16) return false;
  const int C = a->spec.C;
  int g = 8, c = C; while (c) { int t = g % c; g = c; c = t; }
  if (r->elems % 8 != 0 || C / g > 16 || r->elems % C != 0) return false;       // statistics come from the gather kernel
  return conv1_f16_pipes_ok(C, a->conv[0].H, a->conv[0].W, B, a->spec.use_batch_norm != 0);
}

static int step_body(cpp_ddpg* d, cpp_replay* r, int B, int n_batches, const int32_t* rows_dev, uint64_t seed) {
  d->pre_variant = 0;        // (the half steps' presampled minibatch lives in the same step_batch)
  const int C = d->actor->spec.pixel ? d->actor->spec.C : 0;
  cpp_ctx* ctx = d->ctx;
  const bool direct = direct_replay_ok(d->actor, r, B);
  // The sample + statistics pass of minibatch i + 1 depends on nothing minibatch i computes: it rides in the launch of i's
  // dW reductions (reduce_gather_kernel, replay.hip), keyed by the sampler's counter + 1 -- the counter itself moves in i's
  // optimiser kernel as before, so the rows drawn are the same.  Conv trunks on f16 / u8 stores; CPP_RIDE_GATHER=0: in sequence.
  static const bool no_ride = getenv("CPP_RIDE_GATHER") != nullptr && atoi(getenv("CPP_RIDE_GATHER")) == 0;
  const bool ride_ok = !no_ride && C > 0 && (r->store_dtype == CPP_F16 || r->store_dtype == CPP_U8);
  RC(replay_sample_device(r, B, rows_dev, seed, rows_dev ? nullptr : r->counter, C, d->step_batch, direct));
  for (int i = 0; i < n_batches; ++i) {
    GatherArgs ga; int Cg = 0;
    const bool more = i + 1 < n_batches;
    if (more && ride_ok) {
      ga = replay_gather_args(r, B, rows_dev ? rows_dev + (size_t)(i + 1) * B : nullptr, seed, rows_dev ? nullptr : r->counter, C,
                              d->step_batch, direct, &Cg);
      ga.counter_add = 1;
      // with the slots double-buffered the pass can leave as early as conv1's dW (MFMA-bound, HBM idle, and its second
      // round of workgroups leaves the CUs half empty: conv1_dw_gather.hip); otherwise it waits for the dW reductions
      static const bool no_dwride = getenv("CPP_RIDE_DW") != nullptr && atoi(getenv("CPP_RIDE_DW")) == 0;
      ctx->ride_at_dw = direct && !no_dwride;
      if (direct) { ga.out_slot[0] = d->step_batch->slot_alt[0]; ga.out_slot[1] = d->step_batch->slot_alt[1]; }
      ctx->ride = &ga; ctx->ride_done = false; ctx->ride_dtype = r->store_dtype;
    }
    const int rc = compute_gradients(d, d->step_batch);
    const bool rode = ctx->ride != nullptr && ctx->ride_done;
    ctx->ride = nullptr;
    if (rode && direct) { std::swap(d->step_batch->slot[0], d->step_batch->slot_alt[0]); std::swap(d->step_batch->slot[1], d->step_batch->slot_alt[1]); }
    RC(rc);
    RC(apply(d, true, true, 1.0f, rows_dev ? nullptr : r->counter));   // also advances the sampler's counter
    if (more) {
      if (rode) RC(replay_sample_finish(r, B, Cg, C, d->step_batch));
      else RC(replay_sample_device(r, B, rows_dev ? rows_dev + (size_t)(i + 1) * B : nullptr, seed, rows_dev ? nullptr : r->counter, C,
                                   d->step_batch, direct));
    }
  }
  return cpp_ddpg_update_targets(d);
}

extern "C" int cpp_ddpg_train_step(cpp_ddpg* d, cpp_replay* r, int B, int n_batches, const int32_t* idxs, uint64_t seed) {
  ARG_CHECK(d && r, "cpp_ddpg_train_step: NULL argument");
  ARG_CHECK(B >= 1 && B <= d->maxB, "cpp_ddpg_train_step: batch %d outside [1,%d]", B, d->maxB);
  ARG_CHECK(n_batches >= 1 && (size_t)n_batches * B <= 65536, "cpp_ddpg_train_step: n_batches %d", n_batches);
  ARG_CHECK(r->elems == d->actor->state_elems && r->A == d->actor->spec.action_dim, "cpp_ddpg_train_step: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_ddpg_train_step: replay memory is empty"); return CPP_ERR_STATE; }
  cpp_ctx* ctx = d->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!d->step_batch) RC(cpp_batch_create(ctx, d->maxB, r->elems, r->A, &d->step_batch));
  if (idxs) {
    for (int i = 0; i < n_batches * B; ++i)
      ARG_CHECK(idxs[i] >= 0 && idxs[i] < r->size, "cpp_ddpg_train_step: index %d outside [0,%d)", idxs[i], r->size);
    HIP_CHECK(hipMemcpyAsync(r->rows_in, idxs, (size_t)n_batches * B * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    return step_body(d, r, B, n_batches, r->rows_in, seed);
  }
  static const bool no_graph = getenv("CPP_NO_GRAPH") != nullptr;   // plain in-order stream launches (A/B measurements)
  if (ctx->prof || no_graph) return step_body(d, r, B, n_batches, nullptr, seed);
  if (!d->graph_ok || d->g_B != B || d->g_nb != n_batches || d->g_seed != seed || d->g_replay != r || d->g_size != r->size) {
    if (d->gexec) { (void)hipGraphExecDestroy(d->gexec); d->gexec = nullptr; }
    if (d->graph) { (void)hipGraphDestroy(d->graph); d->graph = nullptr; }
    d->graph_ok = false;
    // one eager pass first: it sets every kernel's LDS attribute (not allowed during capture)
    RC(step_body(d, r, B, n_batches, nullptr, seed));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = step_body(d, r, B, n_batches, nullptr, seed);
    hipError_t e = hipStreamEndCapture(ctx->stream, &d->graph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&d->gexec, d->graph, nullptr, nullptr, 0));
    d->graph_ok = true; d->g_B = B; d->g_nb = n_batches; d->g_seed = seed; d->g_replay = r; d->g_size = r->size;
    return CPP_OK;   // the eager pass above was this call's step
  }
  HIP_CHECK(hipGraphLaunch(d->gexec, ctx->stream));
  d->loss_parts = d->heads_grid; d->loss_B = d->heads_B;
  return CPP_OK;
}

// variant 0: sample + gather + statistics of this call's minibatch; 1 / 2: it was presampled by the previous call's rider into
// slot set 1 / 0 (only its whitening tables are still to do).  Every variant tries to send the NEXT minibatch's sample pass
// along with conv1's dW (the sampler's counter has been advanced by then, so the rider draws with the counter as it stands);
// *next: the variant the following call must use.  CPP_RIDE_DP=0: always variant 0, no rider.
static int half_step_body(cpp_ddpg* d, cpp_replay* r, int B, uint64_t seed, int variant, int* next) {
  const int C = d->actor->spec.pixel ? d->actor->spec.C : 0;
  cpp_ctx* ctx = d->ctx;
  cpp_batch* b = d->step_batch;
  const bool direct = direct_replay_ok(d->actor, r, B);
  static const bool no_ride = getenv("CPP_RIDE_DP") != nullptr && atoi(getenv("CPP_RIDE_DP")) == 0;
  const int cur = variant == 1 ? 1 : 0;
  for (int k = 0; k < 2; ++k) { b->slot[k] = d->slot_set[cur][k]; b->slot_alt[k] = d->slot_set[1 - cur][k]; }
  int Cg = 0;
  GatherArgs ga = replay_gather_args(r, B, nullptr, seed, r->counter, C, b, direct, &Cg);
  if (variant == 0) RC(launch_gather_stats(ctx, ga, r->store_dtype));
  RC(replay_sample_finish(r, B, Cg, C, b));
  RC(launch_counter_add(ctx, r->counter, 1));
  const bool ride_ok = !no_ride && direct && Cg > 0 && r->store_dtype == CPP_F16;
  if (ride_ok) {
    ga.out_slot[0] = b->slot_alt[0]; ga.out_slot[1] = b->slot_alt[1];
    ctx->ride = &ga; ctx->ride_done = false; ctx->ride_dtype = r->store_dtype; ctx->ride_at_dw = true;
  }
  const int rc = compute_gradients(d, b);
  const bool rode = ctx->ride != nullptr && ctx->ride_done;
  ctx->ride = nullptr;
  *next = rode ? (cur == 0 ? 1 : 2) : 0;
  return rc;
}

extern "C" int cpp_ddpg_sample_and_compute(cpp_ddpg* d, cpp_replay* r, int B, uint64_t seed) {
  ARG_CHECK(d && r, "cpp_ddpg_sample_and_compute: NULL argument");
  ARG_CHECK(B >= 1 && B <= d->maxB, "cpp_ddpg_sample_and_compute: batch %d outside [1,%d]", B, d->maxB);
  ARG_CHECK(r->elems == d->actor->state_elems && r->A == d->actor->spec.action_dim, "cpp_ddpg_sample_and_compute: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_ddpg_sample_and_compute: replay memory is empty"); return CPP_ERR_STATE; }
  cpp_ctx* ctx = d->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!d->step_batch) {
    RC(cpp_batch_create(ctx, d->maxB, r->elems, r->A, &d->step_batch));
    for (int k = 0; k < 2; ++k) { d->slot_set[0][k] = d->step_batch->slot[k]; d->slot_set[1][k] = d->step_batch->slot_alt[k]; }
  }
  if (d->slot_set[0][0] == nullptr)
    for (int k = 0; k < 2; ++k) { d->slot_set[0][k] = d->step_batch->slot[k]; d->slot_set[1][k] = d->step_batch->slot_alt[k]; }
  const bool key_ok = d->h_B == B && d->h_seed == seed && d->h_replay == r && d->h_size == r->size;
  if (!key_ok) {                                     // another batch size / seed / memory, or rows were added: start over
    for (int v = 0; v < 3; ++v) {
      if (d->hexec[v]) { (void)hipGraphExecDestroy(d->hexec[v]); d->hexec[v] = nullptr; }
      if (d->hgraph[v]) { (void)hipGraphDestroy(d->hgraph[v]); d->hgraph[v] = nullptr; }
      d->hgraph_ok[v] = false;
    }
    d->pre_variant = 0;
    d->h_B = B; d->h_seed = seed; d->h_replay = r; d->h_size = r->size;
  }
  const int v = d->pre_variant;
  d->pre_variant = 0;                                // (stays 0 if anything below fails)
  int next = 0;
  if (ctx->prof) { RC(half_step_body(d, r, B, seed, v, &next)); d->pre_variant = next; return CPP_OK; }
  if (!d->hgraph_ok[v]) {
    // this call's work is done by the captured graph's first launch: an eager pass first would consume the presampled batch
    // and leave another one behind.  Kernel attributes: set by the first eager variant-0 pass below.
    if (v == 0) {
      RC(half_step_body(d, r, B, seed, 0, &next));            // eager pass: sets kernel attributes, is this call's work
      HIP_CHECK(hipStreamSynchronize(ctx->stream));
      HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      int nx = 0;
      int rc = half_step_body(d, r, B, seed, 0, &nx);
      hipError_t e = hipStreamEndCapture(ctx->stream, &d->hgraph[0]);
      if (rc) return rc;
      if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
      HIP_CHECK(hipGraphInstantiate(&d->hexec[0], d->hgraph[0], nullptr, nullptr, 0));
      d->hgraph_ok[0] = true; d->h_next[0] = nx;
      d->pre_variant = next;
      return CPP_OK;
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int nx = 0;
    int rc = half_step_body(d, r, B, seed, v, &nx);
    hipError_t e = hipStreamEndCapture(ctx->stream, &d->hgraph[v]);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&d->hexec[v], d->hgraph[v], nullptr, nullptr, 0));
    d->hgraph_ok[v] = true; d->h_next[v] = nx;
  }
  HIP_CHECK(hipGraphLaunch(d->hexec[v], ctx->stream));
  d->pre_variant = d->h_next[v];
  d->loss_parts = d->heads_grid; d->loss_B = d->heads_B;
  return CPP_OK;
}

extern "C" int cpp_ddpg_last_stats(cpp_ddpg* d, float out[3]) {
  ARG_CHECK(d && out, "cpp_ddpg_last_stats: NULL argument");
  HIP_CHECK(hipMemcpyAsync(out, d->loss_norms, 3 * sizeof(float), hipMemcpyDeviceToHost, d->ctx->stream));
  double parts[DDPG_HEADS_MAX_WGS];
  if (d->loss_parts > 0)
    HIP_CHECK(hipMemcpyAsync(parts, d->heads_part, (size_t)d->loss_parts * sizeof(double), hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(d->ctx->stream));
  if (d->loss_parts > 0) {                          // fused heads kernel: mean(td^2) from its per-workgroup partials, fixed order
    double s = 0.0;
    for (int i = 0; i < d->loss_parts; ++i) s += parts[i];
    out[0] = (float)(s / (double)d->loss_B);
  }
  return CPP_OK;
}

// ---------------------------------------------------------------------------------------------
// NAF (naf_cartpole.py)
// ---------------------------------------------------------------------------------------------
struct cpp_naf {
  cpp_ctx* ctx; cpp_net *value, *tvalue, *mu, *lv; int share; cpp_naf_hyper hp;
  int maxB, A, NL; long nV, nM, nL;
  float* gradbuf; float *m, *v;           // optimiser state over the same flat layout (Momentum / Adam)
  float *adv, *q, *td, *stats;            // stats: [0] loss [1] norm
  int* nonfinite; uint64_t* opt_step; double* norm_part;
  hipGraph_t graph; hipGraphExec_t gexec; bool graph_ok; int g_B, g_nb, g_size; uint64_t g_seed; cpp_replay* g_replay;   // g_size: rows in the replay when captured (the sampler's range is a kernel argument)
  cpp_batch* step_batch;
  Arena arena;
};

extern "C" int cpp_naf_create(cpp_ctx* ctx, cpp_net* value, cpp_net* tvalue, cpp_net* mu, cpp_net* lv, int share,
                              const cpp_naf_hyper* hp, cpp_naf** out) {
  ARG_CHECK(ctx && value && tvalue && mu && lv && hp && out, "cpp_naf_create: NULL argument");
  for (cpp_net* n : {value, tvalue, mu, lv}) ARG_CHECK(n->spec.kind == CPP_HEAD, "cpp_naf_create: networks must be CPP_HEAD");
  ARG_CHECK(value->spec.head_out == 1 && tvalue->spec.head_out == 1 && value->nparams == tvalue->nparams,
            "cpp_naf_create: value / target_value shapes");
  const int A = mu->spec.head_out;
  ARG_CHECK(A >= 1 && A <= 8 && lv->spec.head_out == A * (A + 1) / 2, "cpp_naf_create: mu has %d outputs, l_values %d (want A and A(A+1)/2)",
            A, lv->spec.head_out);
  ARG_CHECK(mu->spec.head_act == 2 && lv->spec.head_act == 0 && value->spec.head_act == 0, "cpp_naf_create: head activations");
  ARG_CHECK(hp->optimiser >= CPP_OPT_SGD && hp->optimiser <= CPP_OPT_ADAM, "cpp_naf_create: optimiser %d", hp->optimiser);
  const int rep = value->fc.back().n_in;
  if (share) {
    for (cpp_net* n : {mu, lv})
      ARG_CHECK(!n->spec.pixel && n->fc.size() == 1 && n->fc[0].n_in == rep,
                "cpp_naf_create: shared heads must be head-only nets over the %d-wide representation", rep);
  } else {
    for (cpp_net* n : {mu, lv}) ARG_CHECK(n->state_elems == value->state_elems, "cpp_naf_create: state shapes differ");
  }
  HIP_CHECK(hipSetDevice(ctx->device));
  cpp_naf* f = new cpp_naf();
  f->arena.stream = ctx->stream;
  f->ctx = ctx; f->value = value; f->tvalue = tvalue; f->mu = mu; f->lv = lv; f->share = share; f->hp = *hp;
  f->maxB = value->maxB; f->A = A; f->NL = A * (A + 1) / 2;
  for (cpp_net* n : {tvalue, mu, lv}) if (n->maxB < f->maxB) f->maxB = n->maxB;
  f->nV = value->nparams; f->nM = mu->nparams; f->nL = lv->nparams;
  f->graph = nullptr; f->gexec = nullptr; f->graph_ok = false; f->step_batch = nullptr; f->g_replay = nullptr;
  const size_t nall = (size_t)(f->nV + f->nM + f->nL);
  int rc = dalloc(f->arena, &f->gradbuf, nall);
  if (!rc) rc = dalloc(f->arena, &f->m, nall);
  if (!rc) rc = dalloc(f->arena, &f->v, nall);
  if (!rc) rc = dalloc(f->arena, &f->adv, (size_t)f->maxB);
  if (!rc) rc = dalloc(f->arena, &f->q, (size_t)f->maxB);
  if (!rc) rc = dalloc(f->arena, &f->td, (size_t)f->maxB);
  if (!rc) rc = dalloc(f->arena, &f->stats, (size_t)4);
  if (!rc) rc = dalloc(f->arena, &f->nonfinite, (size_t)1);
  if (!rc) rc = dalloc(f->arena, &f->opt_step, (size_t)1);
  if (!rc) rc = dalloc(f->arena, &f->norm_part, (size_t)OPT_MAX_SEGS * NORM_PARTS);
  if (rc) { f->arena.release(); delete f; return rc; }
  value->grads = f->gradbuf; mu->grads = f->gradbuf + f->nV; lv->grads = f->gradbuf + f->nV + f->nM;
  if (share) {      // the heads read value's input_state_representation in place (naf_cartpole.py:151-152,176-177)
    mu->ws[0].fcin[0] = value->ws[0].fcin.back();
    lv->ws[0].fcin[0] = value->ws[0].fcin.back();
  }
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  *out = f;
  return CPP_OK;
}

extern "C" int cpp_naf_destroy(cpp_naf* f) {
  if (!f) return CPP_OK;
  (void)hipSetDevice(f->ctx->device);
  (void)hipStreamSynchronize(f->ctx->stream);
  if (f->gexec) (void)hipGraphExecDestroy(f->gexec);
  if (f->graph) (void)hipGraphDestroy(f->graph);
  if (f->step_batch) cpp_batch_destroy(f->step_batch);
  f->value->grads = nullptr; f->mu->grads = nullptr; f->lv->grads = nullptr;
  f->arena.release(); delete f; return CPP_OK;
}

static int naf_check_batch(cpp_naf* f, cpp_batch* b, const char* who) {
  ARG_CHECK(f && b, "%s: NULL argument", who);
  ARG_CHECK(b->B >= 1 && b->B <= f->maxB, "%s: batch size %d outside [1,%d]", who, b->B, f->maxB);
  ARG_CHECK(b->elems == f->value->state_elems && b->A == f->A, "%s: batch shape does not match the networks", who);
  if (f->value->spec.pixel) RC(batch_ensure_stats(b, f->value->spec.C));
  return CPP_OK;
}

// value / mu / l_values on state_1 (device pointer), optionally V'(state_2)
static int naf_forward(cpp_naf* f, const void* s1, const void* s2, int dtype, const float* w1, const float* w2, int B) {
  cpp_net *v = f->value, *tv = f->tvalue;
  RC(net_forward_trunk(v, v->ws[0], s1, dtype, w1, B));
  RC(net_forward_fc(v, v->ws[0], 0, B, nullptr));
  for (cpp_net* n : {f->mu, f->lv}) {
    if (!f->share) RC(net_forward_trunk(n, n->ws[0], s1, dtype, w1, B));
    RC(net_forward_fc(n, n->ws[0], 0, B, nullptr));
  }
  if (s2) {
    RC(net_forward_trunk(tv, tv->ws[0], s2, dtype, w2, B));
    RC(net_forward_fc(tv, tv->ws[0], 0, B, nullptr));
  }
  return CPP_OK;
}

static int naf_head(cpp_naf* f, cpp_batch* b, bool backward) {
  NafHeadArgs a; memset(&a, 0, sizeof(a));
  a.value = f->value->ws[0].out; a.mu = f->mu->ws[0].out; a.lv = f->lv->ws[0].out;
  a.action = b->a; a.reward = b->r; a.mask = b->m; a.target_value = f->tvalue->ws[0].out;
  a.discount = f->hp.discount; a.B = b->B; a.A = f->A;
  a.adv = f->adv; a.q = f->q; a.td = f->td; a.loss = f->stats; a.nonfinite = f->nonfinite;
  if (backward) {
    a.d_value = f->value->ws[0].dz.back(); a.d_mu_z = f->mu->ws[0].dz.back(); a.d_l = f->lv->ws[0].dz.back();
  }
  return launch_naf_head(f->ctx, a);
}

// backward of the fully connected stack of a network without an action splice, from layer `start` down:
// per layer {[dW;db], dX} as two independent GEMMs.  Returns the op that completes d(flat) (pixel) / dz[0].
static int add_fc_backward(OpGraph& G, cpp_net* n, Workspace& w, int B, int start, int dep) {
  for (int l = start; l >= 0; --l) {
    const FcL& L = n->fc[l];
    G.gemm(fc_dw_args(n, w, l, B, w.dz[l]), {dep});
    if (l > 0)
      dep = G.gemm(fc_dx_args(n, l, B, w.dz[l], L.n_out, 0, L.n_in, w.dz[l - 1], L.n_in, relu_grad_epi(n, l - 1), w.fcin[l], L.n_in + 1), {dep});
    else if (n->spec.pixel)
      dep = G.gemm(fc_dx_args(n, 0, B, w.dz[0], L.n_out, 0, n->flat, w.dpool[2], n->flat, GE_NONE, nullptr, 0), {dep});
  }
  return dep;
}

// One NAF minibatch (naf_cartpole.py:264-272 without the apply) as a dependency graph, batched like the DDPG
// step: conv layers of the networks that run them share launches, independent GEMMs share launches.
static int naf_compute_gradients(cpp_naf* f, cpp_batch* b) {
  cpp_ctx* ctx = f->ctx;
  cpp_net *v = f->value, *tv = f->tvalue, *mu = f->mu, *lv = f->lv;
  const int B = b->B, C = v->spec.pixel ? v->spec.C : 0, dt = b->dtype;
  const float *w1 = white_of(b, 0, C), *w2 = white_of(b, 1, C);
  const void *s1 = b->direct_store ? b->direct_store : b->s[0], *s2 = b->direct_store ? b->direct_store : b->s[1];
  struct SlotScope {      // conv1 addresses its images through the sampled slots while this graph runs
    cpp_net* n[4];
    SlotScope(cpp_net* v_, cpp_net* mu_, cpp_net* lv_, cpp_net* tv_, cpp_batch* b_) : n{v_, mu_, lv_, tv_} {
      if (b_->direct_store) { v_->img_slot = mu_->img_slot = lv_->img_slot = b_->slot[0]; tv_->img_slot = b_->slot[1]; }
    }
    ~SlotScope() { for (cpp_net* x : n) x->img_slot = nullptr; }
  } slot_scope(v, mu, lv, tv, b);
  const bool share = f->share != 0;
  OpGraph G;

  // ---- forward trunks
  cpp_net* tn[4]; const void* ts[4]; const float* tw[4]; int nt = 0;
  tn[nt] = v; ts[nt] = s1; tw[nt] = w1; ++nt;
  if (!share) { tn[nt] = mu; ts[nt] = s1; tw[nt] = w1; ++nt; tn[nt] = lv; ts[nt] = s1; tw[nt] = w1; ++nt; }
  tn[nt] = tv; ts[nt] = s2; tw[nt] = w2; ++nt;
  int t1;
  if (v->spec.pixel && !v->spec.use_batch_norm) {
    std::vector<cpp_net*> nets(tn, tn + nt); std::vector<const void*> sts(ts, ts + nt); std::vector<const float*> whs(tw, tw + nt);
    t1 = G.fn([=] {
      for (int k = 0; k < nt; ++k) nets[k]->use_b16 = trunk_b16(nets[k], dt, B, 0);
      for (int i = 0; i < 3; ++i) {
        ConvArgs cl[CONV_BATCH_MAX]; int mode = 0;
        for (int k = 0; k < nt; ++k) cl[k] = conv_fwd_args(nets[k], nets[k]->ws[0], i, sts[k], dt, whs[k], B, &mode);
        RC(launch_conv_fwd_multi(ctx, kFwdKid[i], v->conv[i].Cin, v->conv[i].ks, mode, EPI_RELU_POOL, cl, nt));
      }
      return (int)CPP_OK; }, {});
  } else {
    std::vector<cpp_net*> nets(tn, tn + nt); std::vector<const void*> sts(ts, ts + nt); std::vector<const float*> whs(tw, tw + nt);
    t1 = G.fn([=] {        // low-dim states, or batch-norm trunks (training mode: naf_cartpole.py:271)
      if (nets[0]->spec.pixel) return nets_forward_trunk_bn(ctx, nets.data(), nt, sts.data(), whs.data(), dt, B);
      for (int k = 0; k < nt; ++k) RC(net_forward_trunk(nets[k], nets[k]->ws[0], sts[k], dt, whs[k], B));
      return (int)CPP_OK; }, {});
  }
  // ---- forward MLPs
  auto chain = [&](cpp_net* n, int from, int dep) {
    for (int l = from; l < (int)n->fc.size(); ++l) dep = G.gemm(fc_fwd_args(n, n->ws[0], l, B), {dep});
    return dep;
  };
  const int Lh = (int)v->fc.size() - 1;                 // value's 'fc' head
  int vrep = t1;
  for (int l = 0; l < Lh; ++l) vrep = G.gemm(fc_fwd_args(v, v->ws[0], l, B), {vrep});
  const int vout = G.gemm(fc_fwd_args(v, v->ws[0], Lh, B), {vrep});
  const int tvout = chain(tv, 0, t1);
  const int muout = share ? chain(mu, 0, vrep) : chain(mu, 0, t1);
  const int lvout = share ? chain(lv, 0, vrep) : chain(lv, 0, t1);
  if (v->drop_counter) {     // --use-dropout: count this training-mode forward of every network with a hidden stack
    G.fn([=] { return bump_dropout(v); }, {vout});
    G.fn([=] { return bump_dropout(tv); }, {tvout});
    if (!share) { G.fn([=] { return bump_dropout(mu); }, {muout}); G.fn([=] { return bump_dropout(lv); }, {lvout}); }
  }
  // ---- NAF head: L, advantage, TD loss and the gradients of the three head outputs
  const int head = G.fn([=] { return naf_head(f, b, true); }, {vout, tvout, muout, lvout});

  // ---- backward
  if (!share) {
    const int dv = add_fc_backward(G, v, v->ws[0], B, (int)v->fc.size() - 1, head);
    const int dm = add_fc_backward(G, mu, mu->ws[0], B, (int)mu->fc.size() - 1, head);
    const int dl = add_fc_backward(G, lv, lv->ws[0], B, (int)lv->fc.size() - 1, head);
    if (v->spec.pixel) {
      cpp_net* bn[3] = {v, mu, lv};
      G.fn([=] { return nets_backward_conv(ctx, bn, 3, B, s1, dt, w1); }, {dv, dm, dl});
    }
  } else {
    // shared representation: head gradients, then d(rep) = sum of the three heads' contributions
    const FcL& hv = v->fc[Lh];
    const int rep = hv.n_in;
    struct Head { cpp_net* n; const FcL* L; const float* dz; };
    Head heads[3] = {{v, &hv, v->ws[0].dz[Lh]}, {mu, &mu->fc[0], mu->ws[0].dz[0]}, {lv, &lv->fc[0], lv->ws[0].dz[0]}};
    float* drep = nullptr; long ldd = rep; int final_epi = GE_NONE; const float* Y = nullptr; long ldy = 0;
    if (Lh > 0) { drep = v->ws[0].dz[Lh - 1]; final_epi = relu_grad_epi(v, Lh - 1); Y = v->ws[0].fcin[Lh]; ldy = rep + 1; }
    else if (v->spec.pixel) { drep = v->ws[0].dpool[2]; }
    int dep = head;
    for (int k = 0; k < 3; ++k) {
      const Head& h = heads[k];
      const float* x = v->ws[0].fcin[Lh];        // [rep, 1] rows, shared by the three heads
      G.gemm(mk_gemm(x, 1, rep + 1, h.dz, h.L->n_out, 1, h.n->grads + h.L->w_off, h.L->n_out, rep + 1, h.L->n_out, B, GE_NONE), {head});
      if (drep) {
        GemmArgs g = mk_gemm(h.dz, h.L->n_out, 1, h.n->params + h.L->w_off, 1, h.L->n_out, drep, ldd, B, rep, h.L->n_out,
                             k == 2 ? final_epi : GE_NONE, k == 2 ? Y : nullptr, ldy);
        g.accumulate = k > 0;
        dep = G.gemm(g, {dep});                  // accumulation order value -> mu -> l_values is fixed
      }
    }
    const int dv = add_fc_backward(G, v, v->ws[0], B, Lh - 1, dep);
    if (v->spec.pixel) G.fn([=] { return net_backward_conv(v, v->ws[0], B, s1, dt, w1); }, {dv});
  }
  RC(G.run(ctx));
  return flush_dw_reduce(ctx);
}

static int naf_apply(cpp_naf* f, float grad_scale) {
  OptSegs s; memset(&s, 0, sizeof(s));
  s.nseg = 3; s.kind = f->hp.optimiser; s.momentum = f->hp.momentum; s.beta1 = f->hp.beta1; s.beta2 = f->hp.beta2;
  s.epsilon = f->hp.epsilon; s.step = f->opt_step;
  cpp_net* nets[3] = {f->value, f->mu, f->lv};
  long off = 0;
  for (int k = 0; k < 3; ++k) {
    s.p[k] = nets[k]->params; s.g[k] = f->gradbuf + off; s.m[k] = f->m + off; s.v[k] = f->v + off;
    s.n[k] = nets[k]->nparams; s.lr[k] = f->hp.learning_rate; s.group[k] = 0;      // ONE list, one global norm
    off += nets[k]->nparams;
  }
  RC(launch_counter_add(f->ctx, f->opt_step, 1));
  RC(launch_sumsq(f->ctx, s, grad_scale, f->norm_part, NORM_PARTS));
  return launch_opt_apply(f->ctx, s, grad_scale, f->hp.gradient_clip, f->norm_part, NORM_PARTS, f->stats + 1);
}

extern "C" int cpp_naf_action(cpp_naf* f, const void* state, int dtype, int B, float* out) {
  ARG_CHECK(f && state && out, "cpp_naf_action: NULL argument");
  ARG_CHECK(B >= 1 && B <= f->maxB, "cpp_naf_action: batch %d outside [1,%d]", B, f->maxB);
  ARG_CHECK(dtype == CPP_F32 || dtype == CPP_F16, "cpp_naf_action: dtype %d", dtype);
  cpp_ctx* ctx = f->ctx;
  cpp_net* n = f->share ? f->value : f->mu;       // the network whose trunk sees the state
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!n->stage_state) {
    RC(n->arena.alloc(&n->stage_state, (size_t)n->maxB * n->state_elems * sizeof(float), false));
    RC(dalloc(n->arena, &n->stage_action, (size_t)n->maxB * f->A));
  }
  HIP_CHECK(hipMemcpyAsync(n->stage_state, state, (size_t)B * n->state_elems * (dtype == CPP_F16 ? 2 : 4), hipMemcpyHostToDevice, ctx->stream));
  if (n->spec.pixel) RC(batch_stats(ctx, n->stage_state, nullptr, dtype, n->state_elems, B, n->spec.C, n->stats_part, n->white));
  n->is_training = false; f->mu->is_training = false;  // IS_TRAINING: False (naf_cartpole.py:253)
  int frc = net_forward_trunk(n, n->ws[0], n->stage_state, dtype, n->white, B);
  if (!frc) frc = net_forward_fc(n, n->ws[0], 0, B, nullptr);
  if (!frc && f->share) frc = net_forward_fc(f->mu, f->mu->ws[0], 0, B, nullptr);
  n->is_training = true; f->mu->is_training = true;
  if (frc) return frc;
  HIP_CHECK(hipMemcpyAsync(out, f->mu->ws[0].out, (size_t)B * f->A * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return CPP_OK;
}

extern "C" int cpp_naf_compute_gradients(cpp_naf* f, cpp_batch* b) {
  RC(naf_check_batch(f, b, "cpp_naf_compute_gradients"));
  HIP_CHECK(hipSetDevice(f->ctx->device));
  return naf_compute_gradients(f, b);
}
extern "C" int cpp_naf_grad_buffer(cpp_naf* f, void** p, int64_t* n) {
  ARG_CHECK(f && p && n, "cpp_naf_grad_buffer: NULL argument");
  *p = f->gradbuf; *n = f->nV + f->nM + f->nL;
  return CPP_OK;
}
extern "C" int cpp_naf_apply_gradients(cpp_naf* f, float grad_scale) {
  ARG_CHECK(f, "cpp_naf_apply_gradients: NULL argument");
  HIP_CHECK(hipSetDevice(f->ctx->device));
  return naf_apply(f, grad_scale);
}
extern "C" int cpp_naf_update_targets(cpp_naf* f) {
  ARG_CHECK(f, "cpp_naf_update_targets: NULL argument");
  HIP_CHECK(hipSetDevice(f->ctx->device));
  return launch_soft_update(f->ctx, f->tvalue->params, f->value->params, f->nV, nullptr, nullptr, 0, f->hp.target_update_rate);
}

extern "C" int cpp_naf_train(cpp_naf* f, cpp_batch* b, float* loss) {
  RC(naf_check_batch(f, b, "cpp_naf_train"));
  cpp_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  HIP_CHECK(hipMemsetAsync(f->nonfinite, 0, sizeof(int), ctx->stream));
  RC(naf_compute_gradients(f, b));
  int bad = 0; float l = 0.f;
  HIP_CHECK(hipMemcpyAsync(&bad, f->nonfinite, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipMemcpyAsync(&l, f->stats, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (loss) *loss = l;
  if (bad) { cpp_set_error("check_numerics: l_values / L / loss is not finite (naf_cartpole.py:242-245)"); return CPP_ERR_NUMERIC; }
  return naf_apply(f, 1.0f);
}

extern "C" int cpp_naf_debug_values(cpp_naf* f, cpp_batch* b, float* l_values, float* loss, float* value, float* advantage,
                                    float* target_value) {
  RC(naf_check_batch(f, b, "cpp_naf_debug_values"));
  cpp_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  const int B = b->B, C = f->value->spec.pixel ? f->value->spec.C : 0;
  for (cpp_net* n : {f->value, f->tvalue, f->mu, f->lv}) n->is_training = false;      // IS_TRAINING: False (naf_cartpole.py:282)
  const int frc = naf_forward(f, b->s[0], b->s[1], b->dtype, white_of(b, 0, C), white_of(b, 1, C), B);
  for (cpp_net* n : {f->value, f->tvalue, f->mu, f->lv}) n->is_training = true;
  if (frc) return frc;
  RC(naf_head(f, b, false));
  hipStream_t st = ctx->stream;
  if (l_values) HIP_CHECK(hipMemcpyAsync(l_values, f->lv->ws[0].out, (size_t)B * f->NL * sizeof(float), hipMemcpyDeviceToHost, st));
  if (loss) HIP_CHECK(hipMemcpyAsync(loss, f->stats, sizeof(float), hipMemcpyDeviceToHost, st));
  if (value) HIP_CHECK(hipMemcpyAsync(value, f->value->ws[0].out, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, st));
  if (advantage) HIP_CHECK(hipMemcpyAsync(advantage, f->adv, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, st));
  if (target_value) HIP_CHECK(hipMemcpyAsync(target_value, f->tvalue->ws[0].out, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

static int naf_step_body(cpp_naf* f, cpp_replay* r, int B, int n_batches, const int32_t* rows_dev, uint64_t seed) {
  const int C = f->value->spec.pixel ? f->value->spec.C : 0;
  for (int i = 0; i < n_batches; ++i) {
    RC(replay_sample_device(r, B, rows_dev ? rows_dev + (size_t)i * B : nullptr, seed, rows_dev ? nullptr : r->counter, C, f->step_batch,
                            direct_replay_ok(f->value, r, B)));
    if (!rows_dev) RC(launch_counter_add(f->ctx, r->counter, 1));
    RC(naf_compute_gradients(f, f->step_batch));
    RC(naf_apply(f, 1.0f));
  }
  return cpp_naf_update_targets(f);
}

extern "C" int cpp_naf_train_step(cpp_naf* f, cpp_replay* r, int B, int n_batches, const int32_t* idxs, uint64_t seed) {
  ARG_CHECK(f && r, "cpp_naf_train_step: NULL argument");
  ARG_CHECK(B >= 1 && B <= f->maxB, "cpp_naf_train_step: batch %d outside [1,%d]", B, f->maxB);
  ARG_CHECK(n_batches >= 1 && (size_t)n_batches * B <= 65536, "cpp_naf_train_step: n_batches %d", n_batches);
  ARG_CHECK(r->elems == f->value->state_elems && r->A == f->A, "cpp_naf_train_step: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_naf_train_step: replay memory is empty"); return CPP_ERR_STATE; }
  cpp_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!f->step_batch) RC(cpp_batch_create(ctx, f->maxB, r->elems, r->A, &f->step_batch));
  if (idxs) {
    for (int i = 0; i < n_batches * B; ++i)
      ARG_CHECK(idxs[i] >= 0 && idxs[i] < r->size, "cpp_naf_train_step: index %d outside [0,%d)", idxs[i], r->size);
    HIP_CHECK(hipMemcpyAsync(r->rows_in, idxs, (size_t)n_batches * B * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    return naf_step_body(f, r, B, n_batches, r->rows_in, seed);
  }
  if (ctx->prof) return naf_step_body(f, r, B, n_batches, nullptr, seed);
  if (!f->graph_ok || f->g_B != B || f->g_nb != n_batches || f->g_seed != seed || f->g_replay != r || f->g_size != r->size) {
    if (f->gexec) { (void)hipGraphExecDestroy(f->gexec); f->gexec = nullptr; }
    if (f->graph) { (void)hipGraphDestroy(f->graph); f->graph = nullptr; }
    f->graph_ok = false;
    RC(naf_step_body(f, r, B, n_batches, nullptr, seed));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = naf_step_body(f, r, B, n_batches, nullptr, seed);
    hipError_t e = hipStreamEndCapture(ctx->stream, &f->graph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&f->gexec, f->graph, nullptr, nullptr, 0));
    f->graph_ok = true; f->g_B = B; f->g_nb = n_batches; f->g_seed = seed; f->g_replay = r; f->g_size = r->size;
    return CPP_OK;
  }
  HIP_CHECK(hipGraphLaunch(f->gexec, ctx->stream));
  return CPP_OK;
}

extern "C" int cpp_naf_last_stats(cpp_naf* f, float out[3]) {
  ARG_CHECK(f && out, "cpp_naf_last_stats: NULL argument");
  int bad = 0;
  HIP_CHECK(hipMemcpyAsync(out, f->stats, 2 * sizeof(float), hipMemcpyDeviceToHost, f->ctx->stream));
  HIP_CHECK(hipMemcpyAsync(&bad, f->nonfinite, sizeof(int), hipMemcpyDeviceToHost, f->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
  out[2] = (float)bad;
  return CPP_OK;
}
