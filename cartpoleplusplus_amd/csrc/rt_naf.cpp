// NAF train ops and inner step (cpp_naf_*)
#include <utility>
#include "rt_internal.h"

// ---------------------------------------------------------------------------------------------
// NAF (naf_cartpole.py)
// ---------------------------------------------------------------------------------------------
struct cpp_naf {
  cpp_ctx* ctx; cpp_net *value, *tvalue, *mu, *lv; int share; cpp_naf_hyper hp;
  int maxB, A, NL; long nV, nM, nL;
  float* gradbuf; float *m, *v;           // optimiser state over the same flat layout (Momentum / Adam)
  float *adv, *q, *td, *stats;            // stats: [0] loss [1] norm
  int* nonfinite; uint64_t* opt_step; double* norm_part;
  double* heads_part; unsigned* heads_ticket;      // fused heads kernel (naf_heads_kernel): per-workgroup td^2 sums + bad flags, arrival counter
  int sq_cnt;              // norm partials the last folded gradient pass left in cpp_ctx::sq_part (<= 0: none, run the sumsq kernel)
  bool step_bumped;        // the last gradient pass advanced opt_step in its heads kernel (the next apply must not)
  hipGraph_t graph; hipGraphExec_t gexec; bool graph_ok; int g_B, g_nb; uint64_t g_seed, g_replay_uid;
  // the data-parallel half step (sample + gradients) as a graph of its own
  hipGraph_t hgraph; hipGraphExec_t hexec; bool hgraph_ok; int h_B; uint64_t h_seed, h_replay_uid;
  // ONE minibatch on host-drawn rows up to (not including) the optimiser (cpp_naf_train_rows)
  hipGraph_t rgraph; hipGraphExec_t rgexec; bool rgraph_ok; int rg_B; uint64_t rg_replay_uid;
  hipGraph_t dgraph; hipGraphExec_t dgexec; bool dgraph_ok; int dg_B, dg_nb; uint64_t dg_seed, dg_replay_uid; uint64_t dg_comm_uid; bool dgraph_refused; char dg_reason[256];   // the data-parallel step
  // ... and including it, the loss coming back later (cpp_naf_train_rows_async / cpp_naf_loss_wait): pinned (loss, flag) slots
  hipGraph_t agraph; hipGraphExec_t agexec; bool agraph_ok; int ag_B; uint64_t ag_replay_uid;
  uint64_t epoch;            // cpp_ctx::kernel_epoch the cached graphs were captured under (naf_route_check)
  bool targets_in_apply, targets_applied;      // the next naf_apply closes an outer step: its launch carries the target update (rt_ddpg.cpp's twin)
  float* res_pin; hipEvent_t res_ev[CPP_NAF_TICKETS]; uint64_t next_ticket;
  uint64_t dp_local;       // minibatches applied locally since the last parameter averaging (periodic mode)
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
  f->graph = nullptr; f->gexec = nullptr; f->graph_ok = false; f->step_batch = nullptr; f->g_replay_uid = 0;
  f->sq_cnt = 0; f->step_bumped = false;
  f->rgraph = nullptr; f->rgexec = nullptr; f->rgraph_ok = false; f->rg_B = 0; f->rg_replay_uid = 0;
  f->dgraph = nullptr; f->dgexec = nullptr; f->dgraph_ok = false; f->dg_B = f->dg_nb = 0; f->dg_seed = f->dg_replay_uid = 0; f->dg_comm_uid = 0; f->dgraph_refused = false; f->dg_reason[0] = 0;
  f->agraph = nullptr; f->agexec = nullptr; f->agraph_ok = false; f->ag_B = 0; f->ag_replay_uid = 0;
  f->epoch = ctx->kernel_epoch;
  f->res_pin = nullptr; f->next_ticket = 0; memset(f->res_ev, 0, sizeof(f->res_ev));
  f->hgraph = nullptr; f->hexec = nullptr; f->hgraph_ok = false; f->h_B = 0; f->h_seed = 0; f->h_replay_uid = 0; f->dp_local = 0;
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
  if (!rc) rc = dalloc(f->arena, &f->heads_part, (size_t)2 * NAF_HEADS_MAX_WGS);
  if (!rc) rc = dalloc(f->arena, &f->heads_ticket, (size_t)1);
  if (rc) { f->arena.release(); delete f; return rc; }
  value->grads = f->gradbuf; mu->grads = f->gradbuf + f->nV; lv->grads = f->gradbuf + f->nV + f->nM;
  if (share) {      // the heads read value's input_state_representation in place (naf_cartpole.py:151-152,176-177)
    mu->ws[0].fcin[0] = value->ws[0].fcin.back();
    lv->ws[0].fcin[0] = value->ws[0].fcin.back();
  }
  HIP_CHECK(ctx_sync_stream(ctx));
  ctx->n_trainers += 1;
  *out = f;
  return CPP_OK;
}

extern "C" int cpp_naf_destroy(cpp_naf* f) {
  if (!f) return CPP_OK;
  (void)hipSetDevice(f->ctx->device);
  (void)ctx_sync_stream(f->ctx);
  if (f->hexec) (void)hipGraphExecDestroy(f->hexec);
  if (f->hgraph) (void)hipGraphDestroy(f->hgraph);
  if (f->dgexec) (void)hipGraphExecDestroy(f->dgexec);
  if (f->dgraph) (void)hipGraphDestroy(f->dgraph);
  if (f->rgexec) (void)hipGraphExecDestroy(f->rgexec);
  if (f->rgraph) (void)hipGraphDestroy(f->rgraph);
  if (f->agexec) (void)hipGraphExecDestroy(f->agexec);
  if (f->agraph) (void)hipGraphDestroy(f->agraph);
  if (f->res_pin) (void)hipHostFree(f->res_pin);
  for (hipEvent_t e : f->res_ev) if (e) (void)hipEventDestroy(e);
  if (f->gexec) (void)hipGraphExecDestroy(f->gexec);
  if (f->graph) (void)hipGraphDestroy(f->graph);
  if (f->step_batch) cpp_batch_destroy(f->step_batch);
  f->value->grads = nullptr; f->mu->grads = nullptr; f->lv->grads = nullptr;
  f->ctx->n_trainers -= 1;
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
static int add_fc_backward(OpGraph& G, cpp_net* n, Workspace& w, int B, int start, int dep,
                           const std::function<GemmArgs(GemmArgs)>& dw = [](GemmArgs g) { return g; }) {
  for (int l = start; l >= 0; --l) {
    const FcL& L = n->fc[l];
    G.gemm(dw(fc_dw_args(n, w, l, B, w.dz[l])), {dep});
    if (l > 0)
      dep = G.gemm(fc_dx_args(n, l, B, w.dz[l], L.n_out, 0, L.n_in, w.dz[l - 1], L.n_in, relu_grad_epi(n, l - 1), w.fcin[l], L.n_in + 1), {dep});
    else if (n->spec.pixel)
      dep = G.gemm(fc_dx_args(n, 0, B, w.dz[0], L.n_out, 0, n->flat, w.dpool[2], n->flat, GE_NONE, nullptr, 0), {dep});
  }
  return dep;
}

// One NAF minibatch (naf_cartpole.py:264-272 without the apply) as a dependency graph, batched like the DDPG
// step: conv layers of the networks that run them share launches, independent GEMMs share launches.
// fold (the fused single-learner step only: the gradients are applied as computed): the kernels that write the gradients leave
// their share of the list's squared norm in cpp_ctx::sq_part (as the DDPG step, rt_ddpg.cpp) and the heads kernel advances the
// optimiser's step counter -- naf_apply then runs neither the sumsq nor the counter kernel
// bump_step: the heads kernel also advances the optimiser's step counter (only where no check_numerics flag can stand the update down)
static int naf_compute_gradients(cpp_naf* f, cpp_batch* b, bool fold = false, bool bump_step = true) {
  cpp_ctx* ctx = f->ctx;
  f->sq_cnt = 0; f->step_bumped = false;
  struct SqScope {
    cpp_ctx* c; cpp_naf* f;
    SqScope(cpp_ctx* c_, cpp_naf* f_, bool on) : c(c_), f(f_) {
      if (on) { c->sq_n[0] = 0; c->sq_n[1] = -1; for (int& g : c->sq_conv_group) g = 0; }
    }
    ~SqScope() {
      if (c->sq_n[0] > 0) f->sq_cnt = c->sq_n[0];
      c->sq_n[0] = c->sq_n[1] = -1;
      for (int& g : c->sq_conv_group) g = -1;
    }
  } sq_scope(ctx, f, fold && !f->value->spec.use_batch_norm);
  auto sqg = [ctx](GemmArgs g) {
    const int tiles = gemm_tiles(g.M, g.N, g.K);
    if (ctx->sq_n[0] >= 0 && ctx->sq_n[0] + tiles <= SQ_REGION) { g.sq_part = ctx->sq_part + ctx->sq_n[0]; ctx->sq_n[0] += tiles; }
    else ctx->sq_n[0] = -1;
    return g;
  };
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
    // one launch per conv layer for all trunks, conv3 as conv2's tail, the target network forward-only (rt_net.cpp)
    t1 = G.fn([=] { return nets_forward_trunk_fused(ctx, nets.data(), nt, sts.data(), whs.data(), nt - 1, dt, B); }, {});
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
  // Shared representation with a hidden stack (the reference's pixel NAF, naf_cartpole.py:151-152): the four head layers, the
  // head arithmetic and d(representation) are ONE row-local launch (naf_heads_kernel, gemm.hip) instead of a forward GEMM level,
  // the head kernel and three dependent GEMM levels.  CPP_NAF_HEADS=0 (ablation build): the GEMM levels.
  static const bool no_fused_heads = cpp_switch_off("CPP_NAF_HEADS");
  NafHeadsArgs nh; memset(&nh, 0, sizeof(nh));
  bool fused = share && Lh > 0 && !no_fused_heads;
  if (fused) {
    const FcL& hv = v->fc[Lh];
    nh.B = B; nh.A = f->A; nh.rep = hv.n_in; nh.discount = f->hp.discount;
    nh.x = v->ws[0].fcin[Lh]; nh.xt = tv->ws[0].fcin[Lh]; nh.ldx = hv.n_in + 1;
    nh.Wv = v->params + hv.w_off; nh.Wmu = mu->params + mu->fc[0].w_off; nh.Wl = lv->params + lv->fc[0].w_off; nh.Wvt = tv->params + tv->fc[Lh].w_off;
    nh.action = b->a; nh.reward = b->r; nh.mask = b->m;
    nh.value = v->ws[0].out; nh.mu = mu->ws[0].out; nh.lv = lv->ws[0].out; nh.target_value = tv->ws[0].out;
    nh.adv = f->adv; nh.q = f->q; nh.td = f->td; nh.loss = f->stats; nh.nonfinite = f->nonfinite;
    nh.d_value = v->ws[0].dz[Lh]; nh.d_mu_z = mu->ws[0].dz[0]; nh.d_l = lv->ws[0].dz[0];
    nh.drep = v->ws[0].dz[Lh - 1]; nh.ldd = hv.n_in; nh.epi = relu_grad_epi(v, Lh - 1); nh.Y = v->ws[0].fcin[Lh]; nh.ldy = hv.n_in + 1;
    nh.part = f->heads_part; nh.ticket = f->heads_ticket;
    nh.step_bump = (fold && bump_step) ? (unsigned long long*)f->opt_step : nullptr;
    fused = naf_heads_supported(nh) && (nh.epi == GE_MUL_RELU_GRAD || nh.epi == GE_MUL_RELU_GRAD_X2);
  }
  // ... and with exactly two hidden layers (the reference's 100, 50) the second one joins that launch, forward and backward, on the
  // matrix pipes (naf_mlp_kernel): one forward and one backward GEMM level are left.  CPP_NAF_MLP=0 (ablation build): naf_heads_kernel.
  static const bool no_mlp = cpp_switch_off("CPP_NAF_MLP");
  NafMlpArgs nm; memset(&nm, 0, sizeof(nm));
  bool mlp = fused && !no_mlp && Lh == 2 && v->fc[0].act == GE_RELU && v->fc[1].act == GE_RELU && tv->fc[1].act == GE_RELU &&
             relu_grad_epi(v, 0) == GE_MUL_RELU_GRAD && relu_grad_epi(v, 1) == GE_MUL_RELU_GRAD && !v->drop_counter;
  if (mlp) {
    nm.h = nh;
    nm.x0 = v->ws[0].fcin[1]; nm.x0t = tv->ws[0].fcin[1]; nm.ld0 = v->fc[1].n_in + 1; nm.n0 = v->fc[1].n_in;
    nm.W1 = v->params + v->fc[1].w_off; nm.W1t = tv->params + tv->fc[1].w_off;
    nm.h1_out = v->ws[0].fcin[2]; nm.ld1 = v->fc[2].n_in + 1; nm.dz0 = v->ws[0].dz[0];
    mlp = naf_mlp_supported(nm);
  }
  int vrep = t1;
  for (int l = 0; l < Lh - (mlp ? 1 : 0); ++l) vrep = G.gemm(fc_fwd_args(v, v->ws[0], l, B), {vrep});
  int vout, tvout, muout, lvout;
  if (fused) {
    f->step_bumped = nh.step_bump != nullptr;
    int tvrep = t1;
    for (int l = 0; l < Lh - (mlp ? 1 : 0); ++l) tvrep = G.gemm(fc_fwd_args(tv, tv->ws[0], l, B), {tvrep});
    vout = vrep; tvout = tvrep; muout = lvout = vrep;
  } else {
    vout = G.gemm(fc_fwd_args(v, v->ws[0], Lh, B), {vrep});
    tvout = chain(tv, 0, t1);
    muout = share ? chain(mu, 0, vrep) : chain(mu, 0, t1);
    lvout = share ? chain(lv, 0, vrep) : chain(lv, 0, t1);
  }
  if (v->drop_counter) {     // --use-dropout: count this training-mode forward of every network with a hidden stack
    G.fn([=] { return bump_dropout(v); }, {vout});
    G.fn([=] { return bump_dropout(tv); }, {tvout});
    if (!share) { G.fn([=] { return bump_dropout(mu); }, {muout}); G.fn([=] { return bump_dropout(lv); }, {lvout}); }
  }
  // ---- NAF head: L, advantage, TD loss and the gradients of the three head outputs
  const int head = mlp ? G.fn([=] { return launch_naf_mlp(ctx, nm); }, {vout, tvout})
                 : fused ? G.fn([=] { return launch_naf_heads(ctx, nh); }, {vout, tvout})
                         : G.fn([=] { return naf_head(f, b, true); }, {vout, tvout, muout, lvout});

  // ---- backward
  if (!share) {
    const int dv = add_fc_backward(G, v, v->ws[0], B, (int)v->fc.size() - 1, head, sqg);
    const int dm = add_fc_backward(G, mu, mu->ws[0], B, (int)mu->fc.size() - 1, head, sqg);
    const int dl = add_fc_backward(G, lv, lv->ws[0], B, (int)lv->fc.size() - 1, head, sqg);
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
      G.gemm(sqg(mk_gemm(x, 1, rep + 1, h.dz, h.L->n_out, 1, h.n->grads + h.L->w_off, h.L->n_out, rep + 1, h.L->n_out, B, GE_NONE)), {head});
      if (drep && !fused) {
        GemmArgs g = mk_gemm(h.dz, h.L->n_out, 1, h.n->params + h.L->w_off, 1, h.L->n_out, drep, ldd, B, rep, h.L->n_out,
                             k == 2 ? final_epi : GE_NONE, k == 2 ? Y : nullptr, ldy);
        g.accumulate = k > 0;
        dep = G.gemm(g, {dep});                  // accumulation order value -> mu -> l_values is fixed
      }
    }
    if (mlp) G.gemm(sqg(fc_dw_args(v, v->ws[0], 1, B, v->ws[0].dz[1])), {head});      // (dz[1] and dz[0] came out of naf_mlp_kernel)
    const int dv = add_fc_backward(G, v, v->ws[0], B, mlp ? 0 : Lh - 1, dep, sqg);
    if (v->spec.pixel) {     // conv3's and conv2's dW + dX as one launch each, like the DDPG step (nets_backward_conv)
      cpp_net* bn[1] = {v};
      G.fn([=] { return nets_backward_conv(ctx, bn, 1, B, s1, dt, w1); }, {dv});
    }
  }
  DwPendingGuard pending(ctx);      // (a failure below drops what was queued)
  RC(G.run(ctx));
  return flush_dw_reduce(ctx);
}

// bump: the replay sampler's counter, advanced by this launch; next (+ next_B, next_C, elems): the minibatch whose sample pass has
// already run -- its whitening tables are finished by this launch's extra grid row (as in the DDPG step, rt_ddpg.cpp: apply)
static int naf_apply(cpp_naf* f, float grad_scale, bool unless_nonfinite = false, uint64_t* bump = nullptr,
                     const cpp_batch* next = nullptr, int next_B = 0, int next_C = 0, long elems = 0, bool folded = false,
                     bool tables_done = false) {
  OptSegs s; memset(&s, 0, sizeof(s));
  if (unless_nonfinite) s.skip_if = f->nonfinite;
  s.bump = bump;
  f->value->wimg_key = nullptr; f->mu->wimg_key = nullptr;      // (the parameters change)
  const bool with_targets = f->targets_in_apply && !next && !unless_nonfinite;      // (the outer step's last launch: naf_step_body)
  f->targets_in_apply = false;
  if (next && next_C > 0 && !tables_done) {
    s.st_part = next->part; s.st_white = next->white; s.st_nparts = next_B; s.st_jobs = 2 * next_C; s.st_C = next_C;
    s.st_count = (double)next_B * (double)(elems / next_C); s.st_eps = 1e-6; s.st_wmax = f->ctx->white_max_dev;
  }
  s.nseg = 3; s.kind = f->hp.optimiser; s.momentum = f->hp.momentum; s.beta1 = f->hp.beta1; s.beta2 = f->hp.beta2;
  s.epsilon = f->hp.epsilon; s.step = f->opt_step;
  cpp_net* nets[3] = {f->value, f->mu, f->lv};
  long off = 0;
  for (int k = 0; k < 3; ++k) {
    s.p[k] = nets[k]->params; s.g[k] = f->gradbuf + off; s.m[k] = f->m + off; s.v[k] = f->v + off;
    s.n[k] = nets[k]->nparams; s.lr[k] = f->hp.learning_rate; s.group[k] = 0;      // ONE list, one global norm
    off += nets[k]->nparams;
  }
  if (with_targets) {      // the target value network's soft update (naf_cartpole.py:373) and the route's publish leave with this launch
    s.tgt[0] = f->tvalue->params; s.tgt_coeff = f->hp.target_update_rate;
    f->tvalue->wimg_key = nullptr;
    if (f->ctx->route_pin_dev) { s.pub_wmax = f->ctx->white_max_dev; s.pub_tag = f->ctx->route_tag_dev; s.pub_pin = f->ctx->route_pin_dev; }
    f->targets_applied = true;
  }
  const bool bumped = f->step_bumped; const int sq_cnt = f->sq_cnt;
  f->step_bumped = false; f->sq_cnt = 0;
  // (unless_nonfinite: a skipped update must not count as an optimiser step -- Adam's bias correction reads the count; the heads
  // kernel's bump was undone below by never happening: naf_compute_gradients leaves it to this launch when the flag can stand the update down)
  if (!bumped) RC(launch_counter_add(f->ctx, f->opt_step, 1, unless_nonfinite ? f->nonfinite : nullptr));
  if (folded && sq_cnt > 0 && grad_scale == 1.0f) { s.sq = f->ctx->sq_part; s.sq_begin[0] = 0; s.sq_count[0] = sq_cnt; }
  else RC(launch_sumsq(f->ctx, s, grad_scale, f->norm_part, NORM_PARTS));
  // conv1's operand images of the next minibatch ride along (as rt_ddpg.cpp's apply; shared trunk: the value network's conv1 on
  // state_1, the target value network's on state_2; SGD or Momentum -- Adam's update is not restated in the rider)
  cpp_net* inets[2] = {f->value, f->tvalue};
  const ConvL* L0 = (f->share && f->value->spec.pixel) ? &f->value->conv[0] : nullptr;
  bool img = next && next_C > 0 && L0 && !unless_nonfinite && !f->value->spec.use_batch_norm && next_B >= 2 &&
             (s.kind == OPT_SGD || s.kind == OPT_MOMENTUM) && conv_rs16_ok(f->ctx, L0->Cin, L0->H, L0->W, kConvOut) &&
             L0->w_off == 0 && L0->b_off == (long)L0->ks * L0->ks * L0->Cin * kConvOut;
  if (img) {
    s.img_n = 2; s.img_cin = L0->Cin;
    s.img_skip[0] = L0->b_off + kConvOut;
    for (int j = 0; j < 2; ++j) {
      cpp_net* n = inets[j];
      const ConvL& L = n->conv[0];
      s.img[j].w = n->params + L.w_off; s.img[j].bias = n->params + L.b_off;
      s.img[j].gw = j == 0 ? s.g[0] + L.w_off : nullptr; s.img[j].gb = j == 0 ? s.g[0] + L.b_off : nullptr;
      s.img[j].mw = j == 0 ? s.m[0] + L.w_off : nullptr; s.img[j].mb = j == 0 ? s.m[0] + L.b_off : nullptr;
      s.img[j].rec = reinterpret_cast<unsigned char*>(n->wimg); s.img[j].seg = 0; s.img[j].col = j; s.img[j].nout = kConvOut;
      s.img[j].white = tables_done ? next->white + (long)j * 2 * next_C : nullptr;
    }
  }
  RC(launch_opt_apply(f->ctx, s, grad_scale, f->hp.gradient_clip, f->norm_part, NORM_PARTS, f->stats + 1));
  if (img) for (int j = 0; j < 2; ++j) inets[j]->wimg_key = next->white + (long)j * 2 * next_C;
  return CPP_OK;
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
  HIP_CHECK(ctx_sync_stream(ctx));
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
  f->tvalue->wimg_key = nullptr;
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
  HIP_CHECK(ctx_sync_stream(ctx));
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

// (as rt_ddpg.cpp's route_check: the context may have moved conv1 to the other kernel family -- every cached graph is rebuilt)
static void naf_route_check(cpp_naf* f) {
  ctx_route_update(f->ctx);
  if (f->epoch == f->ctx->kernel_epoch) return;
  f->epoch = f->ctx->kernel_epoch;
  f->graph_ok = false; f->hgraph_ok = false; f->rgraph_ok = false; f->dgraph_ok = false; f->agraph_ok = false;
  f->value->wimg_key = nullptr; f->tvalue->wimg_key = nullptr; f->mu->wimg_key = nullptr;
}

// The inner step naf_cartpole.py:367-373.  As in the DDPG step (rt_ddpg.cpp: step_body) the sample pass of minibatch i + 1 depends on
// nothing minibatch i computes: it rides in the launch of i's conv1 dW (or of its dW reductions), keyed by the sampler's counter + 1
// -- the counter itself moves in i's optimiser launch, which also finishes the whitening tables of i + 1.  CPP_RIDE_GATHER=0: in sequence.
// dp / comm: as rt_ddpg.cpp's step_body -- the gradient all-reduce sits between a minibatch's gradients and its update, inside the graph
static int naf_step_body(cpp_naf* f, cpp_replay* r, int B, int n_batches, const int32_t* rows_dev, uint64_t seed, bool dp = false, cpp_comm* comm = nullptr) {
  const int C = f->value->spec.pixel ? f->value->spec.C : 0;
  cpp_ctx* ctx = f->ctx;
  const bool direct = direct_replay_ok(f->value, r, B);
  static const bool no_ride = cpp_switch_off("CPP_RIDE_GATHER");
  const bool ride_ok = !no_ride && C > 0 && !f->value->spec.use_batch_norm && (r->store_dtype == CPP_F16 || r->store_dtype == CPP_U8);
  RC(replay_sample_device(r, B, rows_dev, seed, rows_dev ? nullptr : r->counter, C, f->step_batch, direct));
  for (int i = 0; i < n_batches; ++i) {
    GatherArgs ga; int Cg = 0;
    const bool more = i + 1 < n_batches;
    if (more && ride_ok) {
      ga = replay_gather_args(r, B, rows_dev ? rows_dev + (size_t)(i + 1) * B : nullptr, seed, rows_dev ? nullptr : r->counter, C,
                              f->step_batch, direct, &Cg);
      ga.counter_add = 1;
      static const bool no_dwride = cpp_switch_off("CPP_RIDE_DW");
      ctx->ride_at_dw = direct && !no_dwride;
      if (direct) { ga.out_slot[0] = f->step_batch->slot_alt[0]; ga.out_slot[1] = f->step_batch->slot_alt[1]; }
      ctx->ride = &ga; ctx->ride_done = false; ctx->ride_dtype = r->store_dtype;
    }
    // (its statistics are finished in the dW reductions' launch when it leaves with conv1's dW: rt_ddpg.cpp, step_body)
    static const bool no_stats_ride = cpp_switch_off("CPP_RIDE_STATS");
    StatsRide sr;
    if (ctx->ride && ctx->ride_at_dw && Cg > 0 && !no_stats_ride) {
      sr.part = f->step_batch->part; sr.white = f->step_batch->white; sr.nparts = B; sr.jobs = 2 * Cg; sr.C = Cg;
      sr.count = (double)B * (double)(r->elems / Cg); sr.eps = 1e-6; sr.wmax = ctx->white_max_dev;
      ctx->st_ride = &sr; ctx->st_ride_done = false;
    }
    const int rc = naf_compute_gradients(f, f->step_batch, true);
    const bool rode = ctx->ride != nullptr && ctx->ride_done;
    const bool tables_done = ctx->st_ride != nullptr && ctx->st_ride_done && rode;
    ctx->ride = nullptr; ctx->st_ride = nullptr;
    if (rode && direct) { std::swap(f->step_batch->slot[0], f->step_batch->slot_alt[0]); std::swap(f->step_batch->slot[1], f->step_batch->slot_alt[1]); }
    RC(rc);
    const bool stats_ride = rode && Cg > 0;
    if (dp && comm) {
      prof_begin(ctx);
      NCCL_CHECK(ncclAllReduce(f->gradbuf, f->gradbuf, (size_t)(f->nV + f->nM + f->nL), ncclFloat, ncclSum, comm->comm, ctx->stream));
      prof_end(ctx, K_ALLREDUCE);
    }
    static const bool no_tgt_ride = cpp_switch_off("CPP_RIDE_TARGETS");
    f->targets_applied = false;
    f->targets_in_apply = !more && !dp && !no_tgt_ride;      // (the last minibatch of an outer step: the target update rides in its optimiser launch)
    RC(naf_apply(f, (dp && comm) ? 1.0f / (float)comm->world : 1.0f, false, rows_dev ? nullptr : r->counter, stats_ride ? f->step_batch : nullptr, B, Cg, r->elems, !dp, tables_done));
    if (more) {
      if (stats_ride) { f->step_batch->B = B; f->step_batch->dtype = CPP_F16; f->step_batch->stats_C = Cg; }     // (replay_sample_finish's bookkeeping)
      else if (rode) RC(replay_sample_finish(r, B, Cg, C, f->step_batch));
      else RC(replay_sample_device(r, B, rows_dev ? rows_dev + (size_t)(i + 1) * B : nullptr, seed, rows_dev ? nullptr : r->counter, C,
                                   f->step_batch, direct));
    }
  }
  if (f->targets_applied) { f->targets_applied = false; return CPP_OK; }      // (the target update and the route's publish left with the optimiser's launch)
  ctx->route_rider = true;                          // (the largest whitening scale of this step rides to the host in that launch)
  return cpp_naf_update_targets(f);      // (the largest whitening scale of this step, for the next call's choice of conv1 kernels)
}

extern "C" int cpp_naf_train_step(cpp_naf* f, cpp_replay* r, int B, int n_batches, const int32_t* idxs, uint64_t seed) {
  if (f) naf_route_check(f);
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
  if (!f->graph_ok || f->g_B != B || f->g_nb != n_batches || f->g_seed != seed || f->g_replay_uid != r->uid) {
    if (f->gexec) { (void)hipGraphExecDestroy(f->gexec); f->gexec = nullptr; }
    if (f->graph) { (void)hipGraphDestroy(f->graph); f->graph = nullptr; }
    f->graph_ok = false;
    RC(naf_step_body(f, r, B, n_batches, nullptr, seed));
    HIP_CHECK(ctx_sync_stream(ctx));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = naf_step_body(f, r, B, n_batches, nullptr, seed);
    hipError_t e = hipStreamEndCapture(ctx->stream, &f->graph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&f->gexec, f->graph, nullptr, nullptr, 0));
    f->graph_ok = true; f->g_B = B; f->g_nb = n_batches; f->g_seed = seed; f->g_replay_uid = r->uid;
    return CPP_OK;
  }
  HIP_CHECK(hipGraphLaunch(f->gexec, ctx->stream));
  return CPP_OK;
}

// naf_cartpole.py:367-371 for ONE minibatch whose rows the HOST drew: `batch = replay_memory.batch(B); loss = naf.train(batch)` without a
// gathered copy of the minibatch crossing PCIe or HBM twice -- the sample pass reads the replay store through those rows.  Like
// cpp_naf_train: gradients, then the loss and the check_numerics flag come back (one stream sync: the reference's train() returns the
// loss), then the optimiser -- which does not run when the flag is set.  The gradient half is one hipGraph per (B, replay).
// sticky: the check_numerics flag is NOT cleared (cpp_naf_train_rows_async: the host learns of a non-finite minibatch up to two calls
// later; until it has, every later optimiser launch must stand down as the first one did -- the reference's check_numerics stops
// training before any further train op, naf_cartpole.py:242-245,265)
static int naf_rows_body(cpp_naf* f, cpp_replay* r, int B, bool fold = false, bool sticky = false) {
  const int C = f->value->spec.pixel ? f->value->spec.C : 0;
  if (!sticky) HIP_CHECK(hipMemsetAsync(f->nonfinite, 0, sizeof(int), f->ctx->stream));
  RC(replay_sample_device(r, B, r->rows_in, 0, nullptr, C, f->step_batch, direct_replay_ok(f->value, r, B)));
  RC(naf_compute_gradients(f, f->step_batch, fold, !sticky));
  return ctx_route_publish(f->ctx);
}
extern "C" int cpp_naf_train_rows(cpp_naf* f, cpp_replay* r, int B, const int32_t* idxs, float* loss) {
  if (f) naf_route_check(f);
  ARG_CHECK(f && r && idxs, "cpp_naf_train_rows: NULL argument");
  ARG_CHECK(B >= 1 && B <= f->maxB, "cpp_naf_train_rows: batch %d outside [1,%d]", B, f->maxB);
  ARG_CHECK(r->elems == f->value->state_elems && r->A == f->A, "cpp_naf_train_rows: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_naf_train_rows: replay memory is empty"); return CPP_ERR_STATE; }
  cpp_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!f->step_batch) RC(cpp_batch_create(ctx, f->maxB, r->elems, r->A, &f->step_batch));
  RC(replay_stage_rows(r, idxs, B, "cpp_naf_train_rows"));
  if (ctx->prof) {
    RC(naf_rows_body(f, r, B));
  } else if (!f->rgraph_ok || f->rg_B != B || f->rg_replay_uid != r->uid) {
    if (f->rgexec) { (void)hipGraphExecDestroy(f->rgexec); f->rgexec = nullptr; }
    if (f->rgraph) { (void)hipGraphDestroy(f->rgraph); f->rgraph = nullptr; }
    f->rgraph_ok = false;
    RC(naf_rows_body(f, r, B));                      // eager pass: sets kernel attributes, is this call's work
    HIP_CHECK(ctx_sync_stream(ctx));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = naf_rows_body(f, r, B);
    const hipError_t e = hipStreamEndCapture(ctx->stream, &f->rgraph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&f->rgexec, f->rgraph, nullptr, nullptr, 0));
    f->rgraph_ok = true; f->rg_B = B; f->rg_replay_uid = r->uid;
  } else {
    HIP_CHECK(hipGraphLaunch(f->rgexec, ctx->stream));
  }
  int bad = 0; float l = 0.f;
  HIP_CHECK(hipMemcpyAsync(&bad, f->nonfinite, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipMemcpyAsync(&l, f->stats, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(ctx_sync_stream(ctx));
  if (loss) *loss = l;
  if (bad) { cpp_set_error("check_numerics: l_values / L / loss is not finite (naf_cartpole.py:242-245)"); return CPP_ERR_NUMERIC; }
  return naf_apply(f, 1.0f);
}

// The same minibatch without the host in the loop: gradients AND the optimiser in one hipGraph -- the optimiser kernel itself stands
// down when the check_numerics flag is set -- and the (loss, flag) pair copied into a pinned slot behind it.  The call returns a
// ticket at once; cpp_naf_loss_wait(ticket) is the sync the reference's `loss = naf.train(batch)` implies, taken when somebody
// reads the loss (the agents only log its mean) -- so a loop of train calls keeps the GPU fed like cpp_naf_train_step does.
// At most CPP_NAF_TICKETS results are outstanding: a slot is reused CPP_NAF_TICKETS calls later.
static int naf_rows_apply_body(cpp_naf* f, cpp_replay* r, int B) {
  RC(naf_rows_body(f, r, B, true, true));  // (gradients and optimiser in ONE captured body: the fold's host-side state is consistent)
  return naf_apply(f, 1.0f, true, nullptr, nullptr, 0, 0, 0, true);
}
extern "C" int cpp_naf_train_rows_async(cpp_naf* f, cpp_replay* r, int B, const int32_t* idxs, uint64_t* ticket) {
  if (f) naf_route_check(f);
  ARG_CHECK(f && r && idxs && ticket, "cpp_naf_train_rows_async: NULL argument");
  ARG_CHECK(B >= 1 && B <= f->maxB, "cpp_naf_train_rows_async: batch %d outside [1,%d]", B, f->maxB);
  ARG_CHECK(r->elems == f->value->state_elems && r->A == f->A, "cpp_naf_train_rows_async: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_naf_train_rows_async: replay memory is empty"); return CPP_ERR_STATE; }
  cpp_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!f->step_batch) RC(cpp_batch_create(ctx, f->maxB, r->elems, r->A, &f->step_batch));
  if (!f->res_pin) {
    HIP_CHECK(hipHostMalloc((void**)&f->res_pin, CPP_NAF_TICKETS * 2 * sizeof(float), hipHostMallocDefault));
    for (hipEvent_t& e : f->res_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  RC(replay_stage_rows(r, idxs, B, "cpp_naf_train_rows_async"));
  if (ctx->prof) {
    RC(naf_rows_apply_body(f, r, B));
  } else if (!f->agraph_ok || f->ag_B != B || f->ag_replay_uid != r->uid) {
    if (f->agexec) { (void)hipGraphExecDestroy(f->agexec); f->agexec = nullptr; }
    if (f->agraph) { (void)hipGraphDestroy(f->agraph); f->agraph = nullptr; }
    f->agraph_ok = false;
    RC(naf_rows_apply_body(f, r, B));                // eager pass: sets kernel attributes, is this call's work
    HIP_CHECK(ctx_sync_stream(ctx));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = naf_rows_apply_body(f, r, B);
    const hipError_t e = hipStreamEndCapture(ctx->stream, &f->agraph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&f->agexec, f->agraph, nullptr, nullptr, 0));
    f->agraph_ok = true; f->ag_B = B; f->ag_replay_uid = r->uid;
  } else {
    HIP_CHECK(hipGraphLaunch(f->agexec, ctx->stream));
  }
  const uint64_t t = f->next_ticket++;
  float* slot = f->res_pin + (t % CPP_NAF_TICKETS) * 2;
  HIP_CHECK(hipMemcpyAsync(slot, f->stats, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipMemcpyAsync(slot + 1, f->nonfinite, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipEventRecord(f->res_ev[t % CPP_NAF_TICKETS], ctx->stream));
  *ticket = t;
  return CPP_OK;
}
extern "C" int cpp_naf_loss_wait(cpp_naf* f, uint64_t ticket, float* loss) {
  ARG_CHECK(f && f->res_pin, "cpp_naf_loss_wait: no asynchronous train call has been made");
  ARG_CHECK(ticket < f->next_ticket && ticket + CPP_NAF_TICKETS >= f->next_ticket, "cpp_naf_loss_wait: ticket %llu is not one of the last %d",
            (unsigned long long)ticket, CPP_NAF_TICKETS);
  HIP_CHECK(hipEventSynchronize(f->res_ev[ticket % CPP_NAF_TICKETS]));
  const float* slot = f->res_pin + (ticket % CPP_NAF_TICKETS) * 2;
  if (loss) *loss = slot[0];
  int bad; memcpy(&bad, slot + 1, sizeof(int));
  if (bad) { cpp_set_error("check_numerics: l_values / L / loss is not finite (naf_cartpole.py:242-245)"); return CPP_ERR_NUMERIC; }
  return CPP_OK;
}

// ---- data-parallel learners: the halves of one minibatch of naf_cartpole.py:367-371 -------------------------
static int naf_half_body(cpp_naf* f, cpp_replay* r, int B, uint64_t seed) {
  const int C = f->value->spec.pixel ? f->value->spec.C : 0;
  bool bumped = false;       // (the statistics kernel advances the sampler's counter when there is one: replay_sample_finish)
  RC(replay_sample_device(r, B, nullptr, seed, r->counter, C, f->step_batch, direct_replay_ok(f->value, r, B), r->counter, &bumped));
  if (!bumped) RC(launch_counter_add(f->ctx, r->counter, 1));
  RC(naf_compute_gradients(f, f->step_batch));
  return ctx_route_publish(f->ctx);
}

static int naf_half_checks(cpp_naf* f, cpp_replay* r, int B, const char* who) {
  ARG_CHECK(f && r, "%s: NULL argument", who);
  ARG_CHECK(B >= 1 && B <= f->maxB, "%s: batch %d outside [1,%d]", who, B, f->maxB);
  ARG_CHECK(r->elems == f->value->state_elems && r->A == f->A, "%s: replay shape does not match the networks", who);
  if (r->size <= 0) { cpp_set_error("%s: replay memory is empty", who); return CPP_ERR_STATE; }
  return CPP_OK;
}

// sample B rows on the device (Philox; the counter advances by one) and leave the gradients of the three networks in the flat
// buffer [value | mu | l_values]; hipGraph-captured after the first call per (B, seed, replay)
extern "C" int cpp_naf_sample_and_compute(cpp_naf* f, cpp_replay* r, int B, uint64_t seed) {
  if (f) naf_route_check(f);
  RC(naf_half_checks(f, r, B, "cpp_naf_sample_and_compute"));
  cpp_ctx* ctx = f->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!f->step_batch) RC(cpp_batch_create(ctx, f->maxB, r->elems, r->A, &f->step_batch));
  if (ctx->prof) return naf_half_body(f, r, B, seed);
  if (!f->hgraph_ok || f->h_B != B || f->h_seed != seed || f->h_replay_uid != r->uid) {
    if (f->hexec) { (void)hipGraphExecDestroy(f->hexec); f->hexec = nullptr; }
    if (f->hgraph) { (void)hipGraphDestroy(f->hgraph); f->hgraph = nullptr; }
    f->hgraph_ok = false;
    RC(naf_half_body(f, r, B, seed));                // eager pass: sets kernel attributes, is this call's work
    HIP_CHECK(ctx_sync_stream(ctx));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    const int rc = naf_half_body(f, r, B, seed);
    const hipError_t e = hipStreamEndCapture(ctx->stream, &f->hgraph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&f->hexec, f->hgraph, nullptr, nullptr, 0));
    f->hgraph_ok = true; f->h_B = B; f->h_seed = seed; f->h_replay_uid = r->uid;
    return CPP_OK;
  }
  HIP_CHECK(hipGraphLaunch(f->hexec, ctx->stream));
  return CPP_OK;
}

extern "C" int cpp_naf_allreduce_grads(cpp_naf* f, cpp_comm* c) {
  ARG_CHECK(f && c, "cpp_naf_allreduce_grads: NULL argument");
  ARG_CHECK(c->ctx == f->ctx, "cpp_naf_allreduce_grads: communicator and networks live on different contexts");
  HIP_CHECK(hipSetDevice(f->ctx->device));
  NCCL_CHECK(ncclAllReduce(f->gradbuf, f->gradbuf, (size_t)(f->nV + f->nM + f->nL), ncclFloat, ncclSum, c->comm, f->ctx->stream));
  return CPP_OK;
}

// periodic mode: parameters, the target and the optimiser slots (Momentum accumulators / Adam moments) of the replicas meet at
// their mean
extern "C" int cpp_naf_average_params(cpp_naf* f, cpp_comm* c) {
  ARG_CHECK(f && c, "cpp_naf_average_params: NULL argument");
  ARG_CHECK(c->ctx == f->ctx, "cpp_naf_average_params: communicator and networks live on different contexts");
  HIP_CHECK(hipSetDevice(f->ctx->device));
  const size_t nall = (size_t)(f->nV + f->nM + f->nL);
  cpp_net* nets[4] = {f->value, f->mu, f->lv, f->tvalue};
  NCCL_CHECK(ncclGroupStart());
  for (cpp_net* n : nets)
    NCCL_CHECK(ncclAllReduce(n->params, n->params, (size_t)n->nparams, ncclFloat, ncclAvg, c->comm, f->ctx->stream));
  if (f->hp.optimiser != CPP_OPT_SGD) NCCL_CHECK(ncclAllReduce(f->m, f->m, nall, ncclFloat, ncclAvg, c->comm, f->ctx->stream));
  if (f->hp.optimiser == CPP_OPT_ADAM) NCCL_CHECK(ncclAllReduce(f->v, f->v, nall, ncclFloat, ncclAvg, c->comm, f->ctx->stream));
  NCCL_CHECK(ncclGroupEnd());
  f->dp_local = 0;
  return CPP_OK;
}

// the inner step naf_cartpole.py:367-373 for N synchronous learners (this rank's part); see cpp_ddpg_dp_train_step
extern "C" int cpp_naf_dp_status(const cpp_naf* f, int* mode, char* reason, int cap) {      // (see cpp_ddpg_dp_status)
  ARG_CHECK(f && mode, "cpp_naf_dp_status: NULL argument");
  *mode = f->dgraph_refused ? 2 : (f->dgraph_ok ? 1 : 0);
  if (reason && cap > 0) snprintf(reason, (size_t)cap, "%s", f->dg_reason);
  return CPP_OK;
}

extern "C" int cpp_naf_dp_train_step(cpp_naf* f, cpp_replay* r, cpp_comm* c, int B, int n_batches, uint64_t seed, int sync_every) {
  if (f) naf_route_check(f);
  RC(naf_half_checks(f, r, B, "cpp_naf_dp_train_step"));
  ARG_CHECK(n_batches >= 1 && sync_every >= 1, "cpp_naf_dp_train_step: n_batches %d, sync_every %d", n_batches, sync_every);
  ARG_CHECK(!c || c->ctx == f->ctx, "cpp_naf_dp_train_step: communicator and networks live on different contexts");
  const float inv = c ? 1.0f / (float)c->world : 1.0f;
  static const bool no_dp_graph = cpp_switch_off("CPP_DP_GRAPH");
  if (sync_every == 1 && !no_dp_graph) {             // ONE hipGraph per outer step, the all-reduce inside (see cpp_ddpg_dp_train_step)
    cpp_ctx* ctx = f->ctx;
    HIP_CHECK(hipSetDevice(ctx->device));
    if (!f->step_batch) RC(cpp_batch_create(ctx, f->maxB, r->elems, r->A, &f->step_batch));
    if (ctx->prof) return naf_step_body(f, r, B, n_batches, nullptr, seed, true, c);
    if (f->dgraph_refused) return naf_step_body(f, r, B, n_batches, nullptr, seed, true, c);
    if (!f->dgraph_ok || f->dg_B != B || f->dg_nb != n_batches || f->dg_seed != seed || f->dg_replay_uid != r->uid || f->dg_comm_uid != (c ? c->uid : 0)) {
      if (f->dgexec) { (void)hipGraphExecDestroy(f->dgexec); f->dgexec = nullptr; }
      if (f->dgraph) { (void)hipGraphDestroy(f->dgraph); f->dgraph = nullptr; }
      f->dgraph_ok = false;
      RC(naf_step_body(f, r, B, n_batches, nullptr, seed, true, c));
      HIP_CHECK(ctx_sync_stream(ctx));
      if (f->dgraph_refused) return CPP_OK;
      // (as cpp_ddpg_dp_train_step: a runtime / RCCL that refuses the capture leaves the same sequence as plain stream launches)
      HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
      const int rc = naf_step_body(f, r, B, n_batches, nullptr, seed, true, c);
      const hipError_t e = hipStreamEndCapture(ctx->stream, &f->dgraph);
      hipError_t ei = hipSuccess;
      if (!rc && e == hipSuccess) ei = hipGraphInstantiate(&f->dgexec, f->dgraph, nullptr, nullptr, 0);
      if (rc || e != hipSuccess || ei != hipSuccess) {
        (void)hipGetLastError();
        if (f->dgexec) { (void)hipGraphExecDestroy(f->dgexec); f->dgexec = nullptr; }
        if (f->dgraph) { (void)hipGraphDestroy(f->dgraph); f->dgraph = nullptr; }
        f->dgraph_refused = true;
        snprintf(f->dg_reason, sizeof(f->dg_reason), "%s", rc ? cpp_last_error() : hipGetErrorString(e != hipSuccess ? e : ei));
        fprintf(stderr, "cartpolepp: the data-parallel NAF step could not be captured as a hipGraph (%s); running it as stream launches\n", f->dg_reason);
        return CPP_OK;
      }
      f->dgraph_ok = true; f->dg_B = B; f->dg_nb = n_batches; f->dg_seed = seed; f->dg_replay_uid = r->uid; f->dg_comm_uid = c ? c->uid : 0;
      return CPP_OK;
    }
    HIP_CHECK(hipGraphLaunch(f->dgexec, ctx->stream));
    return CPP_OK;
  }
  for (int i = 0; i < n_batches; ++i) {
    RC(cpp_naf_sample_and_compute(f, r, B, seed));
    if (sync_every > 1) {
      RC(naf_apply(f, 1.0f));
      if (++f->dp_local >= (uint64_t)sync_every && c) RC(cpp_naf_average_params(f, c));
    } else {
      if (c) RC(cpp_naf_allreduce_grads(f, c));
      RC(naf_apply(f, inv));
    }
  }
  return cpp_naf_update_targets(f);
}

extern "C" int cpp_naf_last_stats(cpp_naf* f, float out[3]) {
  ARG_CHECK(f && out, "cpp_naf_last_stats: NULL argument");
  int bad = 0;
  HIP_CHECK(hipMemcpyAsync(out, f->stats, 2 * sizeof(float), hipMemcpyDeviceToHost, f->ctx->stream));
  HIP_CHECK(hipMemcpyAsync(&bad, f->nonfinite, sizeof(int), hipMemcpyDeviceToHost, f->ctx->stream));
  HIP_CHECK(ctx_sync_stream(f->ctx));
  out[2] = (float)bad;
  return CPP_OK;
}


// optimiser slots for checkpoints (util.py:88-90: tf.train.Saver saves the Momentum / Adam slot variables too)
extern "C" int64_t cpp_naf_opt_state_size(const cpp_naf* f) { return f ? (int64_t)(f->nV + f->nM + f->nL) : -1; }
extern "C" int cpp_naf_get_opt_state(cpp_naf* f, float* m, float* v, int64_t n, uint64_t* step) {
  ARG_CHECK(f, "cpp_naf_get_opt_state: NULL argument");
  ARG_CHECK(n == f->nV + f->nM + f->nL, "cpp_naf_get_opt_state: asked %ld values, the optimiser has %ld", (long)n, f->nV + f->nM + f->nL);
  hipStream_t st = f->ctx->stream;
  HIP_CHECK(hipSetDevice(f->ctx->device));
  if (m) HIP_CHECK(hipMemcpyAsync(m, f->m, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  if (v) HIP_CHECK(hipMemcpyAsync(v, f->v, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  if (step) HIP_CHECK(hipMemcpyAsync(step, f->opt_step, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}
extern "C" int cpp_naf_set_opt_state(cpp_naf* f, const float* m, const float* v, int64_t n, uint64_t step) {
  ARG_CHECK(f, "cpp_naf_set_opt_state: NULL argument");
  ARG_CHECK(n == f->nV + f->nM + f->nL, "cpp_naf_set_opt_state: got %ld values, the optimiser has %ld", (long)n, f->nV + f->nM + f->nL);
  hipStream_t st = f->ctx->stream;
  HIP_CHECK(hipSetDevice(f->ctx->device));
  if (m) HIP_CHECK(hipMemcpyAsync(f->m, m, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
  if (v) HIP_CHECK(hipMemcpyAsync(f->v, v, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemcpyAsync(f->opt_step, &step, sizeof(uint64_t), hipMemcpyHostToDevice, st));
  // a restored checkpoint is a fresh start: the sticky check_numerics flag of cpp_naf_train_rows_async goes with the state it condemned
  HIP_CHECK(hipMemsetAsync(f->nonfinite, 0, sizeof(int), st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}
// After CPP_ERR_NUMERIC on the asynchronous path every later update of this trainer stands down (the reference's check_numerics ends the
// run, naf_cartpole.py:242-245,265).  A caller that has repaired the parameters (cpp_net_set_params of a checkpoint) says so here.
extern "C" int cpp_naf_clear_numeric_error(cpp_naf* f) {
  ARG_CHECK(f, "cpp_naf_clear_numeric_error: NULL argument");
  HIP_CHECK(hipSetDevice(f->ctx->device));
  HIP_CHECK(hipMemsetAsync(f->nonfinite, 0, sizeof(int), f->ctx->stream));
  return CPP_OK;
}
