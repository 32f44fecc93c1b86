// DDPG train ops, the fused inner step and its hipGraph, the data-parallel half steps (cpp_ddpg_*)
#include "rt_internal.h"

// ---------------------------------------------------------------------------------------------
// DDPG
// ---------------------------------------------------------------------------------------------

struct cpp_ddpg {
  cpp_ctx* ctx; cpp_net *actor, *critic, *tactor, *tcritic; cpp_ddpg_hyper hp;
  int maxB; long nA, nC;
  float* gradbuf; float *dq_da, *td, *dq, *loss_norms /* [0] loss [1] actor norm [2] critic norm */, *ones;
  double* norm_part;
  double* heads_part;                              // fused heads kernel: per-workgroup partial sums of td^2
  int heads_grid, heads_B;                         // ... of the last graph built by compute_gradients (0: GEMM levels + td_kernel)
  int loss_parts, loss_B;                          // how cpp_ddpg_last_stats finds the loss of the last call: partials to add, or loss_norms[0]
  // graph replay of the full inner step
  hipGraph_t graph; hipGraphExec_t gexec; bool graph_ok; int g_B, g_nb; uint64_t g_seed, g_replay_uid;   // (the sampler's range is read from the replay's device size word: one graph survives growth)
  cpp_batch* step_batch;
  // graph replay of ONE minibatch on host-drawn rows, no target update (cpp_ddpg_train_rows: the reference's literal loop)
  hipGraph_t rgraph; hipGraphExec_t rgexec; bool rgraph_ok; int rg_B; uint64_t rg_replay_uid;
  uint64_t epoch;            // cpp_ctx::kernel_epoch the cached graphs were captured under (route_check)
  bool publish_in_apply;     // the next apply() closes a training call: its launch publishes the call's whitening scale
  bool targets_in_apply, targets_applied;      // ... and an outer step: its launch carries both target updates (step_body)
  hipGraph_t dgraph; hipGraphExec_t dgexec; bool dgraph_ok; int dg_B, dg_nb; uint64_t dg_seed, dg_replay_uid; uint64_t dg_comm_uid; bool dgraph_refused; char dg_reason[256];   // the data-parallel step (default mode)
  // graph replay of the data-parallel half step (sample + both gradient sets)
  // three variants: 0 samples its own minibatch; 1 / 2 find it presampled (by the previous call's rider, conv1_dw_gather.hip)
  // in the second / first set of slot arrays.  One key for all three.
  // hg[0]: the whole half step as one graph per variant; hg[1]: split at the conv backward (two graphs per variant) so that the
  // all-reduce of the fully-connected layers' gradients can run beside the conv backward (cpp_ddpg_dp_train_step, overlap)
  struct HalfGraphs { hipGraph_t g[3][2]; hipGraphExec_t e[3][2]; bool ok[3]; int next[3]; } hg[2];   // next: variant of the following call
  int h_B; uint64_t h_seed, h_replay_uid, h_write_gen;
  uint64_t dp_local;       // minibatches applied locally since the last parameter averaging (periodic mode)
  int sq_cnt[2];           // norm partials the last gradient pass left per list in cpp_ctx::sq_part (<= 0: none, run the sumsq kernel)
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
  d->graph = nullptr; d->gexec = nullptr; d->graph_ok = false; d->step_batch = nullptr; d->g_replay_uid = 0;
  d->rgraph = nullptr; d->rgexec = nullptr; d->rgraph_ok = false; d->rg_B = 0; d->rg_replay_uid = 0;
  d->epoch = ctx->kernel_epoch;
  d->dgraph = nullptr; d->dgexec = nullptr; d->dgraph_ok = false; d->dg_B = d->dg_nb = 0; d->dg_seed = d->dg_replay_uid = 0; d->dg_comm_uid = 0; d->dgraph_refused = false; d->dg_reason[0] = 0;
  memset(d->hg, 0, sizeof(d->hg)); d->dp_local = 0; d->sq_cnt[0] = d->sq_cnt[1] = 0;
  d->h_replay_uid = 0; d->h_write_gen = 0; d->pre_variant = 0; d->h_B = 0; d->h_seed = 0;
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
  HIP_CHECK(ctx_sync_stream(ctx));
  ctx->n_trainers += 1;
  *out = d;
  return CPP_OK;
}

static void drop_half_graphs(cpp_ddpg* d) {
  for (auto& H : d->hg)
    for (int v = 0; v < 3; ++v) {
      for (int k = 0; k < 2; ++k) {
        if (H.e[v][k]) { (void)hipGraphExecDestroy(H.e[v][k]); H.e[v][k] = nullptr; }
        if (H.g[v][k]) { (void)hipGraphDestroy(H.g[v][k]); H.g[v][k] = nullptr; }
      }
      H.ok[v] = false; H.next[v] = 0;
    }
}

// Every training entry point starts here: the context may have moved conv1 to the other kernel family since the last call (nearly
// constant channels: common.h, cpp_ctx::conv1_f32) -- the cached graphs then hold the wrong launches and a presampled minibatch may be
// in the wrong form (sampled slots against a gathered copy): everything is rebuilt by the calls' own "key changed" paths.
static void route_check(cpp_ddpg* d) {
  ctx_route_update(d->ctx);
  if (d->epoch == d->ctx->kernel_epoch) return;
  d->epoch = d->ctx->kernel_epoch;
  d->graph_ok = false; d->rgraph_ok = false; d->dgraph_ok = false;
  drop_half_graphs(d);
  d->pre_variant = 0;
  for (cpp_net* n : {d->actor, d->critic, d->tactor, d->tcritic}) n->wimg_key = nullptr;
}

extern "C" int cpp_ddpg_destroy(cpp_ddpg* d) {
  if (!d) return CPP_OK;
  (void)hipSetDevice(d->ctx->device);
  (void)ctx_sync_stream(d->ctx);
  if (d->gexec) (void)hipGraphExecDestroy(d->gexec);
  if (d->graph) (void)hipGraphDestroy(d->graph);
  if (d->dgexec) (void)hipGraphExecDestroy(d->dgexec);
  if (d->dgraph) (void)hipGraphDestroy(d->dgraph);
  if (d->rgexec) (void)hipGraphExecDestroy(d->rgexec);
  if (d->rgraph) (void)hipGraphDestroy(d->rgraph);
  drop_half_graphs(d);
  if (d->step_batch) cpp_batch_destroy(d->step_batch);
  d->actor->grads = nullptr; d->critic->grads = nullptr;
  d->ctx->n_trainers -= 1;
  d->arena.release(); delete d; return CPP_OK;
}

static int check_batch(cpp_ddpg* d, cpp_batch* b, const char* who) {
  ARG_CHECK(d && b, "%s: NULL argument", who);
  ARG_CHECK(b->B >= 1 && b->B <= d->maxB, "%s: batch size %d outside [1,%d]", who, b->B, d->maxB);
  ARG_CHECK(b->elems == d->actor->state_elems && b->A == d->actor->spec.action_dim, "%s: batch shape does not match the networks", who);
  return CPP_OK;
}

const float* white_of(cpp_batch* b, int which, int C) { return b->white + (long)which * 2 * C; }

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

// folded: take the lists' squared norms from the partials the gradient pass left (compute_gradients: sq_scope) instead of running
// the sumsq kernel -- only for gradients that are applied as computed (both lists, no scaling, nothing in between)
// next: the minibatch whose sample pass has already run (its per-row statistics are in next->part): its whitening tables are
// computed by this launch's rider instead of a stats_finalize launch behind it
static int apply(cpp_ddpg* d, bool do_actor, bool do_critic, float grad_scale, uint64_t* bump = nullptr, bool folded = false,
                 const cpp_batch* next = nullptr, int next_B = 0, int next_C = 0, long elems = 0, bool tables_done = false) {
  OptSegs s; memset(&s, 0, sizeof(s));
  s.bump = bump;
  if (d->targets_in_apply && !next && do_actor && do_critic) {      // (the outer step's last launch: both target updates leave with it)
    s.tgt[0] = d->tactor->params; s.tgt[1] = d->tcritic->params; s.tgt_coeff = d->hp.target_update_rate;
    d->tactor->wimg_key = nullptr; d->tcritic->wimg_key = nullptr;
    d->targets_applied = true;
  }
  d->targets_in_apply = false;
  if (d->publish_in_apply && !next && d->ctx->route_pin_dev) {      // (the call's last launch: step_body)
    s.pub_wmax = d->ctx->white_max_dev; s.pub_tag = d->ctx->route_tag_dev; s.pub_pin = d->ctx->route_pin_dev;
  }
  d->publish_in_apply = false;
  if (next && next_C > 0 && !tables_done) {
    s.st_part = next->part; s.st_white = next->white; s.st_nparts = next_B; s.st_jobs = 2 * next_C; s.st_C = next_C;
    s.st_count = (double)next_B * (double)(elems / next_C); s.st_eps = 1e-6; s.st_wmax = d->ctx->white_max_dev;
  }
  d->actor->wimg_key = nullptr; d->critic->wimg_key = nullptr;      // (the parameters change)
  s.nseg = 2; s.kind = OPT_SGD;
  s.p[0] = d->actor->params; s.g[0] = d->gradbuf; s.n[0] = do_actor ? d->nA : 0; s.lr[0] = d->hp.actor_learning_rate; s.group[0] = 0;
  s.p[1] = d->critic->params; s.g[1] = d->gradbuf + d->nA; s.n[1] = do_critic ? d->nC : 0; s.lr[1] = d->hp.critic_learning_rate; s.group[1] = 1;
  if (folded && do_actor && do_critic && grad_scale == 1.0f && d->sq_cnt[0] > 0 && d->sq_cnt[1] > 0) {
    s.sq = d->ctx->sq_part; s.sq_begin[0] = 0; s.sq_begin[1] = SQ_REGION; s.sq_count[0] = d->sq_cnt[0]; s.sq_count[1] = d->sq_cnt[1];
  } else {
    RC(launch_sumsq(d->ctx, s, grad_scale, d->norm_part, NORM_PARTS));
  }
  // conv1's operand images of the next minibatch ride along too (conv_rs16.h; opt_apply_kernel's second rider): both updates are in
  // this launch, the next minibatch's statistics are, conv1 of all four networks will run on that kernel
  cpp_net* inets[4] = {d->actor, d->critic, d->tactor, d->tcritic};
  const ConvL* L0 = d->actor->spec.pixel ? &d->actor->conv[0] : nullptr;
  const bool img = next && next_C > 0 && do_actor && do_critic && L0 && !d->actor->spec.use_batch_norm && !d->critic->spec.use_batch_norm &&
                   conv_rs16_ok(d->ctx, L0->Cin, L0->H, L0->W, kConvOut) && next_B >= 2;
  if (img) {
    s.img_n = 4; s.img_cin = L0->Cin;
    for (int k = 0; k < 2; ++k) {      // conv1's weights and biases open the flat buffers (cpp_net_var_info order): [w_off, b_off + nout)
      const ConvL& L = inets[k]->conv[0];
      if (L.w_off != 0 || L.b_off != L.w_off + (long)L.ks * L.ks * L.Cin * kConvOut) { s.img_n = 0; break; }
      s.img_skip[k] = L.b_off + kConvOut;
    }
  }
  if (img && s.img_n) {
    for (int j = 0; j < 4; ++j) {
      cpp_net* n = inets[j];
      const ConvL& L = n->conv[0];
      s.img[j].w = n->params + L.w_off; s.img[j].bias = n->params + L.b_off;
      s.img[j].gw = j < 2 ? s.g[j] + L.w_off : nullptr; s.img[j].gb = j < 2 ? s.g[j] + L.b_off : nullptr;
      s.img[j].rec = reinterpret_cast<unsigned char*>(n->wimg); s.img[j].seg = j < 2 ? j : 0; s.img[j].col = j < 2 ? 0 : 1; s.img[j].nout = kConvOut;
      s.img[j].white = tables_done ? next->white + (long)s.img[j].col * 2 * next_C : nullptr;
    }
  }
  // norms_out[group] is only written for lists that were applied (n > 0)
  RC(launch_opt_apply(d->ctx, s, grad_scale, d->hp.gradient_clip, d->norm_part, NORM_PARTS, d->loss_norms + 1));
  if (img && s.img_n) for (int j = 0; j < 4; ++j) inets[j]->wimg_key = next->white + (long)(j < 2 ? 0 : 1) * 2 * next_C;
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
// phase 0: everything.  The data-parallel step can split the pass at the conv backward: phase 1 = everything before it (all
// gradients of the fully connected layers are then final), phase 2 = the conv backward + the dW reductions.
static int compute_gradients(cpp_ddpg* d, cpp_batch* b, int phase = 0) {
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
  // The kernels that write the gradients also leave their share of the two lists' squared norms (cpp_ctx::sq_part): list 0 = actor,
  // list 1 = critic.  Not with batch norm (dbeta comes out of the BN backward kernels) and not for a split pass.
  struct SqScope {
    cpp_ctx* c; cpp_ddpg* d;
    SqScope(cpp_ctx* c_, cpp_ddpg* d_, bool on) : c(c_), d(d_) {
      d->sq_cnt[0] = d->sq_cnt[1] = 0;
      if (on) { c->sq_n[0] = c->sq_n[1] = 0; c->sq_conv_group[0] = 0; c->sq_conv_group[1] = 1; }
    }
    ~SqScope() {
      if (c->sq_n[0] > 0 && c->sq_n[1] > 0) { d->sq_cnt[0] = c->sq_n[0]; d->sq_cnt[1] = c->sq_n[1]; }
      c->sq_n[0] = c->sq_n[1] = -1;
      for (int& g : c->sq_conv_group) g = -1;
    }
  } sq_scope(ctx, d, phase == 0 && !a->spec.use_batch_norm && !c->spec.use_batch_norm);
  auto sqg = [&](int list, GemmArgs g) {
    const int tiles = gemm_tiles(g.M, g.N, g.K);
    if (ctx->sq_n[list] >= 0 && ctx->sq_n[list] + tiles <= SQ_REGION) { g.sq_part = ctx->sq_part + list * SQ_REGION + ctx->sq_n[list]; ctx->sq_n[list] += tiles; }
    else ctx->sq_n[list] = -1;
    return g;
  };

  // ---- forward: the four conv trunks.  conv1 saturates the chip per network; the narrow conv2 / conv3 layers
  // of all four networks share one launch each.
  int tA, tC, tTA, tTC;
  if (a->spec.pixel && !a->spec.use_batch_norm) {
    cpp_net* nets[4] = {a, c, ta, tc};
    const void* sts[4] = {s1, s1, s2, s2};
    const float* whs[4] = {w1, w1, w2, w2};
    // conv1 / conv2 (+ conv3 as conv2's tail) of the four networks in one launch per layer; the two target networks have no
    // backward pass (nets_forward_trunk_fused, rt_net.cpp)
    const int t1 = G.fn([=] { return nets_forward_trunk_fused(ctx, nets, 4, sts, whs, 2, dt, B); }, {});
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
  static const bool no_heads = cpp_switch_off("CPP_FUSED_HEADS");
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
    static const bool no_pre = cpp_switch_off("CPP_HEADS_PRE");
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
    G.gemm(sqg(0, fc_dw_args(a, a->ws[0], na - 1, B, a->ws[0].dz[na - 1])), {hk});
    adz = hk;
    for (int l = na - 2; l >= 0; --l) {
      const FcL& L = a->fc[l];
      G.gemm(sqg(0, fc_dw_args(a, a->ws[0], l, B, a->ws[0].dz[l])), {adz});
      if (pre && l == na - 2) continue;       // dz[l - 1] came out of the heads kernel
      if (l > 0)
        adz = G.gemm(fc_dx_args(a, l, B, a->ws[0].dz[l], L.n_out, 0, L.n_in, a->ws[0].dz[l - 1], L.n_in, relu_grad_epi(a, l - 1),
                                a->ws[0].fcin[l], L.n_in + 1), {adz});
      else if (a->spec.pixel)
        adz = G.gemm(fc_dx_args(a, 0, B, a->ws[0].dz[0], L.n_out, 0, a->flat, a->ws[0].dpool[2], a->flat, GE_NONE, nullptr, 0), {adz});
    }
    // ---- critic backward below its concat layer
    G.gemm(sqg(1, fc_dw_args(c, c->ws[0], nc - 1, B, c->ws[0].dz[nc - 1])), {hk});
    G.gemm(sqg(1, fc_dw_args(c, c->ws[0], cat, B, c->ws[0].dz[cat])), {hk});
    cdz = hk;
    for (int l = cat - 1; l >= 0; --l) {
      const FcL& L = c->fc[l];
      G.gemm(sqg(1, fc_dw_args(c, c->ws[0], l, B, c->ws[0].dz[l])), {cdz});
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
    G.gemm(sqg(0, fc_dw_args(a, a->ws[0], l, B, a->ws[0].dz[l])), {adz});
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
    G.gemm(sqg(1, fc_dw_args(c, c->ws[0], l, B, c->ws[0].dz[l])), {cdz});
    const int ncols = L.cat ? L.n_in - A : L.n_in;
    if (l > 0)
      cdz = G.gemm(fc_dx_args(c, l, B, c->ws[0].dz[l], L.n_out, 0, ncols, c->ws[0].dz[l - 1], ncols, GE_MUL_RELU_GRAD,
                              c->ws[0].fcin[l], L.n_in + 1), {cdz});
    else if (c->spec.pixel)
      cdz = G.gemm(fc_dx_args(c, 0, B, c->ws[0].dz[0], L.n_out, 0, c->flat, c->ws[0].dpool[2], c->flat, GE_NONE, nullptr, 0), {cdz});
  }
  }
  int conv_bwd = -1;
  if (c->spec.pixel) {     // both conv backward passes, layer by layer, two networks per launch
    cpp_net* bn[2] = {a, c};
    conv_bwd = G.fn([=] { return nets_backward_conv(ctx, bn, 2, B, s1, dt, w1); }, {adz, cdz});
  }
  DwPendingGuard pending(ctx);      // (a failure below drops what was queued)
  if (phase == 2) {
    if (conv_bwd >= 0) RC(G.ops[conv_bwd].fn());
    return flush_dw_reduce(ctx);
  }
  RC(G.run(ctx, phase == 1 ? conv_bwd : -1));
  if (phase == 1) { pending.keep(); return CPP_OK; }
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
  d->tactor->wimg_key = nullptr; d->tcritic->wimg_key = nullptr;
  return launch_soft_update(d->ctx, d->tactor->params, d->actor->params, d->nA, d->tcritic->params, d->critic->params,
                            d->nC, d->hp.target_update_rate);
}

// The fused step does not need a gathered copy of the minibatch when conv1 runs on the f16-pipe kernels: they take the
// replay store plus the sampled slots (the gather kernel then only reads -- statistics -- and writes 2 B ints).
// CPP_DIRECT_REPLAY=0 keeps the copy.
bool direct_replay_ok(cpp_net* a, cpp_replay* r, int B) {
  static const bool off = cpp_switch_off("CPP_DIRECT_REPLAY");
  if (off || !a->spec.pixel || r->store_dtype != CPP_F16) return false;
  const int C = a->spec.C;
  int g = 8, c = C; while (c) { int t = g % c; g = c; c = t; }
  if (r->elems % 8 != 0 || C / g > 16 || r->elems % C != 0) return false;       // statistics come from the gather kernel
  return conv1_f16_pipes_ok(a->ctx, C, a->conv[0].H, a->conv[0].W, B, a->spec.use_batch_norm != 0);
}

static int capture_into(cpp_ctx* ctx, hipGraph_t* g, hipGraphExec_t* e, const std::function<int()>& body);
// dp: this rank's part of the data-parallel step (cpp_ddpg_dp_train_step): between a minibatch's gradients and its update the flat
// gradient buffer is summed over the ranks (comm; NULL: a single learner on the same path) and the update takes the mean -- the
// all-reduce is issued on the context's stream, i.e. it is PART OF THE CAPTURED GRAPH (RCCL's kernels capture like any other).
static int step_body(cpp_ddpg* d, cpp_replay* r, int B, int n_batches, const int32_t* rows_dev, uint64_t seed, bool targets = true,
                     bool dp = false, cpp_comm* comm = nullptr) {
  d->pre_variant = 0;        // (the half steps' presampled minibatch lives in the same step_batch)
  const int C = d->actor->spec.pixel ? d->actor->spec.C : 0;
  cpp_ctx* ctx = d->ctx;
  const bool direct = direct_replay_ok(d->actor, r, B);
  // The sample + statistics pass of minibatch i + 1 depends on nothing minibatch i computes: it rides in the launch of i's
  // dW reductions (reduce_gather_kernel, replay.hip), keyed by the sampler's counter + 1 -- the counter itself moves in i's
  // optimiser kernel as before, so the rows drawn are the same.  Conv trunks on f16 / u8 stores; CPP_RIDE_GATHER=0: in sequence.
  static const bool no_ride = cpp_switch_off("CPP_RIDE_GATHER");
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
      static const bool no_dwride = cpp_switch_off("CPP_RIDE_DW");
      ctx->ride_at_dw = direct && !no_dwride;
      if (direct) { ga.out_slot[0] = d->step_batch->slot_alt[0]; ga.out_slot[1] = d->step_batch->slot_alt[1]; }
      ctx->ride = &ga; ctx->ride_done = false; ctx->ride_dtype = r->store_dtype;
    }
    // ... and when it leaves with conv1's dW, its statistics can be finished in the dW reductions' launch (flush_dw_reduce): the tables
    // are in memory before the optimiser's launch, whose conv1 image rider reads them (apply)
    static const bool no_stats_ride = cpp_switch_off("CPP_RIDE_STATS");
    StatsRide sr;
    if (ctx->ride && ctx->ride_at_dw && Cg > 0 && !no_stats_ride) {
      sr.part = d->step_batch->part; sr.white = d->step_batch->white; sr.nparts = B; sr.jobs = 2 * Cg; sr.C = Cg;
      sr.count = (double)B * (double)(r->elems / Cg); sr.eps = 1e-6; sr.wmax = ctx->white_max_dev;
      ctx->st_ride = &sr; ctx->st_ride_done = false;
    }
    const int rc = compute_gradients(d, d->step_batch);
    const bool rode = ctx->ride != nullptr && ctx->ride_done;
    const bool tables_done = ctx->st_ride != nullptr && ctx->st_ride_done && rode;
    ctx->ride = nullptr; ctx->st_ride = nullptr;
    if (rode && direct) { std::swap(d->step_batch->slot[0], d->step_batch->slot_alt[0]); std::swap(d->step_batch->slot[1], d->step_batch->slot_alt[1]); }
    RC(rc);
    // (also advances the sampler's counter and, when the next minibatch's sample pass rode along above, finishes its statistics --
    // unless the dW reductions' launch already has)
    const bool stats_ride = rode && Cg > 0 && !no_stats_ride;
    if (dp && comm) {
      prof_begin(ctx);
      NCCL_CHECK(ncclAllReduce(d->gradbuf, d->gradbuf, (size_t)(d->nA + d->nC), ncclFloat, ncclSum, comm->comm, ctx->stream));
      prof_end(ctx, K_ALLREDUCE);
    }
    // (dp: the norm is the reduced gradient's -- the partials the gradient kernels folded in are this rank's only: sumsq runs)
    static const bool no_tgt_ride = cpp_switch_off("CPP_RIDE_TARGETS");
    d->targets_applied = false;
    d->targets_in_apply = !more && targets && !dp && !no_tgt_ride;      // (the last minibatch of an outer step: the target updates ride in its optimiser launch)
    d->publish_in_apply = !more && (!targets || d->targets_in_apply);   // (... which then closes the call)
    RC(apply(d, true, true, (dp && comm) ? 1.0f / (float)comm->world : 1.0f, rows_dev ? nullptr : r->counter, !dp,
             stats_ride ? d->step_batch : nullptr, B, Cg, r->elems, tables_done));
    if (more) {
      if (stats_ride) { d->step_batch->B = B; d->step_batch->dtype = CPP_F16; d->step_batch->stats_C = Cg; }     // (replay_sample_finish's bookkeeping)
      else if (rode) RC(replay_sample_finish(r, B, Cg, C, d->step_batch));
      else RC(replay_sample_device(r, B, rows_dev ? rows_dev + (size_t)(i + 1) * B : nullptr, seed, rows_dev ? nullptr : r->counter, C,
                                   d->step_batch, direct));
    }
  }
  if (targets && d->targets_applied) { d->targets_applied = false; return CPP_OK; }      // (both target updates and the route's publish left with the optimiser's launch)
  if (targets) { ctx->route_rider = true; return cpp_ddpg_update_targets(d); }      // (the largest whitening scale of this step rides to the host in that launch)
  return CPP_OK;      // (... or has left with the last minibatch's optimiser launch)
}

// ddpg_cartpole.py:332-334 for ONE minibatch whose rows the HOST drew (replay_memory.random_indexes: numpy's RNG, :123-129):
// sample + gather of exactly those rows, actor update, critic update -- the fused minibatch of cpp_ddpg_train_step, without the
// target updates (the caller's loop runs them after `batches_per_step` minibatches, :336-337: cpp_net_soft_update).  This is what
// `actor.train(batch.state_1); critic.train(batch)` of the reference's loop becomes (cartpoleplusplus_amd/ddpg_cartpole.py defers the
// actor's call until the critic's arrives).  One hipGraph per (B, replay); the rows travel through pinned memory, so the call
// returns while the previous minibatch is still running.
extern "C" int cpp_ddpg_train_rows(cpp_ddpg* d, cpp_replay* r, int B, const int32_t* idxs) {
  ARG_CHECK(d && r && idxs, "cpp_ddpg_train_rows: NULL argument");
  ARG_CHECK(B >= 1 && B <= d->maxB, "cpp_ddpg_train_rows: batch %d outside [1,%d]", B, d->maxB);
  ARG_CHECK(r->elems == d->actor->state_elems && r->A == d->actor->spec.action_dim, "cpp_ddpg_train_rows: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_ddpg_train_rows: replay memory is empty"); return CPP_ERR_STATE; }
  route_check(d);
  cpp_ctx* ctx = d->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!d->step_batch) RC(cpp_batch_create(ctx, d->maxB, r->elems, r->A, &d->step_batch));
  RC(replay_stage_rows(r, idxs, B, "cpp_ddpg_train_rows"));
  static const bool no_graph = cpp_switch_set("CPP_NO_GRAPH");
  if (ctx->prof || no_graph) return step_body(d, r, B, 1, r->rows_in, 0, false);
  if (!d->rgraph_ok || d->rg_B != B || d->rg_replay_uid != r->uid) {
    if (d->rgexec) { (void)hipGraphExecDestroy(d->rgexec); d->rgexec = nullptr; }
    if (d->rgraph) { (void)hipGraphDestroy(d->rgraph); d->rgraph = nullptr; }
    d->rgraph_ok = false;
    RC(step_body(d, r, B, 1, r->rows_in, 0, false));          // eager pass: kernel attributes; it is also this call's minibatch
    HIP_CHECK(ctx_sync_stream(ctx));
    RC(capture_into(ctx, &d->rgraph, &d->rgexec, [&] { return step_body(d, r, B, 1, r->rows_in, 0, false); }));
    d->rgraph_ok = true; d->rg_B = B; d->rg_replay_uid = r->uid;
    return CPP_OK;
  }
  HIP_CHECK(hipGraphLaunch(d->rgexec, ctx->stream));
  d->pre_variant = 0;
  d->loss_parts = d->heads_grid; d->loss_B = d->heads_B;
  return CPP_OK;
}

extern "C" int cpp_ddpg_train_step(cpp_ddpg* d, cpp_replay* r, int B, int n_batches, const int32_t* idxs, uint64_t seed) {
  ARG_CHECK(d && r, "cpp_ddpg_train_step: NULL argument");
  ARG_CHECK(B >= 1 && B <= d->maxB, "cpp_ddpg_train_step: batch %d outside [1,%d]", B, d->maxB);
  ARG_CHECK(n_batches >= 1 && (size_t)n_batches * B <= 65536, "cpp_ddpg_train_step: n_batches %d", n_batches);
  ARG_CHECK(r->elems == d->actor->state_elems && r->A == d->actor->spec.action_dim, "cpp_ddpg_train_step: replay shape does not match the networks");
  if (r->size <= 0) { cpp_set_error("cpp_ddpg_train_step: replay memory is empty"); return CPP_ERR_STATE; }
  route_check(d);
  cpp_ctx* ctx = d->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  if (!d->step_batch) RC(cpp_batch_create(ctx, d->maxB, r->elems, r->A, &d->step_batch));
  if (idxs) {
    for (int i = 0; i < n_batches * B; ++i)
      ARG_CHECK(idxs[i] >= 0 && idxs[i] < r->size, "cpp_ddpg_train_step: index %d outside [0,%d)", idxs[i], r->size);
    HIP_CHECK(hipMemcpyAsync(r->rows_in, idxs, (size_t)n_batches * B * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    return step_body(d, r, B, n_batches, r->rows_in, seed);
  }
  static const bool no_graph = cpp_switch_set("CPP_NO_GRAPH");   // plain in-order stream launches (A/B measurements)
  if (ctx->prof || no_graph) return step_body(d, r, B, n_batches, nullptr, seed);
  if (!d->graph_ok || d->g_B != B || d->g_nb != n_batches || d->g_seed != seed || d->g_replay_uid != r->uid) {
    if (d->gexec) { (void)hipGraphExecDestroy(d->gexec); d->gexec = nullptr; }
    if (d->graph) { (void)hipGraphDestroy(d->graph); d->graph = nullptr; }
    d->graph_ok = false;
    // one eager pass first: it sets every kernel's LDS attribute (not allowed during capture)
    RC(step_body(d, r, B, n_batches, nullptr, seed));
    HIP_CHECK(ctx_sync_stream(ctx));
    HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = step_body(d, r, B, n_batches, nullptr, seed);
    hipError_t e = hipStreamEndCapture(ctx->stream, &d->graph);
    if (rc) return rc;
    if (e != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e)); return CPP_ERR_HIP; }
    HIP_CHECK(hipGraphInstantiate(&d->gexec, d->graph, nullptr, nullptr, 0));
    d->graph_ok = true; d->g_B = B; d->g_nb = n_batches; d->g_seed = seed; d->g_replay_uid = r->uid;
    return CPP_OK;   // the eager pass above was this call's step
  }
  HIP_CHECK(hipGraphLaunch(d->gexec, ctx->stream));
  d->loss_parts = d->heads_grid; d->loss_B = d->heads_B;
  return CPP_OK;
}

// variant 0: sample + gather + statistics of this call's minibatch; 1 / 2: it was presampled by the previous call's rider into
// slot set 1 / 0 (only its whitening tables are still to do).  Every variant tries to send the NEXT minibatch's sample pass
// along with conv1's dW (the sampler's counter has been advanced by then, so the rider draws with the counter as it stands);
// *next: the variant the following call must use.  CPP_RIDE_DP=0 (ablation build): always variant 0, no rider.
// phase 0: the whole half step; 1: up to the conv backward; 2: the conv backward (with the rider) + dW reductions.
static int half_step_body(cpp_ddpg* d, cpp_replay* r, int B, uint64_t seed, int variant, int* next, int phase) {
  const int C = d->actor->spec.pixel ? d->actor->spec.C : 0;
  cpp_ctx* ctx = d->ctx;
  cpp_batch* b = d->step_batch;
  const bool direct = direct_replay_ok(d->actor, r, B);
  static const bool no_ride = cpp_switch_off("CPP_RIDE_DP");
  const int cur = variant == 1 ? 1 : 0;
  for (int k = 0; k < 2; ++k) { b->slot[k] = d->slot_set[cur][k]; b->slot_alt[k] = d->slot_set[1 - cur][k]; }
  int Cg = 0;
  GatherArgs ga = replay_gather_args(r, B, nullptr, seed, r->counter, C, b, direct, &Cg);
  if (phase != 2) {
    if (variant == 0) RC(launch_gather_stats(ctx, ga, r->store_dtype));
    bool bumped = false;
    RC(replay_sample_finish(r, B, Cg, C, b, r->counter, &bumped));
    if (!bumped) RC(launch_counter_add(ctx, r->counter, 1));
  }
  if (phase == 1) return compute_gradients(d, b, 1);
  const bool ride_ok = !no_ride && direct && Cg > 0 && r->store_dtype == CPP_F16;
  if (ride_ok) {
    ga.out_slot[0] = b->slot_alt[0]; ga.out_slot[1] = b->slot_alt[1];
    ctx->ride = &ga; ctx->ride_done = false; ctx->ride_dtype = r->store_dtype; ctx->ride_at_dw = true;
  }
  const int rc = compute_gradients(d, b, phase);
  const bool rode = ctx->ride != nullptr && ctx->ride_done;
  ctx->ride = nullptr;
  *next = rode ? (cur == 0 ? 1 : 2) : 0;
  if (rc) return rc;
  return ctx_route_publish(ctx);
}

static int capture_into(cpp_ctx* ctx, hipGraph_t* g, hipGraphExec_t* e, const std::function<int()>& body) {
  HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  const hipError_t err = hipStreamEndCapture(ctx->stream, g);
  if (rc) return rc;
  if (err != hipSuccess) { cpp_set_error("hipStreamEndCapture -> %s", hipGetErrorString(err)); return CPP_ERR_HIP; }
  HIP_CHECK(hipGraphInstantiate(e, *g, nullptr, nullptr, 0));
  return CPP_OK;
}

// One half step: sample (or find presampled) a minibatch and leave both gradient sets in the flat buffer.  hipGraph replay after
// the first call per (variant, B, seed, replay).  split: two graphs per variant, `between` is called on the host between their
// launches (the data-parallel step starts the all-reduce of the fully connected layers' gradients there).
static int half_step(cpp_ddpg* d, cpp_replay* r, int B, uint64_t seed, bool split, const std::function<int()>& between) {
  cpp_ctx* ctx = d->ctx;
  if (!d->step_batch) RC(cpp_batch_create(ctx, d->maxB, r->elems, r->A, &d->step_batch));
  if (d->slot_set[0][0] == nullptr)
    for (int k = 0; k < 2; ++k) { d->slot_set[0][k] = d->step_batch->slot[k]; d->slot_set[1][k] = d->step_batch->slot_alt[k]; }
  const bool key_ok = d->h_B == B && d->h_seed == seed && d->h_replay_uid == r->uid;
  if (!key_ok) {                                     // another batch size / seed / memory: start over
    drop_half_graphs(d);
    d->pre_variant = 0;
    d->h_B = B; d->h_seed = seed; d->h_replay_uid = r->uid;
  }
  // a minibatch the previous call's rider presampled is only good while the memory is as it was: an episode added since may have
  // overwritten its rows or recycled their state slots (the slots are read at step time).  Draw again (same counter, new contents).
  if (d->h_write_gen != r->write_gen) { d->pre_variant = 0; d->h_write_gen = r->write_gen; }
  cpp_ddpg::HalfGraphs& H = d->hg[split ? 1 : 0];
  const int v = d->pre_variant;
  d->pre_variant = 0;                                // (stays 0 if anything below fails)
  auto eager = [&](int variant, int* nx) -> int {
    if (!split) return half_step_body(d, r, B, seed, variant, nx, 0);
    RC(half_step_body(d, r, B, seed, variant, nx, 1));
    if (between) RC(between());
    return half_step_body(d, r, B, seed, variant, nx, 2);
  };
  auto capture = [&](int variant, int* nx) -> int {
    if (!split) return capture_into(ctx, &H.g[variant][0], &H.e[variant][0], [&] { return half_step_body(d, r, B, seed, variant, nx, 0); });
    RC(capture_into(ctx, &H.g[variant][0], &H.e[variant][0], [&] { return half_step_body(d, r, B, seed, variant, nx, 1); }));
    return capture_into(ctx, &H.g[variant][1], &H.e[variant][1], [&] { return half_step_body(d, r, B, seed, variant, nx, 2); });
  };
  int next = 0;
  if (ctx->prof) { RC(eager(v, &next)); d->pre_variant = next; return CPP_OK; }
  if (!H.ok[v]) {
    // Variants 1 / 2 consume a presampled minibatch: their work must be done by the captured graph's first launch (an eager
    // pass would consume it and leave another one behind).  Kernel attributes (LDS sizes: not allowed during capture) are set
    // by variant 0's eager pass, which is also that call's work.
    int nx = 0;
    if (v == 0) {
      RC(eager(0, &next));
      HIP_CHECK(ctx_sync_stream(ctx));
      RC(capture(0, &nx));
      H.ok[0] = true; H.next[0] = nx;
      d->pre_variant = next;
      return CPP_OK;
    }
    HIP_CHECK(ctx_sync_stream(ctx));
    RC(capture(v, &nx));
    H.ok[v] = true; H.next[v] = nx;
  }
  HIP_CHECK(hipGraphLaunch(H.e[v][0], ctx->stream));
  if (split) {
    if (between) RC(between());
    HIP_CHECK(hipGraphLaunch(H.e[v][1], ctx->stream));
  }
  d->pre_variant = H.next[v];
  d->loss_parts = d->heads_grid; d->loss_B = d->heads_B;
  return CPP_OK;
}

static int half_step_checks(cpp_ddpg* d, cpp_replay* r, int B, const char* who) {
  ARG_CHECK(d && r, "%s: NULL argument", who);
  ARG_CHECK(B >= 1 && B <= d->maxB, "%s: batch %d outside [1,%d]", who, B, d->maxB);
  ARG_CHECK(r->elems == d->actor->state_elems && r->A == d->actor->spec.action_dim, "%s: replay shape does not match the networks", who);
  if (r->size <= 0) { cpp_set_error("%s: replay memory is empty", who); return CPP_ERR_STATE; }
  return CPP_OK;
}

extern "C" int cpp_ddpg_sample_and_compute(cpp_ddpg* d, cpp_replay* r, int B, uint64_t seed) {
  if (d) route_check(d);
  RC(half_step_checks(d, r, B, "cpp_ddpg_sample_and_compute"));
  HIP_CHECK(hipSetDevice(d->ctx->device));
  return half_step(d, r, B, seed, false, nullptr);
}

// ---- collectives of the data-parallel learners (communicator: rt_comm.cpp) --------------------------------------
// the flat gradient buffer is [actor conv | actor fc | critic conv | critic fc]: offsets of the two fc parts
static long fc_start(const cpp_net* n) { return n->fc[0].w_off; }

extern "C" int cpp_ddpg_allreduce_grads(cpp_ddpg* d, cpp_comm* c) {
  ARG_CHECK(d && c, "cpp_ddpg_allreduce_grads: NULL argument");
  ARG_CHECK(c->ctx == d->ctx, "cpp_ddpg_allreduce_grads: communicator and networks live on different contexts");
  HIP_CHECK(hipSetDevice(d->ctx->device));
  NCCL_CHECK(ncclAllReduce(d->gradbuf, d->gradbuf, (size_t)(d->nA + d->nC), ncclFloat, ncclSum, c->comm, d->ctx->stream));
  return CPP_OK;
}

// periodic mode: replicas that took k local steps meet again at the mean of their parameters (targets included: they are
// functions of the parameter history and would otherwise drift apart)
extern "C" int cpp_ddpg_average_params(cpp_ddpg* d, cpp_comm* c) {
  ARG_CHECK(d && c, "cpp_ddpg_average_params: NULL argument");
  ARG_CHECK(c->ctx == d->ctx, "cpp_ddpg_average_params: communicator and networks live on different contexts");
  HIP_CHECK(hipSetDevice(d->ctx->device));
  cpp_net* nets[4] = {d->actor, d->critic, d->tactor, d->tcritic};
  NCCL_CHECK(ncclGroupStart());
  for (cpp_net* n : nets)
    NCCL_CHECK(ncclAllReduce(n->params, n->params, (size_t)n->nparams, ncclFloat, ncclAvg, c->comm, d->ctx->stream));
  NCCL_CHECK(ncclGroupEnd());
  d->dp_local = 0;
  return CPP_OK;
}

// The inner step ddpg_cartpole.py:331-337 for N synchronous learners (this rank's part).  Per minibatch: sample from the own
// replay shard + both gradient sets (hipGraph) -> sum over the ranks of the flat gradient buffer -> clip + SGD on the mean on
// every rank (identical inputs: the replicas stay bit-identical without a broadcast).  sync_every = k > 1 ("periodic"): k local
// minibatch updates, then the parameters are averaged.  overlap: the gradients of the fully connected layers (93 % of the
// buffer) are reduced on a second stream while the conv backward of the same minibatch runs; the conv layers' follow.
// comm == NULL: a single learner taking the same path (tests).  Whitening statistics and target updates are local.
extern "C" int cpp_ddpg_dp_train_step(cpp_ddpg* d, cpp_replay* r, cpp_comm* c, int B, int n_batches, uint64_t seed,
                                      int sync_every, int overlap) {
  RC(half_step_checks(d, r, B, "cpp_ddpg_dp_train_step"));
  ARG_CHECK(n_batches >= 1 && sync_every >= 1, "cpp_ddpg_dp_train_step: n_batches %d, sync_every %d", n_batches, sync_every);
  ARG_CHECK(!c || c->ctx == d->ctx, "cpp_ddpg_dp_train_step: communicator and networks live on different contexts");
  route_check(d);
  cpp_ctx* ctx = d->ctx;
  HIP_CHECK(hipSetDevice(ctx->device));
  const float inv = c ? 1.0f / (float)c->world : 1.0f;
  const long fa = fc_start(d->actor), fcr = fc_start(d->critic);
  static const bool no_dp_graph = cpp_switch_off("CPP_DP_GRAPH");
  if (sync_every == 1 && !overlap && !no_dp_graph) {
    // The default mode as ONE hipGraph per outer step (round 4): sample -> gradients -> ncclAllReduce -> norm -> clip + SGD for each of
    // the n_batches minibatches, then the target updates -- the single learner's fused step (step_body) with the collective and the
    // norm of the REDUCED gradient inside.  No host launch, copy or fill per minibatch; the sample pass of minibatch i + 1 rides in
    // i's conv1 dW as in the fused step.  (Rounds 2-3: one graph per half step, the all-reduce, sumsq and the optimiser as host
    // launches per minibatch: 0.946 of the fused step at world size 1.)
    if (!d->step_batch) RC(cpp_batch_create(ctx, d->maxB, r->elems, r->A, &d->step_batch));
    if (ctx->prof) return step_body(d, r, B, n_batches, nullptr, seed, true, true, c);
    if (d->dgraph_refused) return step_body(d, r, B, n_batches, nullptr, seed, true, true, c);
    if (!d->dgraph_ok || d->dg_B != B || d->dg_nb != n_batches || d->dg_seed != seed || d->dg_replay_uid != r->uid || d->dg_comm_uid != (c ? c->uid : 0)) {
      if (d->dgexec) { (void)hipGraphExecDestroy(d->dgexec); d->dgexec = nullptr; }
      if (d->dgraph) { (void)hipGraphDestroy(d->dgraph); d->dgraph = nullptr; }
      d->dgraph_ok = false;
      RC(step_body(d, r, B, n_batches, nullptr, seed, true, true, c));      // eager pass: kernel attributes; it is also this call's step
      HIP_CHECK(ctx_sync_stream(ctx));
      if (d->dgraph_refused) return CPP_OK;           // (capture failed once on this trainer: every step takes the eager sequence above)
      // No N > 1 box has run this yet: if the runtime or RCCL refuses to capture / instantiate the step with the collective inside,
      // the trainer keeps the SAME sequence as plain stream launches (identical arithmetic on every rank, no graph) instead of failing.
      if (capture_into(ctx, &d->dgraph, &d->dgexec, [&] { return step_body(d, r, B, n_batches, nullptr, seed, true, true, c); }) != CPP_OK) {
        (void)hipGetLastError();
        if (d->dgexec) { (void)hipGraphExecDestroy(d->dgexec); d->dgexec = nullptr; }
        if (d->dgraph) { (void)hipGraphDestroy(d->dgraph); d->dgraph = nullptr; }
        // (the very same calls have just run eagerly and returned CPP_OK: whatever fails here fails BECAUSE of the capture -- the
        // runtime's or RCCL's refusal.  The reason is kept for cpp_ddpg_dp_status; a failure of the eager pass above is returned.)
        d->dgraph_refused = true;
        snprintf(d->dg_reason, sizeof(d->dg_reason), "%s", cpp_last_error());
        fprintf(stderr, "cartpolepp: the data-parallel step could not be captured as a hipGraph (%s); running it as stream launches\n", d->dg_reason);
        return CPP_OK;
      }
      d->dgraph_ok = true; d->dg_B = B; d->dg_nb = n_batches; d->dg_seed = seed; d->dg_replay_uid = r->uid; d->dg_comm_uid = c ? c->uid : 0;
      return CPP_OK;
    }
    HIP_CHECK(hipGraphLaunch(d->dgexec, ctx->stream));
    d->pre_variant = 0;
    d->loss_parts = d->heads_grid; d->loss_B = d->heads_B;
    return CPP_OK;
  }
  for (int i = 0; i < n_batches; ++i) {
    if (sync_every > 1) {                           // local update; every k-th one is followed by the parameter averaging
      RC(half_step(d, r, B, seed, false, nullptr));
      RC(apply(d, true, true, 1.0f));
      if (++d->dp_local >= (uint64_t)sync_every && c) RC(cpp_ddpg_average_params(d, c));
      continue;
    }
    if (c && overlap) {
      RC(half_step(d, r, B, seed, true, [&]() -> int {        // between the two graphs: the fc gradients are final
        HIP_CHECK(hipEventRecord(c->ev_fc, ctx->stream));
        HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_fc, 0));
        NCCL_CHECK(ncclGroupStart());
        NCCL_CHECK(ncclAllReduce(d->gradbuf + fa, d->gradbuf + fa, (size_t)(d->nA - fa), ncclFloat, ncclSum, c->comm, c->side));
        NCCL_CHECK(ncclAllReduce(d->gradbuf + d->nA + fcr, d->gradbuf + d->nA + fcr, (size_t)(d->nC - fcr), ncclFloat, ncclSum, c->comm, c->side));
        NCCL_CHECK(ncclGroupEnd());
        return CPP_OK; }));
      HIP_CHECK(hipEventRecord(c->ev_bwd, ctx->stream));      // conv backward + dW reductions done: the conv parts follow
      HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_bwd, 0));
      if (fa > 0 || fcr > 0) {
        NCCL_CHECK(ncclGroupStart());
        if (fa > 0) NCCL_CHECK(ncclAllReduce(d->gradbuf, d->gradbuf, (size_t)fa, ncclFloat, ncclSum, c->comm, c->side));
        if (fcr > 0) NCCL_CHECK(ncclAllReduce(d->gradbuf + d->nA, d->gradbuf + d->nA, (size_t)fcr, ncclFloat, ncclSum, c->comm, c->side));
        NCCL_CHECK(ncclGroupEnd());
      }
      HIP_CHECK(hipEventRecord(c->ev_done, c->side));
      HIP_CHECK(hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
    } else {
      RC(half_step(d, r, B, seed, false, nullptr));
      if (c) RC(cpp_ddpg_allreduce_grads(d, c));
    }
    RC(apply(d, true, true, inv));
  }
  return cpp_ddpg_update_targets(d);
}

// which form the default data-parallel step of this trainer takes: 0 = none run yet, 1 = one hipGraph replay per outer step (the
// collective inside), 2 = the same sequence as stream launches (the capture was refused; `reason` says by what)
extern "C" int cpp_ddpg_dp_status(const cpp_ddpg* d, int* mode, char* reason, int cap) {
  ARG_CHECK(d && mode, "cpp_ddpg_dp_status: NULL argument");
  *mode = d->dgraph_refused ? 2 : (d->dgraph_ok ? 1 : 0);
  if (reason && cap > 0) snprintf(reason, (size_t)cap, "%s", d->dg_reason);
  return CPP_OK;
}

extern "C" int cpp_ddpg_last_stats(cpp_ddpg* d, float out[3]) {
  ARG_CHECK(d && out, "cpp_ddpg_last_stats: NULL argument");
  HIP_CHECK(hipMemcpyAsync(out, d->loss_norms, 3 * sizeof(float), hipMemcpyDeviceToHost, d->ctx->stream));
  double parts[DDPG_HEADS_MAX_WGS];
  if (d->loss_parts > 0)
    HIP_CHECK(hipMemcpyAsync(parts, d->heads_part, (size_t)d->loss_parts * sizeof(double), hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_CHECK(ctx_sync_stream(d->ctx));
  if (d->loss_parts > 0) {                          // fused heads kernel: mean(td^2) from its per-workgroup partials, fixed order
    double s = 0.0;
    for (int i = 0; i < d->loss_parts; ++i) s += parts[i];
    out[0] = (float)(s / (double)d->loss_B);
  }
  return CPP_OK;
}

extern "C" int cpp_ddpg_last_values(cpp_ddpg* d, int B, float* actions, float* dq_da, float* q, float* td) {
  ARG_CHECK(d, "cpp_ddpg_last_values: NULL argument");
  ARG_CHECK(B >= 1 && B <= d->maxB, "cpp_ddpg_last_values: batch %d outside [1,%d]", B, d->maxB);
  HIP_CHECK(hipSetDevice(d->ctx->device));
  hipStream_t st = d->ctx->stream;
  const int A = d->actor->spec.action_dim;
  if (actions) HIP_CHECK(hipMemcpyAsync(actions, d->actor->ws[0].out, (size_t)B * A * sizeof(float), hipMemcpyDeviceToHost, st));
  if (dq_da) HIP_CHECK(hipMemcpyAsync(dq_da, d->dq_da, (size_t)B * A * sizeof(float), hipMemcpyDeviceToHost, st));
  if (q) HIP_CHECK(hipMemcpyAsync(q, d->critic->ws[0].out, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, st));
  if (td) HIP_CHECK(hipMemcpyAsync(td, d->td, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

