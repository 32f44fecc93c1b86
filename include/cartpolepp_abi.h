/*
 * cartpolepp_abi.h -- C ABI of libcartpolepp_hip.so (MI355X / gfx950).
 *
 * The reference (matpalm/cartpoleplusplus, /root/reference) has no FFI boundary of its own: its
 * DDPG-from-pixels hot path sits behind plain Python classes that call TensorFlow.  This header is
 * the boundary inserted directly beneath those classes; every entry point cites the reference
 * interface it replaces (paths relative to /root/reference).  Plain pointers and sizes only: no
 * torch types, no Python objects, no C++ exceptions cross.  INTEGRATION.md shows the ctypes stubs a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns CPP_OK (0) or a CPP_ERR_* code; cpp_last_error() gives the message
 *     (thread-local).  Python wrappers raise RuntimeError(cpp_last_error()).
 *   - the library owns all device memory behind opaque handles; every *_create has a *_destroy.
 *   - host pointers are read/written only for the duration of the call.
 *   - a cpp_ctx is one GPU + one HIP stream; calls on one ctx are serialised by the caller
 *     (the reference is single-threaded: one implicit tf.Session).
 *   - results are f32-grade: IEEE f32 accumulation of products of the reference's f32 / f16 operands, far inside the 1e-5 the
 *     parity tests allow and measured as close to a float64 evaluation as f32-input matrix instructions get.  Where an operand
 *     already is an f16 number (conv1's input: the replay store's pixels, replay_memory.py:32) the other, f32, operand is split by
 *     round-to-nearest into f16 pieces and the f16 x f16 products -- each exact -- are accumulated in f32 (v_mfma_f32_16x16x32_f16):
 *     two pieces under CPP_PRECISION_FAST (the default: the operand to within 2^-22 relative, about one f32 ulp), three under
 *     CPP_PRECISION_EXACT (the operand itself).  conv2's forward and dW split BOTH f32 operands into three bf16 pieces (exactly) and
 *     issue the six largest of the nine piece products (FAST: the dropped three are at most half an f32 ulp of the product) or all
 *     nine (EXACT) (v_mfma_f32_16x16x32_bf16); everything else multiplies f32 operands directly (v_mfma_f32_16x16x4_f32).  Both
 *     modes are in the one release library: cpp_ctx_set_precision below.  DESIGN.md section 4.
 *     f32 states, odd layouts and B = 1 run on the f32-input MFMA kernels throughout.  The release library has no run-time kernel
 *     switches; the ablation build (libcartpolepp_hip_ablation.so, CARTPOLEPP_ABLATION=1) can force the f32-input kernels
 *     everywhere (CPP_CONV_K16=0 CPP_CONV_B16=0) -- bench.py's `control` run; bench.py's `control_exact_products` run is the
 *     release library under CPP_PRECISION_EXACT.
 *   - replay states are stored as f16 exactly like replay_memory.py:32 (or as their 8-bit pixel codes); indices are int32.
 *   - state batches handed to the conv kernels always live in the library's own guard-banded device allocations (host
 *     pointers are copied in first); the f16-pipe kernels refuse anything else.
 *   - flat parameter order = TF variable creation order "<scope>/weights", "<scope>/biases":
 *     conv1, conv2, conv3, then the fully connected layers.
 */
#ifndef CARTPOLEPP_ABI_H
#define CARTPOLEPP_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPP_ABI_VERSION 1

enum { CPP_OK = 0, CPP_ERR_ARG = 1, CPP_ERR_HIP = 2, CPP_ERR_STATE = 3, CPP_ERR_NUMERIC = 4 };
enum { CPP_F32 = 0, CPP_F16 = 1,           /* host/device element type of state payloads        */
       CPP_U8 = 2 };                        /* replay store only: 8-bit pixel codes k, read back as f16(k/255) */
enum { CPP_ACTOR = 0, CPP_CRITIC = 1,       /* ddpg_cartpole.py:78 ActorNetwork / :148 CriticNetwork */
       CPP_HEAD = 2 };                      /* naf_cartpole.py: state network + one 'fc' head (value / mu / l_values) */
enum { CPP_OPT_SGD = 0, CPP_OPT_MOMENTUM = 1, CPP_OPT_ADAM = 2 };   /* util.py:73-76 tf.train.<name>Optimizer */

typedef struct cpp_ctx cpp_ctx;
typedef struct cpp_net cpp_net;
typedef struct cpp_batch cpp_batch;
typedef struct cpp_replay cpp_replay;
typedef struct cpp_ddpg cpp_ddpg;
typedef struct cpp_naf cpp_naf;
typedef struct cpp_comm cpp_comm;

/* ---- library / context ------------------------------------------------------------------- */
int cpp_abi_version(void);
const char* cpp_last_error(void);

/* One GPU + one stream.  `hip_stream` may be NULL (library creates its own non-blocking stream) or
 * an existing hipStream_t (e.g. torch.cuda.Stream().cuda_stream) so that host-side RCCL collectives
 * issued through torch.distributed are ordered with the kernels.  Replaces the implicit default
 * tf.Session of ddpg_cartpole.py:416. */
int cpp_ctx_create(int device_id, void* hip_stream, cpp_ctx** out);
int cpp_ctx_destroy(cpp_ctx* ctx);
int cpp_sync(cpp_ctx* ctx);

/* The arithmetic contract of the conv kernels that run on the f16 / bf16 matrix pipes (conv1 forward and dW, conv2 forward and
 * dW; everything else is f32-input MFMA or f32 VALU whatever the mode).  Accumulation is f32 and every issued product is exact
 * in both modes; the modes differ in how much of an f32 OPERAND reaches the multiplier:
 *   CPP_PRECISION_FAST  (default)  conv1's f32 operand (W s, dY) as two f16 pieces: the operand to within 2^-22 relative (~ one
 *                                  f32 ulp); conv2's operands as three bf16 pieces each with the six largest of the nine piece
 *                                  products (the dropped three: <= 2^-23 of a product).  Against the float64 oracle it is as
 *                                  close as the f32-input MFMA kernels (tests/test_gpu_fullsize.py, tests/test_gpu_render_inputs.py).
 *   CPP_PRECISION_EXACT            three f16 pieces / all nine products: no operand bit is dropped, the result is the
 *                                  f32-accumulated sum of the exact products of the f32 operands (rounds 1-2; ~0.87 x the speed).
 * The reference's TF CPU kernels multiply and accumulate in f32 (base_network.py:103-123 -> Eigen): both modes sit inside what
 * one f32 FMA chain rounds.  Must be called before a trainer (cpp_ddpg / cpp_naf) is created on the ctx -- captured step graphs
 * hold the kernels of the mode they were captured in -- and fails with CPP_ERR_ARG afterwards. */
#define CPP_PRECISION_FAST 0
#define CPP_PRECISION_EXACT 1
int cpp_ctx_set_precision(cpp_ctx* ctx, int mode);
int cpp_ctx_get_precision(cpp_ctx* ctx, int* mode);
/* Nearly constant channels.  The f16-pipe conv1 kernels multiply the replay store's RAW pixels by whitened weights; on a channel whose
 * whitening scale is ~10^3 and whose values do not cancel exactly (a blind camera with a rare off-colour pixel) that sits a few times
 * further from a float64 evaluation than whitening each element first, as base_network.py:95-99 does.  Every training call leaves the
 * largest scale of its whitening tables in pinned host memory, tagged with the call's number; above `threshold` (default 100; 0 =
 * never) later calls run conv1 forward / dW and conv2 forward on the f32-input kernels (which whiten per element) until the scale has
 * fallen under half of it.  WHICH call's scale a call decides from depends on the order of the caller's calls only, never on host /
 * GPU timing: call k reads call k - 2's (behind an event that guarantees it has landed; no wait in a loop that runs ahead of the
 * GPU), or call k - 1's if the stream has been synchronised since (cpp_sync, a parameter read, the eager pass in front of a graph
 * capture).  The first one or two affected calls therefore still run on the f16 pipes -- the same ones in every run.
 * cpp_ctx_get_route reports the current choice and the scale of the newest call known to have finished. */
int cpp_ctx_set_route_threshold(cpp_ctx* ctx, float threshold);
int cpp_ctx_get_route(cpp_ctx* ctx, int* conv1_f32, float* last_max_scale);

/* HIP-event stopwatch on the ctx stream (bench.py; the reference only has util.StopWatch, util.py:22). */
int cpp_timer_begin(cpp_ctx* ctx);
int cpp_timer_end(cpp_ctx* ctx, float* elapsed_ms);

/* Per-kernel HIP-event profile of the launches issued on this ctx (disables graph replay while on).
 * kernel ids: see cpp_prof_kernel_name(). */
int cpp_prof_enable(cpp_ctx* ctx, int on);
int cpp_prof_reset(cpp_ctx* ctx);
int cpp_prof_num_kernels(void);
const char* cpp_prof_kernel_name(int kernel_id);
int cpp_prof_read(cpp_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

/* ---- networks (base_network.py:13-134, ddpg_cartpole.py:78-100, :148-184) ------------------- */
typedef struct cpp_net_spec {
  int32_t kind;          /* CPP_ACTOR / CPP_CRITIC                                              */
  int32_t pixel;         /* opts.use_raw_pixels: conv trunk (base_network.py:73-127) in front    */
  int32_t H, W, C;       /* pixel: image dims, C = 3*num_cameras*action_repeats (:85-90)         */
  int32_t state_elems;   /* low-dim: flattened state length (repeats*2*7, bullet_cartpole.py:125) */
  int32_t action_dim;
  int32_t n_hidden;      /* actor: opts.actor_hidden_layers; low-dim critic: critic_hidden_layers */
  int32_t hidden[8];     /* (pixel critic is the fixed 200/50/+action/50 head of :168-171)       */
  int32_t head_out;      /* CPP_HEAD: outputs of the 'fc' head (1, action_dim, action_dim*(action_dim+1)/2) */
  int32_t head_act;      /* CPP_HEAD: 0 linear, 2 tanh (naf_cartpole.py:109,161,184)              */
  int32_t use_batch_norm; /* opts.use_batch_norm (base_network.py:74-79): slim.batch_norm after every conv; the conv then has no
                           * bias and the '<conv>/biases' slot of the flat layout is '<conv>/BatchNorm/beta'.  Training-mode
                           * entry points (the train ops) use batch statistics, inference-mode ones (cpp_net_forward*,
                           * check_loss, NAF action / debug) the never-updated moving averages (mean 0, variance 1).  */
  int32_t use_dropout;   /* opts.use_dropout (base_network.py:69-70): slim.dropout (keep 0.5) after the ReLU of every layer made by
                          * hidden_layers_starting_at *with opts* -- the actor's and the NAF networks' hidden stacks; the critics
                          * have none (ddpg_cartpole.py:168-177).  Training-mode entry points draw the keep bits from
                          * Philox4x32-10(key = dropout_seed; counter = (row * units + unit, layer, forward count)). */
  uint32_t dropout_seed;
} cpp_net_spec;

int cpp_net_create(cpp_ctx* ctx, const cpp_net_spec* spec, int max_batch, cpp_net** out);
int cpp_net_destroy(cpp_net* net);
/* Network.trainable_model_vars (base_network.py:51-56): variables in creation order. */
int64_t cpp_net_num_params(const cpp_net* net);
int cpp_net_num_vars(const cpp_net* net);
int cpp_net_var_info(const cpp_net* net, int i, char* name, int name_cap, int* rank,
                     int shape[4], int64_t* offset);
/* variable init / checkpoint restore / SIGUSR2 weight dump (ddpg_cartpole.py:421-427, :402-409) */
int cpp_net_set_params(cpp_net* net, const float* host, int64_t n);
int cpp_net_get_params(cpp_net* net, float* host, int64_t n);
int cpp_net_get_grads(cpp_net* net, float* host, int64_t n);   /* pre-clip gradients of last train */
/* Network._create_variables_copy_op (base_network.py:20-33): target -= coeff * (target - source). */
int cpp_net_soft_update(cpp_net* target, const cpp_net* source, float coeff);
/* session.run(output_action | q_value) on a fed state batch (ddpg_cartpole.py:123-125).  `state` is
 * a host array (B, state_elems) of `state_dtype`; `action` (B, action_dim) host f32, critics only.
 * Whitening always uses the statistics of THIS batch (base_network.py:95-99), also at B = 1. */
int cpp_net_forward(cpp_net* net, const void* state, int state_dtype, int B, const float* action,
                    float* out);
/* B independent `action_given` calls (ddpg_cartpole.py:121-126 once per row) in one pass: every image is whitened with
 * its OWN statistics (base_network.py:95-99 at batch size 1), so row i of `out` equals cpp_net_forward on row i alone.
 * Rollout-side inference for many env workers. */
int cpp_net_forward_each(cpp_net* net, const void* state, int state_dtype, int B, const float* action,
                         float* out);
/* Network.pool1/2/3 (base_network.py:108,116,124) of the last forward: which = 1..3, (B,h,w,10).
 * Debug: which = 11..13 returns the arg-max codes (0..3, as floats) of the same layers' 2x2 windows. */
int cpp_net_get_pool(cpp_net* net, int which, int B, float* out);

/* ---- minibatch resident in HBM (replay_memory.py:9 Batch) ----------------------------------- */
int cpp_batch_create(cpp_ctx* ctx, int max_batch, int64_t state_elems, int action_dim, cpp_batch** out);
int cpp_batch_destroy(cpp_batch* batch);
/* feed_dict of ddpg_cartpole.py:231-237: host Batch -> device. */
int cpp_batch_upload(cpp_batch* batch, int B, const void* state_1, const void* state_2,
                     int state_dtype, const float* action, const float* reward,
                     const float* terminal_mask);
/* np.copy(...) columns of replay_memory.py:134-138: device -> caller-owned host arrays.  States come
 * back in the batch's stored dtype (f16 after cpp_replay_sample). */
int cpp_batch_download(cpp_batch* batch, void* state_1, void* state_2, float* action,
                       float* reward, float* terminal_mask);
int cpp_batch_size(const cpp_batch* batch);
int cpp_batch_state_dtype(const cpp_batch* batch);

/* ---- replay memory payload in HBM (replay_memory.py:11-138) --------------------------------- */
/* Slot allocation / eviction (replay_memory.py:66,84,90,104) stays on the host so the FIFO order is
 * exact; the device holds the f16 state store and mirrors of the five event columns. */
int cpp_replay_create(cpp_ctx* ctx, int buffer_size, int state_slots, int64_t state_elems,
                      int action_dim, cpp_replay** out);
/* The same with a chosen store type: CPP_F16, or CPP_U8 -- 8-bit pixel codes k that read back as f16(k/255), i.e. exactly what
 * the f16 store holds for the reference's renders (bullet_cartpole.py:239-243) in half the HBM.  A CPP_U8 memory refuses states
 * that are not such images (cpp_replay_write_states returns an error). */
int cpp_replay_create_ex(cpp_ctx* ctx, int buffer_size, int state_slots, int64_t state_elems,
                         int action_dim, int store_dtype, cpp_replay** out);
int cpp_replay_destroy(cpp_replay* replay);
/* self.state[idx] = s (replay_memory.py:67,106): n states, f32 is rounded to f16 (RNE) like numpy. */
int cpp_replay_write_states(cpp_replay* replay, const int32_t* slots, int n, const void* states,
                            int state_dtype);
/* rows of the five event columns (replay_memory.py:94-101,105). */
int cpp_replay_write_rows(cpp_replay* replay, const int32_t* rows, int n, const int32_t* state_1_idx,
                          const int32_t* state_2_idx, const float* action, const float* reward,
                          const float* terminal_mask);
/* the same columns read back from the device for n rows (any of the outputs may be NULL) -- replay_memory.py's public attributes
 * state_1_idx / action / reward / terminal_mask / state_2_idx (:22-29) as the sampler sees them */
int cpp_replay_read_rows(cpp_replay* replay, const int32_t* rows, int n, int32_t* s1_idx, int32_t* s2_idx, float* action,
                         float* reward, float* terminal_mask);
/* Keep per-state whitening sums (sum x, sum x^2 per channel, f64) in the store for `channels` interleaved channels: the statistics of
 * base_network.py:95-96 for a sampled minibatch then cost 2 B rows of 2 C doubles instead of a pass over 2 B images.  Results are
 * bit-identical (the same per-image sums, added in the same order).  0 turns them off. */
int cpp_replay_set_stats_channels(cpp_replay* replay, int channels);
int cpp_replay_set_size(cpp_replay* replay, int size);          /* ReplayMemory.size(), :120 */
int cpp_replay_read_states(cpp_replay* replay, const int32_t* slots, int n, void* out_f16);
/* random_indexes + batch (replay_memory.py:123-138) fused: idxs == NULL draws B uniform rows on the
 * device with Philox4x32-10 keyed (seed, counter); otherwise uses the caller's rows (parity tests,
 * numpy-RNG-driven callers).  Also produces the per-channel whitening statistics of both state
 * batches for pixel states of `channels` interleaved channels (channels = 0: none). */
int cpp_replay_sample(cpp_replay* replay, int B, const int32_t* idxs, uint64_t seed, uint64_t counter,
                      int channels, cpp_batch* out);
int cpp_replay_last_indexes(cpp_replay* replay, int B, int32_t* out);   /* rows drawn by last sample */
/* bench/test helper: fill n_rows transitions on the device (SURVEY 8d synthetic inputs):
 * states f16(k/255), k~U{0..255}; a~U(-1,1); reward 1; terminal w.p. 1/50; s2 = s1 slot + 1. */
int cpp_replay_fill_synthetic(cpp_replay* replay, int n_rows, uint64_t seed);

/* ---- DDPG train ops (ddpg_cartpole.py:102-119, :186-248, :329-337) --------------------------- */
typedef struct cpp_ddpg_hyper {
  float actor_learning_rate;     /* --actor-learning-rate  (ddpg_cartpole.py:41)  */
  float critic_learning_rate;    /* --critic-learning-rate (:42)                  */
  float discount;                /* --discount             (:43)                  */
  float gradient_clip;           /* --gradient-clip, util.py:11; <= 0 disables    */
  float target_update_rate;      /* --target-update-rate   (:35)                  */
} cpp_ddpg_hyper;

int cpp_ddpg_create(cpp_ctx* ctx, cpp_net* actor, cpp_net* critic, cpp_net* target_actor,
                    cpp_net* target_critic, const cpp_ddpg_hyper* hyper, cpp_ddpg** out);
int cpp_ddpg_destroy(cpp_ddpg* ddpg);
/* ActorNetwork.train(state) (ddpg_cartpole.py:140-145): uses batch.state_1 only. */
int cpp_ddpg_train_actor(cpp_ddpg* ddpg, cpp_batch* batch);
/* CriticNetwork.train(batch) (:230-237). */
int cpp_ddpg_train_critic(cpp_ddpg* ddpg, cpp_batch* batch);
/* CriticNetwork.check_loss(batch) (:239-248): loss scalar, td (B), q (B). */
int cpp_ddpg_check_loss(cpp_ddpg* ddpg, cpp_batch* batch, float* loss, float* td, float* q);
/* CriticNetwork.q_gradients_wrt_actions() evaluated at a = actor(state_1) (:220-222): (B, A);
 * also returns the actions and q-values of that evaluation when the pointers are non-NULL. */
int cpp_ddpg_q_gradients_wrt_actions(cpp_ddpg* ddpg, cpp_batch* batch, float* dq_da, float* actions,
                                     float* q);
/* Fused form of one loop body of :331-334.  Both gradient sets are taken from the same parameter
 * snapshot (the critic step never reads the live actor, the actor step never writes the critic) and
 * land in ONE flat f32 buffer [actor grads | critic grads] that a data-parallel host all-reduces
 * (RCCL) between the two calls. */
int cpp_ddpg_compute_gradients(cpp_ddpg* ddpg, cpp_batch* batch);
int cpp_ddpg_grad_buffer(cpp_ddpg* ddpg, void** device_ptr, int64_t* n_floats);
/* clip_by_global_norm per list (util.py:47-50) + SGD (ddpg_cartpole.py:118-119,213,218).  The
 * gradients are first multiplied by grad_scale (1/world_size after a sum all-reduce). */
int cpp_ddpg_apply_gradients(cpp_ddpg* ddpg, float grad_scale);
/* target_actor.update_weights(); target_critic.update_weights() (:336-337). */
int cpp_ddpg_update_targets(cpp_ddpg* ddpg);
/* The whole inner step :331-337 on device-resident replay: n_batches x {sample B, both updates},
 * then the target updates.  idxs: NULL (device Philox, counter advances by one per minibatch) or
 * n_batches*B caller-chosen rows.  Captured into a hipGraph after the first call per (B, n_batches)
 * when idxs == NULL and profiling is off. */
int cpp_ddpg_train_step(cpp_ddpg* ddpg, cpp_replay* replay, int B, int n_batches,
                        const int32_t* idxs, uint64_t seed);
/* ddpg_cartpole.py:332-334 for ONE minibatch on B rows the HOST drew (replay_memory.py:123-129, numpy's RNG):
 *     batch = self.replay_memory.batch(batch_size); self.actor.train(batch.state_1); self.critic.train(batch)
 * as the same fused device sequence as one minibatch of cpp_ddpg_train_step -- no gathered copy of the states, no PCIe traffic
 * but the B row indexes -- WITHOUT the target updates (:336-337 stay the caller's: cpp_net_soft_update).  hipGraph-replayed after
 * the first call per (B, replay); returns while the minibatch is still running. */
int cpp_ddpg_train_rows(cpp_ddpg* ddpg, cpp_replay* replay, int B, const int32_t* idxs);
/* Data-parallel learners: the first half of one minibatch of the inner step -- sample B rows on the
 * device (Philox; the counter advances by one) and leave both gradient sets in the flat gradient
 * buffer.  cpp_ddpg_allreduce_grads + cpp_ddpg_apply_gradients(1/N) finish the minibatch
 * (cpp_ddpg_dp_train_step does all three).  hipGraph-captured after the first call per (B, seed, replay). */
int cpp_ddpg_sample_and_compute(cpp_ddpg* ddpg, cpp_replay* replay, int B, uint64_t seed);
/* scalars of the last minibatch: [0] td loss, [1] actor grad norm, [2] critic grad norm (pre-clip). */
int cpp_ddpg_last_stats(cpp_ddpg* ddpg, float out[3]);
/* The per-row values the last minibatch's gradient pass left on the device (whichever entry point ran it: the train ops,
 * cpp_ddpg_compute_gradients, the fused / graph-replayed cpp_ddpg_train_step, cpp_ddpg_sample_and_compute): the actor's
 * actions on state_1 (B, A), dQ/da at those actions (B, A), Q(state_1, fed action) (B) and the temporal difference (B) --
 * what the reference prints under VERBOSE_DEBUG (ddpg_cartpole.py:339-349).  NULL pointers are skipped.  Parity tests read
 * the fused step's values through this call. */
int cpp_ddpg_last_values(cpp_ddpg* ddpg, int B, float* actions, float* dq_da, float* q, float* td);

/* ---- data-parallel actor-learners (the reference's TODO "switch back to async training with multiple replicas",
 * ddpg_cartpole.py:259, naf_cartpole.py:294; its exps only launch independent processes, exps/run_87.sh:12-36) ------------
 * One learner per GPU = one process with one cpp_ctx; replicated weights, an own replay shard and an own minibatch per
 * learner; RCCL collectives over xGMI issued by the library on the context's stream.  Rank 0 makes the id, the host
 * distributes its CPP_COMM_ID_BYTES bytes to the other ranks by any means (bench.py: a torch.distributed / gloo broadcast),
 * every rank then calls cpp_comm_create. */
#define CPP_COMM_ID_BYTES 128
int cpp_comm_unique_id(void* out, int cap);                       /* ncclGetUniqueId */
int cpp_comm_create(cpp_ctx* ctx, const void* unique_id, int rank, int world, cpp_comm** out);   /* ncclCommInitRank */
int cpp_comm_destroy(cpp_comm* comm);
int cpp_comm_info(const cpp_comm* comm, int* rank, int* world);
/* in-place sum (average != 0: mean) over the ranks of n floats at a DEVICE address, on the context's stream */
int cpp_comm_allreduce(cpp_comm* comm, void* device_f32, int64_t n, int average);
/* max over the ranks of one host double / a barrier (bench.py's timed region: max-over-ranks time between two barriers) */
int cpp_comm_max_double(cpp_comm* comm, double* value);
/* element-wise max over the ranks of n (1..8) host doubles.  The agents' --data-parallel loops decide "train this iteration" and
 * "leave the loop" with it once per outer iteration (the per-process tests of ddpg_cartpole.py:329 and :379-383 / naf_cartpole.py:
 * 365,386-389 in front of a collective step would leave the slower ranks blocked in ncclAllReduce). */
int cpp_comm_max_doubles(cpp_comm* comm, double* values, int n);
int cpp_comm_barrier(cpp_comm* comm);
/* sum over the ranks of the flat gradient buffer [actor grads | critic grads] left by cpp_ddpg_sample_and_compute /
 * cpp_ddpg_compute_gradients; cpp_ddpg_apply_gradients(1 / world) then gives every rank the same update. */
int cpp_ddpg_allreduce_grads(cpp_ddpg* ddpg, cpp_comm* comm);
/* periodic mode: the mean over the ranks of all four networks' parameters */
int cpp_ddpg_average_params(cpp_ddpg* ddpg, cpp_comm* comm);
/* This rank's part of the inner step ddpg_cartpole.py:331-337 for N synchronous learners: n_batches x {sample from the own
 * shard + both gradient sets (hipGraph) -> all-reduce -> clip + SGD on the mean}, then the (local) target updates.
 * sync_every = k > 1: k local minibatch updates between parameter averagings instead of a gradient all-reduce per minibatch.
 * overlap != 0: the gradients of the fully connected layers are reduced on a second stream while the conv backward of the
 * same minibatch runs.  comm == NULL: one learner on the same code path. */
int cpp_ddpg_dp_train_step(cpp_ddpg* ddpg, cpp_replay* replay, cpp_comm* comm, int B, int n_batches, uint64_t seed,
                           int sync_every, int overlap);
/* the same for NAF (naf_cartpole.py:367-373): flat buffer [value | mu | l_values]; the averaging includes the optimiser slots */
int cpp_naf_sample_and_compute(cpp_naf* naf, cpp_replay* replay, int B, uint64_t seed);
int cpp_naf_allreduce_grads(cpp_naf* naf, cpp_comm* comm);
int cpp_naf_average_params(cpp_naf* naf, cpp_comm* comm);
int cpp_naf_dp_train_step(cpp_naf* naf, cpp_replay* replay, cpp_comm* comm, int B, int n_batches, uint64_t seed,
                          int sync_every);
/* Which form the default data-parallel step (sync_every 1, no overlap) of this trainer takes -- the reference has no counterpart
 * (ddpg_cartpole.py:259 / naf_cartpole.py:294 are its "TODO: distributed" notes); *mode: 0 = none run yet, 1 = one hipGraph replay
 * per outer step with the all-reduce inside, 2 = the same launches issued on the stream because the runtime or RCCL refused the
 * capture (`reason`, if given, receives what it said). */
int cpp_ddpg_dp_status(const cpp_ddpg* ddpg, int* mode, char* reason, int cap);
int cpp_naf_dp_status(const cpp_naf* naf, int* mode, char* reason, int cap);

/* ---- NAF train ops (naf_cartpole.py:93-284, :365-373) ----------------------------------------- */
typedef struct cpp_naf_hyper {
  float discount;                /* --discount (naf_cartpole.py:46)                               */
  float gradient_clip;           /* --gradient-clip (util.py:11); <= 0 disables                    */
  float target_update_rate;      /* --target-update-rate (:36)                                     */
  int32_t optimiser;             /* --optimiser: CPP_OPT_* (util.py:15, :73-76)                    */
  float learning_rate;           /* --optimiser-args                                               */
  float momentum;                /*   Momentum                                                     */
  float beta1, beta2, epsilon;   /*   Adam (TF defaults .9 / .999 / 1e-8)                          */
} cpp_naf_hyper;

/* NafNetwork.__init__ (:117-245).  value / target_value: CPP_HEAD nets with head_out 1 (ValueNetwork,
 * :93-114).  mu: tanh head with action_dim outputs, l_values: linear head with A(A+1)/2 outputs.  With
 * share != 0 (--share-input-state-representation, :151-152,176-177) mu and l_values must be head-only
 * nets (pixel = 0, n_hidden = 0, state_elems = width of value's last hidden layer) and read value's
 * input_state_representation; otherwise they are full nets with their own trunks on state_1. */
int cpp_naf_create(cpp_ctx* ctx, cpp_net* value, cpp_net* target_value, cpp_net* mu, cpp_net* l_values,
                   int share, const cpp_naf_hyper* hyper, cpp_naf** out);
int cpp_naf_destroy(cpp_naf* naf);
/* NafNetwork.action_given without the noise (:247-253): output_action for a host state batch. */
int cpp_naf_action(cpp_naf* naf, const void* state, int state_dtype, int B, float* out);
/* NafNetwork.train(batch) (:264-272): check_numerics + train_op + loss.  Returns CPP_ERR_NUMERIC when
 * l_values, L or the loss is not finite (tf.check_numerics, :242-245); parameters are then untouched. */
int cpp_naf_train(cpp_naf* naf, cpp_batch* batch, float* loss);
/* NafNetwork.debug_values(batch) (:274-284): l_values (B, A(A+1)/2), loss, value (B), advantage (B),
 * target value (B). */
int cpp_naf_debug_values(cpp_naf* naf, cpp_batch* batch, float* l_values, float* loss, float* value,
                         float* advantage, float* target_value);
/* Gradient halves for data-parallel learners, as for DDPG: flat buffer [value | mu | l_values]. */
int cpp_naf_compute_gradients(cpp_naf* naf, cpp_batch* batch);
int cpp_naf_grad_buffer(cpp_naf* naf, void** device_ptr, int64_t* n_floats);
int cpp_naf_apply_gradients(cpp_naf* naf, float grad_scale);
/* target_value_net.update_weights() (:373). */
int cpp_naf_update_targets(cpp_naf* naf);
/* The whole inner step :367-373 on device-resident replay; hipGraph-captured like cpp_ddpg_train_step.
 * A non-finite minibatch sets a sticky flag that cpp_naf_last_stats reports (out[2] != 0). */
int cpp_naf_train_step(cpp_naf* naf, cpp_replay* replay, int B, int n_batches, const int32_t* idxs,
                       uint64_t seed);
/* naf_cartpole.py:367-371 for ONE minibatch on B rows the HOST drew: `batch = replay_memory.batch(B); loss = naf.train(batch)` with the
 * sample pass reading the replay store through those rows (no gathered copy).  *loss = the minibatch's loss; CPP_ERR_NUMERIC (the
 * optimiser does not run) when l_values, L or the loss is not finite, like cpp_naf_train.  No target update (:373 stays the caller's). */
int cpp_naf_train_rows(cpp_naf* naf, cpp_replay* replay, int B, const int32_t* idxs, float* loss);
/* The same minibatch without waiting for it: gradients and optimiser are enqueued (the optimiser kernel stands down by itself when
 * the check_numerics flag is set), *ticket names the call.  cpp_naf_loss_wait(ticket, &loss) waits for THAT minibatch and returns
 * its loss, or CPP_ERR_NUMERIC; a ticket stays readable until CPP_NAF_TICKETS later calls have been made.  The reference's
 * `losses.append(naf.train(batch))` (naf_cartpole.py:369-371) only ever averages the losses for its STATS line. */
#define CPP_NAF_TICKETS 8
int cpp_naf_train_rows_async(cpp_naf* naf, cpp_replay* replay, int B, const int32_t* idxs, uint64_t* ticket);
int cpp_naf_loss_wait(cpp_naf* naf, uint64_t ticket, float* loss);
/* [0] loss of the last minibatch, [1] pre-clip global gradient norm, [2] non-finite flag (sticky). */
int cpp_naf_last_stats(cpp_naf* naf, float out[3]);
/* The optimiser's slot variables, which tf.train.Saver checkpoints with everything else (util.py:88-90): Momentum accumulators
 * / Adam first moments `m` and Adam second moments `v`, each n = params(value) + params(mu) + params(l_values) floats in the
 * flat order [value | mu | l_values], and the number of applied updates `step` (Adam's beta powers are beta^step).  NULL
 * pointers are skipped. */
int64_t cpp_naf_opt_state_size(const cpp_naf* naf);
int cpp_naf_get_opt_state(cpp_naf* naf, float* m, float* v, int64_t n, uint64_t* step);
int cpp_naf_set_opt_state(cpp_naf* naf, const float* m, const float* v, int64_t n, uint64_t step);
/* cpp_naf_train_rows_async keeps the check_numerics flag of a non-finite minibatch set: every later update stands down, as the
 * reference's run ends there (naf_cartpole.py:242-245,265).  cpp_naf_set_opt_state (a restored checkpoint) clears it; so does this
 * call, for a caller that has put good parameters back with cpp_net_set_params. */
int cpp_naf_clear_numeric_error(cpp_naf* naf);

#ifdef __cplusplus
}
#endif
#endif /* CARTPOLEPP_ABI_H */
