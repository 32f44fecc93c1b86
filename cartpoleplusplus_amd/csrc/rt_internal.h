// Internal declarations shared by the rt_*.cpp translation units (the C-ABI implementation of include/cartpolepp_abi.h):
// handle structs, the device-memory arena, the launch-sequence helpers of the networks and the level-synchronous
// launch scheduler.  Host-side logic only; all arithmetic is in the HIP kernels.
#pragma once
#include "../../include/cartpolepp_abi.h"
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include <algorithm>

#include <rccl/rccl.h>

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
    cpp_arena_register(raw, bytes + 2 * GUARD);
    *p = (char*)raw + GUARD;
    if (zero) HIP_CHECK(hipMemsetAsync(raw, 0, bytes + 2 * GUARD, stream));
    return 0;
  }
  void release() { for (void* p : ptrs) { cpp_arena_unregister(p); (void)hipFree(p); } ptrs.clear(); }
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
  float* dpool_imax = nullptr;             // [maxB][DX_IMAX_SLOTS]: bounds of |dpool[0]| per image, left by conv2's dX (conv_dx_rs.h) for conv1's dW
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
  void* wimg;               // conv1's operand image on the f16 pipes (conv_rs16.h)
  const float* wimg_key;    // the whitening table (scale pointer) the optimiser's launch built wimg for, with the weights it left; nullptr: stale --
                            // the next conv1 forward builds it itself.  Cleared by everything that changes the parameters.
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

#define CPP_ROWS_RING 8
#define CPP_ROWS_RING_SLOT 4096      // ints per slot (larger draws take the synchronous copy)
struct cpp_replay {
  cpp_ctx* ctx; int rows, slots, A, size; long elems;
  int store_dtype;         // CPP_F16 (replay_memory.py:32) or CPP_U8 (pixel codes k, read back as f16(k/255): half the HBM)
  void* store; int32_t *s1, *s2, *rows_in, *rows_out; float *action, *reward, *mask;
  uint64_t* counter;       // device-side Philox counter of the train steps (graph replay)
  uint64_t* counter_adhoc; // the same for cpp_replay_sample(idxs == NULL): inspection draws never move the training sampler
  int32_t* size_dev;       // rows currently in the memory, on the device: the sampler's range of captured launches
  uint64_t uid;            // unique per cpp_replay_create (graph keys: an address can be reused, this cannot)
  bool sampled;            // a sample pass has been built on this memory since the uid was issued: captured step graphs may hold its slot_stats pointer
  uint64_t write_gen;      // bumped by every call that changes rows, states or the size: a minibatch presampled before it is stale
  __half* lut; int* bad; uint16_t lut_host[256];      // CPP_U8: f16(k/255) table, "not a pixel image" flag
  // per-state whitening sums (cpp_replay_set_stats_channels): [slots][2 * stats_C] doubles, kept current by every call that writes states
  double* slot_stats; int stats_C, stats_cap; int32_t* slot_list; size_t slot_list_cap;
  void* stage; size_t stage_cap;                      // device staging of incoming states (conversion source)
  void* pinned; size_t pinned_cap; hipEvent_t pinned_free; bool pinned_busy;   // host staging: writes return before the copy ends
  // host-drawn minibatch rows on their way to rows_in (cpp_ddpg_train_rows / cpp_naf_train_rows): a ring of pinned slots, so that the
  // call returns while the previous minibatch is still running (a pageable hipMemcpyAsync would wait for the stream)
  int32_t* rows_pin; hipEvent_t rows_pin_ev[CPP_ROWS_RING]; bool rows_pin_used[CPP_ROWS_RING]; int rows_pin_k;
  Arena arena;
};
static size_t replay_esz(const cpp_replay* r) { return r->store_dtype == CPP_U8 ? 1 : sizeof(__half); }

constexpr int NORM_PARTS = 64;

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
  // skip >= 0: that op is left out (the caller runs it on its own later -- the data-parallel step's split at the conv backward)
  int run(cpp_ctx* ctx, int skip = -1) {
    size_t remaining = ops.size();
    if (skip >= 0 && skip < (int)ops.size() && !ops[skip].done) { ops[skip].done = true; --remaining; }
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
      static const bool dbg = cpp_switch_set("CPP_OPGRAPH_DEBUG");
      if (dbg) {
        fprintf(stderr, "[opgraph] level:");
        for (int i : ready) {
          if (ops[i].is_gemm) fprintf(stderr, " gemm#%d(M%d N%d K%d e%d)", i, ops[i].g.M, ops[i].g.N, ops[i].g.K, ops[i].g.epi);
          else fprintf(stderr, " fn#%d", i);
        }
        fprintf(stderr, "\n");
      }
      for (int i : ready) if (!ops[i].is_gemm) RC(ops[i].fn());
      for (int i : ready) ops[i].done = true;
      remaining -= ready.size();
      for (int i : ready) if (ops[i].is_gemm) batch.push_back(ops[i].g);
      if (!batch.empty()) RC(launch_gemm_batch(ctx, batch.data(), (int)batch.size()));
    }
    return CPP_OK;
  }
};


// kernel ids of conv layer i's forward / dW / dX launches (profile rows)
static const int kFwdKid[3] = {K_CONV1_FWD, K_CONV2_FWD, K_CONV3_FWD};
static const int kDwKid[3] = {K_CONV1_DW, K_CONV2_DW, K_CONV3_DW};
static const int kDxKid[3] = {-1, K_CONV2_DX, K_CONV3_DX};

// ---- launch-sequence helpers shared by the translation units (definitions: rt_net.cpp, rt_replay.cpp, rt_ddpg.cpp)
int gemm(cpp_ctx* ctx, const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc, int M, int N, int K, int epi, const float* Y = nullptr, long ldy = 0, int accumulate = 0);
ConvArgs conv_fwd_args(cpp_net* n, Workspace& w, int i, const void* state, int dtype, const float* white, int B, int* mode, long white_bstride = 0);
void conv_dy_desc(cpp_net* n, Workspace& w, int i, ConvArgs& a, int B);
ConvArgs conv_dw_args(cpp_net* n, Workspace& w, int i, const void* state, int dtype, const float* white, int B, int* mode);
ConvArgs conv_dx_args(cpp_net* n, Workspace& w, int i, int B);
BnNet bn_net_desc(cpp_net* n, Workspace& w, int i);
BnBatch bn_batch(cpp_net* const* nets, int nn, int i, int B);
bool trunk_b16(const cpp_net* n, int dtype, int B, long white_bstride);
int net_forward_trunk(cpp_net* n, Workspace& w, const void* state, int dtype, const float* white, int B, long white_bstride = 0);
int nets_forward_trunk_fused(cpp_ctx* ctx, cpp_net* const* nets, int nn, const void* const* sts, const float* const* whs, int first_target, int dt, int B);
int nets_forward_trunk_bn(cpp_ctx* ctx, cpp_net* const* nets, int nn, const void* const* states, const float* const* whites, int dtype, int B);
GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc, int M, int N, int K, int epi);
GemmArgs mk_gemm(const float* A, long sAm, long sAk, const float* Bm, long sBk, long sBn, float* C, long ldc, int M, int N, int K, int epi, const float* Y, long ldy);
void set_dropout(GemmArgs& g, cpp_net* n, int l);
int relu_grad_epi(const cpp_net* n, int producer_layer);
int bump_dropout(cpp_net* n);
int net_forward_fc(cpp_net* n, Workspace& w, int from, int B, const float* action);
int net_backward_conv(cpp_net* n, Workspace& w, int B, const void* state, int dtype, const float* white);
int nets_backward_conv(cpp_ctx* ctx, cpp_net* const* nets, int nn, int B, const void* state, int dtype, const float* white);
int net_backward(cpp_net* n, Workspace& w, int B, bool want_params, float* d_action, const void* state, int dtype, const float* white, int start_layer = -2);
GemmArgs fc_fwd_args(cpp_net* n, Workspace& w, int l, int B);
GemmArgs fc_dw_args(cpp_net* n, Workspace& w, int l, int B, const float* dz);
GemmArgs fc_dx_args(cpp_net* n, int l, int B, const float* dz, long dz_ld, int col0, int ncols, float* C, long ldc, int epi, const float* Y, long ldy);
int batch_stats(cpp_ctx* ctx, const void* s0, const void* s1, int dtype, long elems, int B, int C, double* part, float* white);
int batch_ensure_stats(cpp_batch* b, int C);
uint64_t replay_next_uid();      // graph keys: a fresh uid per cpp_replay_create and per change of a sampled memory's statistics setting
GatherArgs replay_gather_args(cpp_replay* r, int B, const int32_t* rows_dev, uint64_t seed, const uint64_t* counter_dev, int channels, cpp_batch* out, bool direct, int* C_out);
int replay_sample_finish(cpp_replay* r, int B, int C, int channels, cpp_batch* out, uint64_t* bump = nullptr, bool* bumped = nullptr);
int replay_sample_device(cpp_replay* r, int B, const int32_t* rows_dev, uint64_t seed, const uint64_t* counter_dev, int channels, cpp_batch* out, bool direct = false,
                         uint64_t* bump = nullptr, bool* bumped = nullptr);
int replay_stage_rows(cpp_replay* r, const int32_t* idxs, int n, const char* who);
const float* white_of(cpp_batch* b, int which, int C);
bool direct_replay_ok(cpp_net* a, cpp_replay* r, int B);

// ---- communicator of the data-parallel learners (rt_comm.cpp): one rank per cpp_ctx, RCCL over xGMI
struct cpp_comm {
  cpp_ctx* ctx; ncclComm_t comm; int rank, world;
  uint64_t uid;                // unique per cpp_comm_create (graph keys: an address can be reused by the allocator, this cannot)
  hipStream_t side;            // second stream: collectives that overlap the conv backward of the same minibatch
  hipEvent_t ev_fc, ev_bwd, ev_done;
  double* scratch;             // CPP_COMM_SCRATCH_WORDS device doubles of cpp_comm_max_doubles (allocated on first use)
};
#define CPP_COMM_SCRATCH_WORDS 8
#define NCCL_CHECK(expr)                                                                   \
  do {                                                                                     \
    ncclResult_t _r = (expr);                                                              \
    if (_r != ncclSuccess) {                                                               \
      cpp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); \
      return CPP_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)
