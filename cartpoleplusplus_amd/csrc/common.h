// Shared declarations of libcartpolepp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Run-time kernel-selection switches (CPP_CONV_K16=0, CPP_FUSED_HEADS=0, ...) exist only in the ablation build
// (-DCPP_ABLATION -> libcartpolepp_hip_ablation.so: A/B measurements and the parity tests of the fallback kernels).
// The release library reads no environment variable: every switch below is a compile-time `false`.
#ifdef CPP_ABLATION
#include <cstdlib>
inline bool cpp_switch_off(const char* name) { const char* v = getenv(name); return v != nullptr && atoi(v) == 0; }
inline bool cpp_switch_set(const char* name) { return getenv(name) != nullptr; }
inline int cpp_switch_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr bool cpp_switch_off(const char*) { return false; }
constexpr bool cpp_switch_set(const char*) { return false; }
constexpr int cpp_switch_int(const char*, int dflt) { return dflt; }
#endif

// Two kernel bodies in one grid (conv2_bwd_pair.hip, conv3_bwd_pair.hip): which body workgroup i of (na + nb) runs.  The dispatcher
// hands out workgroups in index order, so the order decides who starts first: order 0 = all of a, then b; 1 = all of b, then a;
// 2 = b spread evenly through a.  Returns true for body b; *idx = the workgroup's index within its own body's grid.
__host__ __device__ inline bool pair_grid_place(int i, int na, int nb, int order, int* idx) {
  if (order == 1) { if (i < nb) { *idx = i; return true; } *idx = i - nb; return false; }
  if (order == 2 && nb > 0 && na > 0) {
    const int k = (na + nb) / nb;                       // every k-th workgroup of the first nb * k is one of b
    const int head = nb * k;
    if (i < head) {
      if (i % k == 0) { *idx = i / k; return true; }
      *idx = i - i / k - 1; return false;
    }
    *idx = i - nb; return false;
  }
  if (i < na) { *idx = i; return false; }
  *idx = i - na; return true;
}

#define CPP_NOUT_MAX 16       // one MFMA N tile; the reference hard-codes 10 filters (base_network.py:103)
#define CPP_MAX_CHANNELS 64

// kernel ids for the per-kernel HIP-event profile (cpp_prof_*)
enum KernelId {
  K_GATHER_STATS = 0, K_STATS_FINALIZE, K_STATS_GENERIC,
  K_CONV1_FWD, K_CONV2_FWD, K_CONV3_FWD,
  K_CONV1_DW, K_CONV2_DW, K_CONV3_DW,
  K_CONV2_DX, K_CONV3_DX,
  K_DW_REDUCE, K_GEMM, K_ELEMENTWISE, K_TD, K_SUMSQ, K_CLIP_SGD, K_SOFT_UPDATE,
  K_REPLAY_FILL, K_NAF_HEAD,
  K_CONV1_FWD_F16X3,      // conv1 forward on the f16 pipes with three-piece weights (conv_k16.h)
  K_CONV1_DW_F16X3,       // conv1 dW on the f16 pipes with three-piece dY (conv_dw16.h)
  K_HEADS,                // fused DDPG heads (heads.hip)
  K_CONV3_BWD,            // conv3's dW and dX in one launch (conv3_bwd_pair.hip)
  K_CONV2_BWD,            // conv2's dW and dX in one launch (conv2_bwd_pair.hip)
  K_REDUCE_GATHER,        // the dW reductions of a minibatch + sample / statistics of the next one in one launch (replay.hip)
  K_CONV1_DW_GATHER,      // conv1 dW (f16 pipes) + sample / statistics of the next minibatch in one launch (conv1_dw_gather.hip)
  K_ALLREDUCE,            // the data-parallel step's gradient all-reduce (RCCL), as the stream sees it between the gradient kernels and the update
  K_CONV1_IMAGE,          // conv1's operand images as a launch of their own (conv_rs16.h; in the fused step they ride in the optimiser's launch)
  K_NUM_KERNELS
};

// deferred second-stage reductions of the conv dW partials (flushed in one launch)
struct DwReduceDesc { const float* partial; int nblocks, pstride, nw, nout; float* grad_w; float* grad_b;
                      double* sq_part; };      // non-null: reduction block b also leaves the sum of squares of its 64 gradients in sq_part[b]
#define DW_REDUCE_MAX 8

#ifndef CPP_PRECISION_FAST
#define CPP_PRECISION_FAST 0      /* (include/cartpolepp_abi.h) */
#define CPP_PRECISION_EXACT 1
#endif
// The whitening tables of the NEXT minibatch as a rider of the dW reductions' launch (conv_dw_reduce_kernel): its sample pass has left with
// conv1's dW (conv1_dw_gather.hip), so its per-row sums are complete when the reductions start -- one wave per (state column, channel)
struct StatsRide { const double* part; float* white; int nparts, jobs, C; double count, eps; unsigned* wmax; };
struct cpp_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  hipEvent_t t0, t1;
  // profiling
  bool prof;
  double prof_ms[K_NUM_KERNELS];
  int64_t prof_n[K_NUM_KERNELS];
  hipEvent_t pe0, pe1;
  int num_cus;
  DwReduceDesc pending[DW_REDUCE_MAX];
  int npending;
  struct ConvPairSlot* pair;      // non-null: conv3's dW / dX launchers park their launch here instead of launching
  const struct GatherArgs* ride;  // non-null: the next minibatch's sample + statistics kernel rides in the dW reduction's launch
  bool ride_done; int ride_dtype;
  bool ride_at_dw;                // the rider may already leave with conv1's dW (its slots are double-buffered: direct replay)
  const StatsRide* st_ride; bool st_ride_done;      // non-null: flush_dw_reduce may finish that pass's statistics (sets st_ride_done)
  // Global-norm partials folded into the kernels that write the gradients (fused single-learner step): every dW GEMM tile and every
  // conv dW reduction block leaves the f64 sum of squares of its outputs in sq_part (one region of SQ_REGION slots per gradient
  // list); the optimiser kernel adds a list's partials in slot order.  sq_n: slots handed out so far, -1: folding off (the sumsq
  // kernel runs instead: data-parallel steps, whose gradients change in the all-reduce; batch norm; NAF).
  double* sq_part; int sq_n[2]; int sq_conv_group[4];
  int precision;                  // CPP_PRECISION_FAST / CPP_PRECISION_EXACT (cpp_ctx_set_precision; conv_k16.h)
  // Nearly constant channels (a blind camera with a rare off-colour pixel: whitening scale ~ 10^3 on values that do not cancel exactly):
  // the f16-pipe conv1 kernels multiply RAW pixels and cancel inside the MFMA, which on such a table sits ~4 x further from float64 than
  // whitening each element first, as the f32-input kernels (and the reference's TF graph, base_network.py:95-99) do.  The statistics
  // kernels keep the largest scale of a training call in white_max_dev; the call's last launch publishes it -- tagged with the call's
  // number -- to pinned host memory and resets it; above route_threshold conv1 (forward and dW) and conv2 forward run on the f32-input
  // kernels until the scale is back under half of it.  kernel_epoch: bumped by every flip -- the trainers' captured graphs are keyed on it.
  // WHICH call's scale decides is a function of the program's order alone, never of host / GPU timing (round 6; rt_core.cpp:
  // ctx_route_update): call k reads the scale of call k - 2 -- behind an event that guarantees it has landed -- or of call k - 1 if the
  // caller has synchronised the stream since.  Two runs of the same program take the same routes at the same steps.
  unsigned* white_max_dev; bool conv1_f32; uint64_t kernel_epoch; float route_threshold;
  unsigned long long* route_pin;       // pinned: slot t & 1 = (tag t << 32 | float bits of the largest scale of call t)
  unsigned long long* route_pin_dev;   // ... its device address: the call's closing soft-update kernel (route_rider) or route_publish_kernel writes it
  unsigned* route_tag_dev;             // the number of the call that is running (written in stream order at every training entry point)
  bool route_tag_signal;               // ... by hipStreamWriteValue32 (signal memory) instead of a one-dword fill kernel
  hipEvent_t route_ev[2];              // recorded at the entry of call k (slot k & 1): everything before call k has finished when it fires
  uint64_t route_calls, route_done, route_min_tag;      // calls entered; calls known complete (a stream synchronisation since); tags below are ignored (threshold changed)
  float route_last_max;                // the last scale a decision or cpp_ctx_get_route read
  bool route_rider;               // set by a step body in front of its target update: that launch also publishes and resets the scale word
  int n_trainers;                 // live cpp_ddpg / cpp_naf objects: their captured graphs pin the precision mode
};
#define SQ_REGION 2048

// ---------------------------------------------------------------------------------------------
// conv kernels (conv.hip)
// ---------------------------------------------------------------------------------------------
enum ConvInMode { IN_F16_WHITEN = 0, IN_F32_WHITEN = 1, IN_F32_PLAIN = 2, IN_DY = 3, IN_F32_FLIP = 4 };
enum ConvEpi { EPI_RELU_POOL = 0, EPI_PLAIN = 1 };

// (the arg-max code byte of a pooled cell: bits 0-1 = window position dy * 2 + dx of the first maximum, bit 2 = the pooled output is > 0)
constexpr int POOL_ACTIVE = 4;
// floats per image of the bounds a dX launch leaves for the next layer's dW (conv_dx_rs.h: [4 tiles of a row][2 bands of rows])
#define DX_IMAX_SLOTS 8
// How to rebuild the gradient w.r.t. a conv's pre-activation output from the pooled-resolution
// gradient: dY[b,y,x,o] = dpool[b,y/2,x/2,o] if amax == (POOL_ACTIVE | (y&1)*2+(x&1)) else 0  (`pool` is no longer read by any backward kernel).
struct DyDesc {
  const float* dpool; const float* pool; const uint8_t* amax;
  const float* imax;         // optional: per image DX_IMAX_SLOTS floats whose maximum bounds |dpool| of that image (left by the kernel that wrote dpool:
                             // conv_dx_rs.h, ConvArgs::dx_imax) -- conv_dw16_rs.h then takes its 2^S from them instead of scanning the rows
  long dpool_bstride, pool_bstride;       // elements between images
  int Hp, Wp;
};

struct ConvArgs {
  const void* in;            // f16/f32 images, or f32 activations (IN_F32_PLAIN)
  long in_bstride;
  const float* scale; const float* shift;   // whitening (IN_*_WHITEN)
  long white_bstride;        // floats between the (scale, shift) tables of consecutive images; 0: one table for the batch
  float wscale;              // weights are multiplied by this when loaded (batch norm, inference mode); 0 means 1
  int flip;                  // plain f32 input rows, but flipped / transposed weights (dX from a dense dY)
  const float* dy_dense; long dy_dense_bstride;   // dW kernels: dense dY rows (batch norm) instead of the pooled DyDesc
  DyDesc dy;                 // IN_DY (forward kernel in "dX" mode) / dW kernel B operand
  const float* w;            // HWIO weights of the layer
  const float* bias;
  float* out; long out_bstride; uint8_t* out_amax;   // EPI_RELU_POOL: pooled + argmax code
  float* partial;            // dW kernel: per-block partial sums [grid][pstride]
  int pstride;
  int B, H, W;               // conv spatial dims (SAME: input == output)
  int cin_rt;                // runtime copy of CIN (checked against the template)
  int nout;                  // valid output channels of this kernel (<= 16)
  int tiles_x, tiles_y, ntiles;
  int vec_ok;                // input rows may be staged with aligned 16-byte loads
  int nbands, band_rows;     // conv_fwd_kyo_kernel: bands of output rows per image (0 / 1: whole images)
  const unsigned short* in_b16; long plane_stride;   // conv_fwd_k16_kernel in B16 mode: the input as three bf16 planes (stride in bytes; in_bstride: halves per image)
  unsigned short* out_b16; long out_b16_plane;   // conv_fwd_k16_kernel: also write the pooled output as three bf16 planes (plane stride in halves)
  const int32_t* img_slot;   // conv1 on the f16 pipes only: image b is row img_slot[b] of `in` (the replay store itself: no gathered copy)
  // conv2 forward on the bf16 pipes only (n3_w != nullptr): the workgroup also runs conv3 + pool3 of its two images from the
  // pooled rows it has just produced (kept in LDS as zero-haloed 16x16 images: conv3_img.h) -- conv3's launch disappears
  const float* n3_w; const float* n3_bias; float* n3_out; long n3_out_bstride; uint8_t* n3_amax;
  float* dx_imax;            // dX launches (conv_dx_rs.h): per image [tile][band] slots for the largest |value| the wave stored (nullptr: not wanted)
  const void* wimg;          // conv1 on the f16 pipes: the network's operand image buffer (conv_rs16.h: conv1_image_kernel), or nullptr
  const float* wimg_key;     // == scale: the image in wimg was built by the optimiser's launch for this very table and these weights
};

// Same-geometry convolutions of several networks in ONE launch (blockIdx.y selects the descriptor): the
// narrow conv2/conv3 layers and their backward kernels do not fill the chip on their own.
#define CONV_BATCH_MAX 4
struct ConvArgsN { ConvArgs a[CONV_BATCH_MAX]; int n; };

// true if conv1 forward (plain: batch norm) and dW (dense dY: batch norm) of this geometry run on conv_k16.h / conv_dw16.h
bool conv1_f16_pipes_ok(const cpp_ctx* ctx, int cin, int H, int W, int B, bool batch_norm);
// true if conv1 forward runs on conv_k16.h (and can leave bf16 planes of its output) and conv2 forward has a B16 instance
bool conv12_b16_ok(const cpp_ctx* ctx, int cin, int H, int W, int B);
int launch_conv_fwd(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, int epi, ConvArgs a);
int launch_conv_fwd_multi(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, int epi, const ConvArgs* list, int n);
// conv3 forward of 16x16 inputs with the whole image in LDS (conv3_img.hip)
// conv3's dW and dX are independent, latency-bound launches of ~12 us each: parked by their launchers and sent as ONE grid
struct ConvPairSlot { int layer; bool have_dw, have_dx; ConvArgsN dw, dx; int dw_gx, dx_gx; size_t dw_lds, dx_lds; int upi, band;
                      bool dx_rs, dw_rs; };      // dx_rs: the dX half is conv_dx_rs.h's body (bf16 pipes), not conv_kyo.h's; dw_rs: the dW half conv_dw_rs.h's
int launch_conv3_bwd_pair(cpp_ctx* ctx, const ConvPairSlot& slot);
int launch_conv2_bwd_pair(cpp_ctx* ctx, const ConvPairSlot& slot);
bool conv3_img_ok(int cin, int ks, int H, int W, int nout);
// true if conv2 forward of this geometry runs on conv_k16.h's B16 instance that can carry conv3 as its tail
bool conv23_fuse_ok(int H2, int W2, int B, int nout);
int launch_conv3_img(cpp_ctx* ctx, const struct ConvArgsN& batch);
int launch_conv_dw_multi(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, const ConvArgs* list, int n,
                         float* const* grad_w, float* const* grad_b);
int launch_conv_dw(cpp_ctx* ctx, int kid, int cin, int ks, int in_mode, ConvArgs a, float* grad_w,
                   float* grad_b);
size_t conv_dw_partial_floats(cpp_ctx* ctx, int cin, int ks, int nout);
// conv2's dX on the bf16 pipes (conv_dx_rs.h)
int conv_dx_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, bool* handled);
bool conv_dx_rs_ok(const cpp_ctx* ctx, int cin, int ks, int H, int W, int nout);      // would a layer's dX run on conv_dx_rs.h?
int conv_dw_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, int* grid, bool* handled);
// conv2 / conv3 forward on the bf16 pipes, row-streaming (conv_fw_rs.h)
int conv_fw_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, int epi, const ConvArgsN& a, bool* handled);
// conv3's backward at 16-wide rows (64x64 images): both halves on the row-streaming bodies, in conv3_bwd_pair.hip's one launch
bool conv3_pair_rs_ok(const cpp_ctx* ctx, int H, int W);
size_t conv_rs16_image_bytes();     // a network's conv1 operand image (conv_rs16.h)
bool conv_rs16_ok(cpp_ctx* ctx, int cin, int H, int W, int nout);
inline bool conv_rs16_channels_ok(int cin) { return cin == 3 || cin == 6 || cin == 9 || cin == 12 || cin == 18; }      // conv_fwd_rs16.hip's instances (15: two copies of three-chunk rows spill)
int flush_dw_reduce(cpp_ctx* ctx);     // one launch for every dW reduction queued by launch_conv_dw
// A backward pass that fails half way (a geometry without a kernel, a launch error) must not leave its queued reductions behind:
// they point into that network's buffers, which may be gone by the time the next pass flushes the queue.
struct DwPendingGuard {
  cpp_ctx* ctx; bool armed;
  explicit DwPendingGuard(cpp_ctx* c) : ctx(c), armed(true) {}
  void keep() { armed = false; }         // (the queue is handed on on purpose: the split half step's first phase)
  ~DwPendingGuard() { if (armed) ctx->npending = 0; }
};

// Philox4x32-10 (Salmon et al., SC'11): the replay sampler's and the dropout masks' counter-based generator
struct u32x4 { uint32_t x, y, z, w; };
__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// ---------------------------------------------------------------------------------------------
// gemm + elementwise (gemm.hip)
// ---------------------------------------------------------------------------------------------
enum GemmEpi { GE_NONE = 0, GE_RELU = 1, GE_TANH = 2, GE_MUL_RELU_GRAD = 3, GE_MUL_TANH_GRAD = 4,
               GE_ACTOR_HEAD = 5,     // C = v (dQ/da), C2 = -v * (1 - Y^2): grad_ys of ddpg_cartpole.py:111-113 through tanh
               // --use-dropout (base_network.py:69-70: slim.dropout, keep 0.5, after the ReLU of a hidden layer):
               GE_RELU_DROPOUT = 6,   // relu, then keep ? 2 v : 0 with the keep bit from Philox(seed; element, layer, *counter);
                                      // without a counter (inference: IS_TRAINING False) plain relu
               GE_MUL_RELU_GRAD_X2 = 7 };   // backward through such a layer: Y > 0 <=> kept and active -> 2 v
struct GemmArgs {
  const float* A; long sAm, sAk;
  const float* B; long sBk, sBn;
  float* C; long ldc;
  const float* Y; long ldy;     // activation output for the *_GRAD epilogues
  int M, N, K, epi;
  int accumulate;               // C = C + A*B before the epilogue (sums several heads' d(representation))
  float* C2; long ldc2;         // optional second copy of the output (e.g. actions straight into the critic's input)
  const uint64_t* drop_counter; uint32_t drop_seed, drop_layer;   // GE_RELU_DROPOUT
  double* sq_part;              // non-null: tile t also leaves the f64 sum of squares of its outputs in sq_part[t] (see cpp_ctx::sq_part)
};
#define GEMM_BATCH_MAX 16
struct GemmBatch { GemmArgs g[GEMM_BATCH_MAX]; int tile_start[GEMM_BATCH_MAX + 1]; int n; };
// kernel attributes (dynamic LDS size) are set once per kernel AND device: the launchers keep one flag per device slot
#define CPP_MAX_DEVICES 16
static inline int cpp_dev_slot(const cpp_ctx* ctx) { return ctx->device >= 0 && ctx->device < CPP_MAX_DEVICES ? ctx->device : 0; }
// sub-tiles per workgroup edge (gemm.hip: 2 x 2 tiles of 16 x 16 where K is long enough for operand traffic to bound the level)
#ifndef GEMM_SUB_MIN_K
#define GEMM_SUB_MIN_K 100000      // (2 x 2 measured slower: 0.0565 vs 0.0433 ms for the four levels -- a quarter of the workgroups, each four times as long)
#endif
static inline __host__ __device__ int gemm_sub(int M, int N, int K) { return (K >= GEMM_SUB_MIN_K && M > 16 && N > 16) ? 2 : 1; }
static inline int gemm_tiles(int M, int N, int K) { const int e = 16 * gemm_sub(M, N, K); return ((M + e - 1) / e) * ((N + e - 1) / e); }
int launch_gemm(cpp_ctx* ctx, const GemmArgs& g);
int launch_gemm_batch(cpp_ctx* ctx, const GemmArgs* list, int n);   // independent GEMMs in one launch
int launch_copy_cols(cpp_ctx* ctx, float* dst, long ldd, int dcol0, const float* src, long lds_,
                     int scol0, int ncols, int rows);
int launch_fill(cpp_ctx* ctx, float* dst, long ld, int col0, int ncols, int rows, float v);
int launch_state_to_f32(cpp_ctx* ctx, float* dst, long ldd, const void* src, int dtype, long elems,
                        int rows);
int launch_actor_head_grad(cpp_ctx* ctx, float* dz, const float* dq_da, const float* act, int n);
int launch_td(cpp_ctx* ctx, const float* q, const float* tq, const float* r, const float* mask,
              float discount, int B, float* td, float* dq, float* loss);

// ---------------------------------------------------------------------------------------------
// replay + whitening statistics (replay.hip)
// ---------------------------------------------------------------------------------------------
struct GatherArgs {
  const void* store[2];         // per state column: [slots, elems] store (or the batch itself when s_idx == nullptr)
  const int32_t* s_idx[2];      // state_1_idx / state_2_idx columns (nullptr: identity rows)
  const int32_t* rows;          // caller-provided rows, or nullptr -> philox
  int32_t* rows_out;            // rows actually used
  const float* action; const float* reward; const float* mask;
  void* out_state[2];           // gathered states (nullptr: no copy, statistics only)
  int32_t* out_slot[2];         // store row of every sampled state (nullptr: not wanted): conv1 can read the store itself
  float* out_action; float* out_reward; float* out_mask;
  double* part;                 // [2][B][2*C] per-row partial sums
  const double* slot_stats;     // non-null (and no copy wanted): [slots][2*C] sums the store keeps per state (cpp_replay_set_stats_channels) --
                                // a sampled row's partial is COPIED from there instead of re-reading its pixels
  uint64_t seed; const uint64_t* counter;
  int counter_add;              // the draw is keyed by *counter + counter_add (a gather that runs before the step that bumps the counter)
  const __half* lut;            // u8 store: f16(k / 255) for the 256 pixel codes
  long elems; int B; int size; int action_dim; int C;
  const int32_t* size_ptr;      // non-null: the sampler's range is read from here (device word kept by cpp_replay_set_size), so
                                // that a captured launch keeps sampling the whole memory as it grows; `size` otherwise
};
int launch_gather_stats(cpp_ctx* ctx, const GatherArgs& a, int dtype);
// per-state sufficient statistics of store rows [first, first + n) (slots == nullptr) or of the n rows listed in `slots` (device): the very
// sums a gather of that state would leave in GatherArgs::part, computed once when the state is written
int launch_slot_stats(cpp_ctx* ctx, const void* store, int dtype, long elems, int C, double* slot_stats, const int32_t* slots, int first, int n,
                      const __half* lut);
struct DwReduceBatch;
int launch_reduce_gather(cpp_ctx* ctx, const DwReduceBatch& rb, const GatherArgs& a, int dtype);      // f16 / u8 store; replay.hip   // dtype of the store: CPP_F32 / CPP_F16 / CPP_U8 (gathers to f16)
int launch_u8_to_f16(cpp_ctx* ctx, __half* dst, const uint8_t* src, long n, const __half* lut);
int launch_to_u8(cpp_ctx* ctx, uint8_t* dst, const void* src, int src_dtype, long n, const __half* lut, int* bad);
int launch_replay_fill_u8(cpp_ctx* ctx, uint8_t* store, long total, uint64_t seed);
// batch norm (bn.hip): up to four same-shaped networks per launch
struct BnNet {
  float* z;                    // (B, H, W, C) plain conv output; overwritten by dz in the backward pass
  float* stat;                 // [2][C]: inv, -mean * inv
  const float* beta;
  float* pool; long pool_bstride; uint8_t* amax;
  const float* dpool; long dpool_bstride;
  double* part;                // [BN_BLOCKS][2][C] reduction partials
  float* means;                // [2][C]: mean dy, mean dy*zhat
  float* dbeta;
};
struct BnBatch { BnNet n[CONV_BATCH_MAX]; int count; int B, H, W, C; };
size_t bn_part_doubles(int C);
int launch_bn_forward(cpp_ctx* ctx, const BnBatch& bb, double eps);
int launch_bn_backward(cpp_ctx* ctx, const BnBatch& bb);
// bump: a device counter the kernel's first thread advances by one (the sampler's counter, which otherwise costs the data-parallel
// half step a launch of its own right behind this one)
// (the next step's choice of conv1 kernels: reads the pinned word, flips cpp_ctx::conv1_f32; and the step's closing copy + reset)
void ctx_route_update(cpp_ctx* ctx);
int ctx_route_publish(cpp_ctx* ctx);
// every stream synchronisation of the library goes through here: the calls entered so far are complete, which the route decision may use
inline hipError_t ctx_sync_stream(cpp_ctx* ctx) {
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) ctx->route_done = ctx->route_calls;
  return e;
}
// the publisher (device side): the scale word of the call that is running goes to its pinned slot, tagged; the word starts over
__device__ __forceinline__ void route_publish_device(unsigned* wmax_dev, const unsigned* tag_dev, unsigned long long* pin) {
  const unsigned m = *wmax_dev;
  *wmax_dev = 0u;
  const unsigned t = __hip_atomic_load(tag_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(pin + (t & 1u), ((unsigned long long)t << 32) | (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_stats_finalize(cpp_ctx* ctx, const double* part, int nparts, int which_count, int C,
                          double count, float* white, double eps = 1e-6, uint64_t* bump = nullptr, unsigned* wmax = nullptr);
int launch_stats_generic(cpp_ctx* ctx, const void* x, int dtype, long npix, int C, float* white, double eps = 1e-6);
int launch_replay_fill(cpp_ctx* ctx, __half* store, long elems, int slots, int32_t* s1, int32_t* s2,
                       float* action, float* reward, float* mask, int rows, int action_dim,
                       uint64_t seed);
int launch_f32_to_f16(cpp_ctx* ctx, __half* dst, const float* src, long n);
int launch_counter_add(cpp_ctx* ctx, uint64_t* counter, uint64_t inc, const int* unless = nullptr);

// ---------------------------------------------------------------------------------------------
// optimiser (optim.hip)
// ---------------------------------------------------------------------------------------------
enum OptKind { OPT_SGD = 0, OPT_MOMENTUM = 1, OPT_ADAM = 2 };
#define OPT_MAX_SEGS 4
// Up to 4 flat (param, grad) segments.  Segments with the same `group` share one global gradient norm
// (tf.clip_by_global_norm over one gradient list): DDPG clips the actor and critic lists separately
// (ddpg_cartpole.py:116,216), NAF clips one list spanning all its networks (naf_cartpole.py:237).
struct OptSegs {
  float* p[OPT_MAX_SEGS]; const float* g[OPT_MAX_SEGS]; float* m[OPT_MAX_SEGS]; float* v[OPT_MAX_SEGS];
  long n[OPT_MAX_SEGS]; float lr[OPT_MAX_SEGS]; int group[OPT_MAX_SEGS];
  int nseg; int kind; float momentum, beta1, beta2, epsilon;
  const double* sq; int sq_begin[2], sq_count[2];   // sq != nullptr: group g's squared norm = sum of sq[sq_begin[g] .. + sq_count[g]) instead of `part`
  const uint64_t* step;        // Adam: number of applies so far INCLUDING this one (device counter)
  uint64_t* bump;              // optional: device counter incremented once by this launch (the replay sampler's Philox counter)
  // optional (tgt[seg] != nullptr): this launch closes an OUTER step -- a segment's target network takes its soft update from
  // the parameter values this launch writes (ddpg_cartpole.py:336-337 behind the last minibatch: soft_update_kernel's launch disappears)
  float* tgt[OPT_MAX_SEGS]; float tgt_coeff;
  // optional (pub_wmax != nullptr): this launch closes a training call that has no target update -- its first thread publishes the call's
  // largest whitening scale (common.h: route_publish_device) instead of a launch of its own; never together with the statistics rider
  unsigned* pub_wmax; const unsigned* pub_tag; unsigned long long* pub_pin;
  const int* skip_if;          // optional: device flag; non-zero = leave the parameters alone (NAF's check_numerics flag: tf.check_numerics
                               // raises before the train op runs, naf_cartpole.py:242-245 -- decided on the device when nobody waits for the loss)
  // optional rider (st_part != nullptr): the whitening tables of the NEXT minibatch, whose sample pass has already run beside this
  // one's conv1 dW -- stats_finalize_kernel's work (stats_body.h) in st_jobs = columns x channels waves of an extra grid row
  const double* st_part; float* st_white; int st_nparts, st_jobs, st_C; double st_count, st_eps; unsigned* st_wmax;
  // second optional rider (img_n > 0; needs the first one's inputs or finished tables): conv1's operand images for the NEXT minibatch's forward (conv_rs16.h) -- one
  // workgroup per network recomputes its conv1 weights as this launch updates them (gw == nullptr: a target network, untouched), its
  // state column's whitening table as the first rider computes it, and builds the image: conv1_image_kernel's launch disappears
  int img_n; int img_cin;        // img_cin: conv1's input channels (the image body's instance)
  long img_skip[OPT_MAX_SEGS];   // leading parameters of a segment (conv1's weights and biases) that its image workgroup updates itself
  struct { float* w; float* bias; const float* gw; const float* gb; unsigned char* rec; int seg, col, nout;
           const float* white; float* mw; float* mb; } img[4];      // mw / mb: the Momentum slots of those parameters (OPT_MOMENTUM)      // white != nullptr: the column's table is already in memory (the dW reductions' launch computed it)
};
// the SGD update of one parameter, p - lr * (g * scale), with its roundings pinned (one product, one fused multiply-add): opt_apply_kernel
// writes it, the conv1 image rider of the same launch recomputes it, and both must hold the same bits whatever the compiler contracts
__host__ __device__ inline float sgd_update(float p, float g, float scale, float lr) { return __builtin_fmaf(-lr, g * scale, p); }
// ... and Momentum's pair (util.py:73-76: accum = momentum * accum + g; p -= lr * accum), pinned the same way
__host__ __device__ inline float momentum_accum(float m, float g, float scale, float momentum) { return __builtin_fmaf(momentum, m, g * scale); }
__host__ __device__ inline float momentum_step(float p, float accum, float lr) { return __builtin_fmaf(-lr, accum, p); }
// ... and the target networks' update, target - coeff * (target - source) (base_network.py:31), pinned the same way: soft_update_kernel
// writes it, and so does the optimiser's launch when it closes an outer step (OptSegs::tgt)
__host__ __device__ inline float soft_update_value(float t, float s, float coeff) { return __builtin_fmaf(-coeff, t - s, t); }
int launch_sumsq(cpp_ctx* ctx, const OptSegs& s, float grad_scale, double* part, int nparts);
int launch_opt_apply(cpp_ctx* ctx, const OptSegs& s, float grad_scale, float clip, const double* part,
                     int nparts, float* norms_out);
int launch_soft_update(cpp_ctx* ctx, float* t0, const float* s0, long n0, float* t1, const float* s1,
                       long n1, float coeff);

// fused DDPG heads (heads.hip): actor heads, the critic's concat layer + q on three inputs, TD, dQ/da, one backward layer
struct DdpgHeadsArgs {
  int B, A; float discount;
  const float *h2a, *h2ta; int ld_h2a, n2a;        // inputs of the actors' last layer (B x (n2a + 1))
  const float *Wo, *Wo_t;                          // [(n2a + 1)][A]
  const float *h2c, *h2tc; int ld_h2c, n2c;        // inputs of the critics' concat layer, first n2c columns (B x (n2c + A + 1))
  const float *W3, *W3_t; int n3;                  // [(n2c + A + 1)][n3]
  const float *wq, *wq_t;                          // [(n3 + 1)][1]
  const float *act, *r, *mask;                     // the batch's actions, rewards, terminal masks
  float *a_out, *dq_da, *adz, *dz_h2a; int relu_x2;
  float* cat_splice;                               // the concat layer's action columns (row stride ld_h2c)
  float* h3_out; int ld_h3;                        // input buffer of the q layer
  float *q_out, *tq_out, *td, *dzq, *dz3, *dz2c;
  double* loss_part;                               // [DDPG_HEADS_MAX_WGS] per-workgroup sums of td^2
  // optional: the actors' last hidden layer as well (n1a > 0): h2a = relu([h1a, 1] [W2; b2]) is computed here (and left in
  // h2a_out for the head's dW), and dz of the layer below comes out in dz_h1a.  h2a / h2ta are then unused.
  const float *h1a, *h1ta; int ld_h1a, n1a;        // B x (n1a + 1)
  const float *W2, *W2_t;                          // [(n1a + 1)][n2a]
  float *h2a_out, *dz_h1a;                         // B x ld_h2a (first n2a columns), B x n1a
};
#define DDPG_HEADS_MAX_WGS 256
size_t ddpg_heads_lds_bytes(const DdpgHeadsArgs& h);
bool ddpg_heads_supported(const DdpgHeadsArgs& h);
int launch_ddpg_heads(cpp_ctx* ctx, const DdpgHeadsArgs& h);

struct NafHeadArgs {
  const float* value; const float* mu; const float* lv; const float* action; const float* reward;
  const float* mask; const float* target_value;
  float discount; int B, A;
  float* adv; float* q; float* td; float* loss;     // loss[0]
  float* d_value; float* d_mu_z; float* d_l;        // nullptr: forward only
  int* nonfinite;                                   // set to 1 when l_values / L / loss are not finite
};
int launch_naf_head(cpp_ctx* ctx, const NafHeadArgs& a);
// NAF with the shared representation (naf_cartpole.py:151-152, :176-177), everything between the last hidden layer and the layer
// below it in ONE row-local launch (gemm.hip: naf_heads_kernel): the four head layers (value, mu, l_values on state_1, the target
// value on state_2), naf_head_kernel's body, and d(representation) = the three heads' contributions in the order value, mu,
// l_values, masked by the hidden layer's ReLU -- instead of a forward GEMM level, the head kernel and three dependent GEMM levels.
struct NafHeadsArgs {
  int B, A, rep; float discount;
  const float *x, *xt; long ldx;                     // value's / target value's input_state_representation rows, [rep values, 1.0]
  const float *Wv, *Wmu, *Wl, *Wvt;                  // [(rep + 1)][1 | A | A(A+1)/2 | 1]
  const float *action, *reward, *mask;
  float *value, *mu, *lv, *target_value;             // head outputs (B x 1 | A | NL | 1)
  float *adv, *q, *td, *loss;                        // loss[0]
  float *d_value, *d_mu_z, *d_l;                     // gradients at the heads' pre-activations
  float* drep; long ldd; const float* Y; long ldy; int epi;   // d(rep) (GE_NONE | GE_MUL_RELU_GRAD | GE_MUL_RELU_GRAD_X2 on Y)
  int* nonfinite;
  double* part; unsigned* ticket;                    // per-workgroup sums of td^2 (and bad flags behind them), arrival counter
  unsigned long long* step_bump;                     // non-null: += 1 (the optimiser's step counter: nobody reads it before the optimiser launch)
};
#define NAF_HEADS_MAX_WGS 64
// ... and with TWO hidden layers (the reference's 100, 50) the second one as well, forward and backward, on the matrix pipes
// (gemm.hip: naf_mlp_kernel): h = the heads' arguments (x / xt / Y unused: the representation is computed here; drep = dz of layer 1)
struct NafMlpArgs {
  NafHeadsArgs h;
  const float *x0, *x0t; long ld0; int n0;           // first hidden layer's activations [n0 values, 1.0], live and target
  const float *W1, *W1t;                             // [(n0 + 1)][rep]
  float* h1_out; long ld1;                           // the live representation, where the heads' dW GEMMs read it (B x (rep + 1))
  float* dz0;                                        // B x n0: dz of the first hidden layer
};
bool naf_mlp_supported(const NafMlpArgs& m);
int launch_naf_mlp(cpp_ctx* ctx, const NafMlpArgs& m);
bool naf_heads_supported(const NafHeadsArgs& a);
int launch_naf_heads(cpp_ctx* ctx, const NafHeadsArgs& a);

// ---------------------------------------------------------------------------------------------
// launch bookkeeping
// ---------------------------------------------------------------------------------------------
void cpp_set_error(const char* fmt, ...);
// Device allocations of the library (rt_core.cpp: every Arena block sits between two 256-byte guard bands and is registered).
// The f16-pipe conv1 kernels load 16-byte operand windows that may start up to `before` bytes in front of an image batch and
// end up to `after` bytes behind it (masked out afterwards): their launchers refuse any pointer for which those bytes are not
// inside a registered block -- nothing outside the library's own allocations can reach such a kernel.
bool cpp_arena_covers(const void* p, size_t bytes, size_t before, size_t after);
void cpp_arena_register(const void* raw, size_t bytes);
void cpp_arena_unregister(const void* raw);
void prof_begin(cpp_ctx* ctx);
void prof_end(cpp_ctx* ctx, int kid);

#define HIP_CHECK(expr)                                                                    \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      cpp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)

#define LAUNCH_CHECK()                                                                     \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess) {                                                                \
      cpp_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)
