// errors, the per-kernel HIP-event profile, contexts (cpp_ctx_*, cpp_sync, cpp_timer_*, cpp_prof_*)
#include "rt_internal.h"

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

// registry of the library's device allocations (see common.h: cpp_arena_covers)
#include <map>
#include <mutex>
static std::mutex g_arena_mu;
static std::map<const char*, size_t> g_arena_blocks;       // block start -> bytes (guard bands included)
void cpp_arena_register(const void* raw, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  g_arena_blocks[(const char*)raw] = bytes;
}
void cpp_arena_unregister(const void* raw) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  g_arena_blocks.erase((const char*)raw);
}
bool cpp_arena_covers(const void* p, size_t bytes, size_t before, size_t after) {
  std::lock_guard<std::mutex> lk(g_arena_mu);
  const char* lo = (const char*)p - before;
  const char* hi = (const char*)p + bytes + after;
  auto it = g_arena_blocks.upper_bound(lo);
  if (it == g_arena_blocks.begin()) return false;
  --it;
  return lo >= it->first && hi <= it->first + it->second;
}
extern "C" int cpp_abi_version(void) { return CPP_ABI_VERSION; }

void prof_begin(cpp_ctx* ctx) {
  if (ctx->prof) (void)hipEventRecord(ctx->pe0, ctx->stream);
}
void prof_end(cpp_ctx* ctx, int kid) {
  if (!ctx->prof) return;
  // a dW / dX that its launcher parked in the open pair slot was not launched (conv*_bwd_pair.hip times the joint launch); one that the
  // slot did not take -- a geometry without a pair instance: cfg5's 64x64 conv2, the 25x25 conv2 of the 50x50 render -- was, and counts
  if (ctx->pair && (((kid == K_CONV3_DW || kid == K_CONV2_DW) && ctx->pair->have_dw) ||
                    ((kid == K_CONV3_DX || kid == K_CONV2_DX) && ctx->pair->have_dx))) return;
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
    "td", "sumsq", "clip_sgd", "soft_update", "replay_fill", "naf_head", "conv1_fwd_f16", "conv1_dw_f16", "heads", "conv3_bwd", "conv2_bwd", "reduce_gather", "conv1_dw_gather", "allreduce", "conv1_image"};

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
  {  // (ablation build: plan grids, bands and partial buffers as for a smaller device -- a 32-CU CPX partition -- on the whole chip:
     // tests/test_gpu_small_partition.py; the release build's cpp_switch_int is a constant)
    const int pretend = cpp_switch_int("CPP_NUM_CUS", 0);
    if (pretend > 0 && pretend < c->num_cus) c->num_cus = pretend;
  }
  HIP_CHECK(hipMalloc((void**)&c->sq_part, 2 * SQ_REGION * sizeof(double)));
  HIP_CHECK(hipMemsetAsync(c->sq_part, 0, 2 * SQ_REGION * sizeof(double), c->stream));
  c->sq_n[0] = c->sq_n[1] = -1;
  for (int& g : c->sq_conv_group) g = -1;
  // the largest whitening scale of a training call, for a later call's choice of conv1 kernels (common.h: cpp_ctx::conv1_f32)
  HIP_CHECK(hipMalloc((void**)&c->white_max_dev, sizeof(unsigned)));
  HIP_CHECK(hipMemsetAsync(c->white_max_dev, 0, sizeof(unsigned), c->stream));
  // (signal memory: hipStreamWriteValue32 -- a command-processor packet, no kernel -- writes the call's number into it; where the
  // runtime offers neither, an ordinary word and a one-dword fill per call)
  c->route_tag_signal = hipExtMallocWithFlags((void**)&c->route_tag_dev, 8, hipMallocSignalMemory) == hipSuccess && c->route_tag_dev != nullptr;
  if (!c->route_tag_signal) { (void)hipGetLastError(); HIP_CHECK(hipMalloc((void**)&c->route_tag_dev, 8)); }
  HIP_CHECK(hipMemsetAsync(c->route_tag_dev, 0xFF, sizeof(unsigned), c->stream));      // (no call is running)
  HIP_CHECK(hipHostMalloc((void**)&c->route_pin, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  c->route_pin[0] = c->route_pin[1] = ~0ull;
  HIP_CHECK(hipHostGetDevicePointer((void**)&c->route_pin_dev, c->route_pin, 0));
  for (hipEvent_t& e : c->route_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  c->route_calls = c->route_done = c->route_min_tag = 0; c->route_last_max = 0.f;
  c->route_threshold = 100.f;
  *out = c;
  return CPP_OK;
}

// the largest scale of call t, if its publish has landed under tag t (0: nothing published -- no statistics in that call, or it failed)
static bool route_read(cpp_ctx* ctx, uint64_t t, float* m) {
  const unsigned long long w = *reinterpret_cast<volatile unsigned long long*>(ctx->route_pin + (t & 1));
  const unsigned bits = (unsigned)w;
  if ((unsigned)(w >> 32) != (unsigned)t || bits == 0u) return false;
  memcpy(m, &bits, sizeof(*m));
  return true;
}
// Called by every training entry point before it launches anything.  Call k decides from the scale call T published, T = k - 2 -- or
// k - 1 if the stream has been synchronised since call k - 1 was entered (ctx_sync_stream: eager passes in front of a graph capture,
// cpp_sync, parameter reads): T depends on the ORDER of the caller's calls only.  The event recorded at the entry of call k - 1 fires
// when everything before it -- call k - 2 and its publish -- has finished; in a loop that runs ahead of the GPU by one call it has
// fired long ago.  (Round 5 read the pinned word "without waiting for anything": which later step first saw a glint depended on when
// the closing kernel of an earlier graph happened to land.)  Above the threshold conv1 runs on the f32-input kernels; it comes back
// once the scale has fallen under half of it.  A flip moves kernel_epoch, on which the trainers' captured graphs are keyed.
void ctx_route_update(cpp_ctx* ctx) {
  if (!ctx->route_pin) return;
  const uint64_t k = ctx->route_calls;
  long long T = (long long)k - 2;
  if ((long long)ctx->route_done - 1 > T) T = (long long)ctx->route_done - 1;
  else if (k >= 2) (void)hipEventSynchronize(ctx->route_ev[(k - 1) & 1]);
  float m = 0.f;
  if (ctx->route_threshold > 0.f && T >= (long long)ctx->route_min_tag && route_read(ctx, (uint64_t)T, &m)) {
    ctx->route_last_max = m;
    const bool want = ctx->conv1_f32 ? m > 0.5f * ctx->route_threshold : m > ctx->route_threshold;
    if (want != ctx->conv1_f32) { ctx->conv1_f32 = want; ctx->kernel_epoch++; }
  }
  // this call: its entry event, and its number for the publishers among its launches (stream order: behind call k - 1's last launch)
  (void)hipEventRecord(ctx->route_ev[k & 1], ctx->stream);
  if (ctx->route_tag_signal && hipStreamWriteValue32(ctx->stream, ctx->route_tag_dev, (uint32_t)k, 0) != hipSuccess) {
    (void)hipGetLastError();
    ctx->route_tag_signal = false;                     // (not on this runtime: the fill from here on)
  }
  if (!ctx->route_tag_signal) (void)hipMemsetD32Async((hipDeviceptr_t)ctx->route_tag_dev, (int)(unsigned)k, 1, ctx->stream);
  ctx->route_calls = k + 1;
}
// ... and by every training step whose last launch is not a target update (which carries the publish as a rider: optim.hip), inside its captured graph
__global__ void route_publish_kernel(unsigned* wmax_dev, const unsigned* tag_dev, unsigned long long* pin) { route_publish_device(wmax_dev, tag_dev, pin); }
int ctx_route_publish(cpp_ctx* ctx) {
  if (!ctx->route_pin) return CPP_OK;
  hipLaunchKernelGGL(route_publish_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->white_max_dev, ctx->route_tag_dev, ctx->route_pin_dev);
  LAUNCH_CHECK();
  return CPP_OK;
}

extern "C" int cpp_ctx_set_route_threshold(cpp_ctx* c, float threshold) {
  ARG_CHECK(c, "cpp_ctx_set_route_threshold: ctx is NULL");
  ARG_CHECK(threshold >= 0.f, "cpp_ctx_set_route_threshold: threshold %g", threshold);
  c->route_threshold = threshold;
  c->route_min_tag = c->route_calls;                   // (what the calls so far saw under the old threshold decides nothing any more)
  if (threshold == 0.f && c->conv1_f32) { c->conv1_f32 = false; c->kernel_epoch++; }
  return CPP_OK;
}
// last_max_scale: the largest whitening scale of the newest call known to have finished (after cpp_sync: the last one entered)
extern "C" int cpp_ctx_get_route(cpp_ctx* c, int* conv1_f32, float* last_max_scale) {
  ARG_CHECK(c, "cpp_ctx_get_route: ctx is NULL");
  if (conv1_f32) *conv1_f32 = c->conv1_f32 ? 1 : 0;
  if (last_max_scale) {
    float m = c->route_last_max;
    if (c->route_pin && c->route_done >= 1) (void)route_read(c, c->route_done - 1, &m);
    *last_max_scale = m;
  }
  return CPP_OK;
}
extern "C" int cpp_ctx_set_precision(cpp_ctx* c, int mode) {
  ARG_CHECK(c, "cpp_ctx_set_precision: ctx is NULL");
  ARG_CHECK(mode == CPP_PRECISION_FAST || mode == CPP_PRECISION_EXACT, "cpp_ctx_set_precision: mode %d is neither CPP_PRECISION_FAST nor CPP_PRECISION_EXACT", mode);
  ARG_CHECK(c->n_trainers == 0 || mode == c->precision,
            "cpp_ctx_set_precision: %d trainer(s) exist on this ctx (their step graphs hold the kernels of the current mode): set the mode first", c->n_trainers);
  c->precision = mode;
  return CPP_OK;
}

extern "C" int cpp_ctx_get_precision(cpp_ctx* c, int* mode) {
  ARG_CHECK(c && mode, "cpp_ctx_get_precision: NULL argument");
  *mode = c->precision;
  return CPP_OK;
}

extern "C" int cpp_ctx_destroy(cpp_ctx* c) {
  if (!c) return CPP_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipEventDestroy(c->t0); (void)hipEventDestroy(c->t1);
  (void)hipEventDestroy(c->pe0); (void)hipEventDestroy(c->pe1);
  if (c->sq_part) (void)hipFree(c->sq_part);
  if (c->white_max_dev) (void)hipFree(c->white_max_dev);
  if (c->route_tag_dev) (void)hipFree(c->route_tag_dev);
  if (c->route_pin) { (void)hipHostFree(c->route_pin); for (hipEvent_t e : c->route_ev) (void)hipEventDestroy(e); }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return CPP_OK;
}

extern "C" int cpp_sync(cpp_ctx* c) {
  ARG_CHECK(c, "cpp_sync: ctx is NULL");
  HIP_CHECK(ctx_sync_stream(c));
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

