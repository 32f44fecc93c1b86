// Communicator of the data-parallel actor-learners: one rank per GPU / cpp_ctx, RCCL over xGMI.  The reference is
// single-process (TODO "switch back to async training with multiple replicas", ddpg_cartpole.py:259, naf_cartpole.py:294); the
// collective steps that use this communicator are cpp_ddpg_dp_train_step (rt_ddpg.cpp) and cpp_naf_dp_train_step (rt_naf.cpp).
#include "rt_internal.h"

extern "C" int cpp_comm_unique_id(void* out, int cap) {
  ARG_CHECK(out && cap >= (int)sizeof(ncclUniqueId), "cpp_comm_unique_id: need a buffer of %d bytes", (int)sizeof(ncclUniqueId));
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return CPP_OK;
}

extern "C" int cpp_comm_create(cpp_ctx* ctx, const void* unique_id, int rank, int world, cpp_comm** out) {
  ARG_CHECK(ctx && unique_id && out, "cpp_comm_create: NULL argument");
  ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "cpp_comm_create: rank %d of %d", rank, world);
  HIP_CHECK(hipSetDevice(ctx->device));
  cpp_comm* c = new cpp_comm();
  memset(c, 0, sizeof(*c));
  c->ctx = ctx; c->rank = rank; c->world = world;
  static uint64_t next_uid = 1;
  c->uid = next_uid++;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { cpp_set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world, ncclGetErrorString(r)); delete c; return CPP_ERR_HIP; }
  HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_fc, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_bwd, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  *out = c;
  return CPP_OK;
}

extern "C" int cpp_comm_destroy(cpp_comm* c) {
  if (!c) return CPP_OK;
  (void)hipSetDevice(c->ctx->device);
  (void)ctx_sync_stream(c->ctx);
  (void)hipStreamSynchronize(c->side);
  (void)ncclCommDestroy(c->comm);
  if (c->scratch) (void)hipFree(c->scratch);
  (void)hipEventDestroy(c->ev_fc); (void)hipEventDestroy(c->ev_bwd); (void)hipEventDestroy(c->ev_done);
  (void)hipStreamDestroy(c->side);
  delete c;
  return CPP_OK;
}

extern "C" int cpp_comm_info(const cpp_comm* c, int* rank, int* world) {
  ARG_CHECK(c, "cpp_comm_info: NULL argument");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return CPP_OK;
}

// sum (average != 0: mean) over the ranks of n floats at a DEVICE address, in place, on the context's stream
extern "C" int cpp_comm_allreduce(cpp_comm* c, void* device_f32, int64_t n, int average) {
  ARG_CHECK(c && device_f32 && n >= 1, "cpp_comm_allreduce: bad argument");
  HIP_CHECK(hipSetDevice(c->ctx->device));
  NCCL_CHECK(ncclAllReduce(device_f32, device_f32, (size_t)n, ncclFloat, average ? ncclAvg : ncclSum, c->comm, c->ctx->stream));
  return CPP_OK;
}

// max over the ranks of n (<= 8) host doubles, element-wise, via the communicator's device scratch words: bench.py's slowest
// rank's time, and the agents' per-iteration loop agreement under --data-parallel ("every rank past burn-in?", "every rank done?":
// ddpg_cartpole.py:329,379-383 decided per rank would leave the other ranks blocked in the next ncclAllReduce)
extern "C" int cpp_comm_max_doubles(cpp_comm* c, double* values, int n) {
  ARG_CHECK(c && values && n >= 1 && n <= CPP_COMM_SCRATCH_WORDS, "cpp_comm_max_doubles: need 1..%d values", CPP_COMM_SCRATCH_WORDS);
  HIP_CHECK(hipSetDevice(c->ctx->device));
  if (!c->scratch) HIP_CHECK(hipMalloc((void**)&c->scratch, CPP_COMM_SCRATCH_WORDS * sizeof(double)));
  hipStream_t st = c->ctx->stream;
  HIP_CHECK(hipMemcpyAsync(c->scratch, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  NCCL_CHECK(ncclAllReduce(c->scratch, c->scratch, (size_t)n, ncclDouble, ncclMax, c->comm, st));
  HIP_CHECK(hipMemcpyAsync(values, c->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

extern "C" int cpp_comm_max_double(cpp_comm* c, double* value) { return cpp_comm_max_doubles(c, value, 1); }

// barrier: a one-word all-reduce, then wait for it (bench.py brackets its timed region with this + cpp_sync)
extern "C" int cpp_comm_barrier(cpp_comm* c) {
  double one = 1.0;
  return cpp_comm_max_double(c, &one);
}
