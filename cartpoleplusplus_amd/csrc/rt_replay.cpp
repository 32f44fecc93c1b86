// minibatches resident in HBM (cpp_batch_*) and the replay memory payload (cpp_replay_*)
#include "rt_internal.h"

// ---------------------------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------------------------

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

int batch_ensure_stats(cpp_batch* b, int C) {
  if (C <= 0 || b->stats_C == C) return CPP_OK;
  RC(batch_stats(b->ctx, b->s[0], b->s[1], b->dtype, b->elems, b->B, C, b->part, b->white));
  b->stats_C = C;
  return CPP_OK;
}

// ---------------------------------------------------------------------------------------------
// replay
// ---------------------------------------------------------------------------------------------

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
  r->slot_stats = nullptr; r->stats_C = 0; r->stats_cap = 0; r->sampled = false; r->slot_list = nullptr; r->slot_list_cap = 0;
  r->rows_pin = nullptr; r->rows_pin_k = 0; memset(r->rows_pin_used, 0, sizeof(r->rows_pin_used)); memset(r->rows_pin_ev, 0, sizeof(r->rows_pin_ev));
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
  if (!rc) rc = dalloc(r->arena, &r->counter_adhoc, (size_t)1);
  if (!rc) rc = dalloc(r->arena, &r->size_dev, (size_t)1);
  if (!rc) {
    rc = r->arena.alloc((void**)&r->lut, 256 * sizeof(__half), false);
    if (!rc) rc = r->arena.alloc((void**)&r->bad, sizeof(int), true);
    if (!rc) {
      for (int k = 0; k < 256; ++k) r->lut_host[k] = f16_of_code(k);
      if (hipMemcpy(r->lut, r->lut_host, sizeof(r->lut_host), hipMemcpyHostToDevice) != hipSuccess) rc = CPP_ERR_HIP;
    }
  }
  if (rc) { r->arena.release(); delete r; return rc; }
  r->uid = replay_next_uid();
  *out = r;
  return CPP_OK;
}
// rows in the memory: host copy (argument checks) + device word (the sampler's range, also of captured launches)
static int replay_set_size(cpp_replay* r, int size) {
  r->size = size;
  HIP_CHECK(hipMemcpyAsync(r->size_dev, &r->size, sizeof(int32_t), hipMemcpyHostToDevice, r->ctx->stream));
  HIP_CHECK(ctx_sync_stream(r->ctx));
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
  (void)ctx_sync_stream(r->ctx);
  if (r->stage) (void)hipFree(r->stage);
  if (r->slot_list) (void)hipFree(r->slot_list);
  if (r->pinned) (void)hipHostFree(r->pinned);
  if (r->rows_pin) (void)hipHostFree(r->rows_pin);
  for (hipEvent_t e : r->rows_pin_ev) if (e) (void)hipEventDestroy(e);
  (void)hipEventDestroy(r->pinned_free);
  r->arena.release(); delete r; return CPP_OK;
}

// Keep, per state slot, the per-channel sums sum(x), sum(x^2) over the state's pixels for `channels` interleaved channels (NHWC:
// base_network.py:88-96 whitens over (batch, height, width)).  A minibatch's whitening statistics are then the sum of 2 B stored rows of
// 2 C doubles instead of a pass over 2 B images (75 MB per minibatch at 64x64x18, B = 256): the fused steps, whose conv1 reads the store
// itself, never touch the pixels for the sample.  The sums are (re)computed here for every slot and kept current by
// cpp_replay_write_states / cpp_replay_fill_synthetic.  channels = 0, or a channel count the vector statistics path cannot take, turns
// them off (the gather then reads the images, as before).
uint64_t replay_next_uid() { static uint64_t next_uid = 1; return next_uid++; }

extern "C" int cpp_replay_set_stats_channels(cpp_replay* r, int channels) {
  ARG_CHECK(r && channels >= 0 && channels <= CPP_MAX_CHANNELS, "cpp_replay_set_stats_channels: channels %d", channels);
  HIP_CHECK(hipSetDevice(r->ctx->device));
  int C = channels;
  if (C > 0) {
    int g = 8, c = C; while (c) { int t = g % c; g = c; c = t; }
    if (r->elems % 8 != 0 || C / g > 16 || r->elems % C != 0) C = 0;
  }
  // the captured step graphs (rt_ddpg.cpp / rt_naf.cpp, keyed on batch size and memory) have the slot_stats pointer and the
  // direct-store decision baked in.  Their keys compare the memory's uid: a memory whose setting changes after it has been
  // sampled takes a NEW uid, and every such graph is captured again at its next use.
  if (C == r->stats_C && (C == 0 || r->slot_stats)) return CPP_OK;
  if (r->sampled) { r->uid = replay_next_uid(); r->sampled = false; }
  ++r->write_gen;              // (a minibatch presampled under the old setting is stale)
  if (C == 0) { r->stats_C = 0; return CPP_OK; }          // (the buffer, if any, stays allocated and unused)
  if (!r->slot_stats || r->stats_cap < C) {          // (switching off and on again reuses the buffer)
    HIP_CHECK(ctx_sync_stream(r->ctx));
    RC(dalloc(r->arena, &r->slot_stats, (size_t)r->slots * 2 * C));
    r->stats_cap = C;
  }
  r->stats_C = C;
  RC(launch_slot_stats(r->ctx, r->store, r->store_dtype, r->elems, C, r->slot_stats, nullptr, 0, r->slots, r->lut));
  HIP_CHECK(ctx_sync_stream(r->ctx));
  return CPP_OK;
}

// self.state[idx] = s (replay_memory.py:67,106).  The host rows go through a pinned staging buffer, so the call returns
// as soon as they are copied there: the H2D transfer and the conversions run on the context's stream, in order with
// everything launched later (rendered frames straight into replay slots, no stall of the rollout loop).
extern "C" int cpp_replay_write_states(cpp_replay* r, const int32_t* slots, int n, const void* states, int dtype) {
  if (r) ++r->write_gen;
  ARG_CHECK(r && slots && states, "cpp_replay_write_states: NULL argument");
  ARG_CHECK(dtype == CPP_F32 || dtype == CPP_F16 || dtype == CPP_U8, "cpp_replay_write_states: dtype %d", dtype);
  ARG_CHECK(n >= 1 && (double)n * (double)r->elems < 4.0e9, "cpp_replay_write_states: %d states of %ld elements in one call (the conversion kernels run one thread per element: split the episode)", n, r ? r->elems : 0);
  hipStream_t st = r->ctx->stream;
  HIP_CHECK(hipSetDevice(r->ctx->device));
  for (int i = 0; i < n; ++i)
    ARG_CHECK(slots[i] >= 0 && slots[i] < r->slots, "cpp_replay_write_states: slot %d outside [0,%d)", slots[i], r->slots);
  const size_t esz = dtype == CPP_U8 ? 1 : dtype == CPP_F16 ? sizeof(__half) : sizeof(float);
  const size_t need = (size_t)n * r->elems * esz;
  const size_t need_ids = ((need + 15) & ~(size_t)15) + (size_t)n * sizeof(int32_t);     // the slot numbers ride behind the payload
  if (r->pinned_busy) { HIP_CHECK(hipEventSynchronize(r->pinned_free)); r->pinned_busy = false; }   // previous transfer done
  if (need_ids > r->pinned_cap) {
    if (r->pinned) HIP_CHECK(hipHostFree(r->pinned));
    HIP_CHECK(hipHostMalloc(&r->pinned, need_ids, hipHostMallocDefault)); r->pinned_cap = need_ids;
  }
  memcpy(r->pinned, states, need);
  memcpy((char*)r->pinned + ((need + 15) & ~(size_t)15), slots, (size_t)n * sizeof(int32_t));
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
  if (r->slot_stats && r->stats_C > 0) {       // the new states' whitening sums (what a gather of them would compute), once, here
    if ((size_t)n > r->slot_list_cap) {
      HIP_CHECK(hipStreamSynchronize(st));
      if (r->slot_list) HIP_CHECK(hipFree(r->slot_list));
      r->slot_list_cap = (size_t)n < 256 ? 256 : (size_t)n;
      HIP_CHECK(hipMalloc((void**)&r->slot_list, r->slot_list_cap * sizeof(int32_t)));
    }
    HIP_CHECK(hipMemcpyAsync(r->slot_list, (const char*)r->pinned + ((need + 15) & ~(size_t)15), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    RC(launch_slot_stats(r->ctx, r->store, r->store_dtype, r->elems, r->stats_C, r->slot_stats, r->slot_list, 0, n, r->lut));
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
  if (r) ++r->write_gen;
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

// the event columns of n rows, as the device holds them (debug / parity; fill_synthetic's host mirrors).  NULL outputs are skipped.
extern "C" int cpp_replay_read_rows(cpp_replay* r, const int32_t* rows, int n, int32_t* s1, int32_t* s2, float* action, float* reward,
                                    float* mask) {
  ARG_CHECK(r && rows && n >= 0, "cpp_replay_read_rows: bad argument");
  hipStream_t st = r->ctx->stream;
  HIP_CHECK(hipSetDevice(r->ctx->device));
  int i = 0;
  while (i < n) {          // contiguous runs of rows come back as one copy per column
    ARG_CHECK(rows[i] >= 0 && rows[i] < r->rows, "cpp_replay_read_rows: row %d outside [0,%d)", rows[i], r->rows);
    int j = i + 1;
    while (j < n && rows[j] == rows[j - 1] + 1) ++j;
    const int cnt = j - i, r0 = rows[i];
    if (s1) HIP_CHECK(hipMemcpyAsync(s1 + i, r->s1 + r0, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (s2) HIP_CHECK(hipMemcpyAsync(s2 + i, r->s2 + r0, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (action) HIP_CHECK(hipMemcpyAsync(action + (size_t)i * r->A, r->action + (size_t)r0 * r->A, (size_t)cnt * r->A * sizeof(float), hipMemcpyDeviceToHost, st));
    if (reward) HIP_CHECK(hipMemcpyAsync(reward + i, r->reward + r0, cnt * sizeof(float), hipMemcpyDeviceToHost, st));
    if (mask) HIP_CHECK(hipMemcpyAsync(mask + i, r->mask + r0, cnt * sizeof(float), hipMemcpyDeviceToHost, st));
    i = j;
  }
  HIP_CHECK(hipStreamSynchronize(st));
  return CPP_OK;
}

extern "C" int cpp_replay_set_size(cpp_replay* r, int size) {
  if (r) ++r->write_gen;
  ARG_CHECK(r && size >= 0 && size <= r->rows, "cpp_replay_set_size: size %d outside [0,%d]", size, r ? r->rows : 0);
  HIP_CHECK(hipSetDevice(r->ctx->device));
  return replay_set_size(r, size);
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

// n host-drawn row indexes (replay_memory.py:123-129: numpy's RNG on the host) -> r->rows_in, checked, without waiting for the
// stream: through a ring of pinned slots (slot k is reused once the copy that read it has completed)
int replay_stage_rows(cpp_replay* r, const int32_t* idxs, int n, const char* who) {
  ARG_CHECK(n >= 1 && n <= 65536, "%s: %d rows", who, n);
  for (int i = 0; i < n; ++i)
    ARG_CHECK(idxs[i] >= 0 && idxs[i] < r->size, "%s: index %d outside [0,%d)", who, idxs[i], r->size);
  hipStream_t st = r->ctx->stream;
  if (n > CPP_ROWS_RING_SLOT) {
    HIP_CHECK(hipMemcpyAsync(r->rows_in, idxs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return CPP_OK;
  }
  if (!r->rows_pin) {
    HIP_CHECK(hipHostMalloc((void**)&r->rows_pin, (size_t)CPP_ROWS_RING * CPP_ROWS_RING_SLOT * sizeof(int32_t), hipHostMallocDefault));
    for (hipEvent_t& e : r->rows_pin_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int k = r->rows_pin_k;
  r->rows_pin_k = (k + 1) % CPP_ROWS_RING;
  if (r->rows_pin_used[k]) HIP_CHECK(hipEventSynchronize(r->rows_pin_ev[k]));
  int32_t* slot = r->rows_pin + (size_t)k * CPP_ROWS_RING_SLOT;
  memcpy(slot, idxs, (size_t)n * sizeof(int32_t));
  HIP_CHECK(hipMemcpyAsync(r->rows_in, slot, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipEventRecord(r->rows_pin_ev[k], st));
  r->rows_pin_used[k] = true;
  return CPP_OK;
}

// device-only part of sampling (graph-capturable when rows_dev == nullptr or already resident)
// descriptor of the fused sample + gather + statistics pass into `out` (C_out: channels of the vector statistics path, 0: none)
GatherArgs replay_gather_args(cpp_replay* r, int B, const int32_t* rows_dev, uint64_t seed, const uint64_t* counter_dev,
                                     int channels, cpp_batch* out, bool direct, int* C_out) {
  int C = channels;
  r->sampled = true;
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
  a.slot_stats = (direct && C > 0 && r->slot_stats && r->stats_C == C) ? r->slot_stats : nullptr;
  a.out_action = out->a; a.out_reward = out->r; a.out_mask = out->m;
  a.part = out->part; a.seed = seed; a.counter = counter_dev;
  a.elems = r->elems; a.B = B; a.size = r->size; a.size_ptr = r->size_dev; a.action_dim = r->A; a.C = C;
  *C_out = C;
  return a;
}
// what follows the gather kernel: the batch's bookkeeping and the whitening tables
// bump (optional): the sampler's counter, to be advanced by one behind this minibatch's draw -- done by the statistics kernel
// when there is one (*bumped = true), left to the caller otherwise
int replay_sample_finish(cpp_replay* r, int B, int C, int channels, cpp_batch* out, uint64_t* bump, bool* bumped) {
  out->B = B; out->dtype = CPP_F16; out->stats_C = 0;
  if (bumped) *bumped = false;
  if (C > 0) {
    RC(launch_stats_finalize(r->ctx, out->part, B, 2, C, (double)B * (double)(r->elems / C), out->white, 1e-6, bump, r->ctx->white_max_dev));
    if (bumped && bump) *bumped = true;
    out->stats_C = C;
  } else if (channels > 0) {
    RC(batch_ensure_stats(out, channels));
  }
  return CPP_OK;
}
int replay_sample_device(cpp_replay* r, int B, const int32_t* rows_dev, uint64_t seed, const uint64_t* counter_dev,
                                int channels, cpp_batch* out, bool direct, uint64_t* bump, bool* bumped) {
  int C = 0;
  const GatherArgs a = replay_gather_args(r, B, rows_dev, seed, counter_dev, channels, out, direct, &C);
  RC(launch_gather_stats(r->ctx, a, r->store_dtype));      // a CPP_U8 store gathers to f16 as well
  return replay_sample_finish(r, B, C, channels, out, bump, bumped);
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
    HIP_CHECK(hipMemcpyAsync(r->counter_adhoc, &counter, sizeof(uint64_t), hipMemcpyHostToDevice, st));
  }
  RC(replay_sample_device(r, B, rows_dev, seed, idxs ? nullptr : r->counter_adhoc, channels, out));
  if (idxs) HIP_CHECK(hipStreamSynchronize(st));     // the caller's index array may go away after return
  return CPP_OK;
}

extern "C" int cpp_replay_last_indexes(cpp_replay* r, int B, int32_t* out) {
  ARG_CHECK(r && out && B >= 1 && B <= 65536, "cpp_replay_last_indexes: bad argument");
  HIP_CHECK(hipMemcpyAsync(out, r->rows_out, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, r->ctx->stream));
  HIP_CHECK(ctx_sync_stream(r->ctx));
  return CPP_OK;
}

extern "C" int cpp_replay_fill_synthetic(cpp_replay* r, int n_rows, uint64_t seed) {
  if (r) ++r->write_gen;
  ARG_CHECK(r && n_rows >= 1 && n_rows <= r->rows, "cpp_replay_fill_synthetic: rows %d", n_rows);
  ARG_CHECK(n_rows + n_rows / 50 + 1 <= r->slots, "cpp_replay_fill_synthetic: not enough state slots");
  HIP_CHECK(hipSetDevice(r->ctx->device));
  RC(launch_replay_fill(r->ctx, r->store_dtype == CPP_U8 ? nullptr : (__half*)r->store, r->elems, r->slots, r->s1, r->s2,
                        r->action, r->reward, r->mask, n_rows, r->A, seed));
  if (r->store_dtype == CPP_U8) RC(launch_replay_fill_u8(r->ctx, (uint8_t*)r->store, r->elems * (long)r->slots, seed));
  if (r->slot_stats && r->stats_C > 0) RC(launch_slot_stats(r->ctx, r->store, r->store_dtype, r->elems, r->stats_C, r->slot_stats, nullptr, 0, r->slots, r->lut));
  HIP_CHECK(ctx_sync_stream(r->ctx));
  return replay_set_size(r, n_rows);
}

