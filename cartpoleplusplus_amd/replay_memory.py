"""Device-resident replay memory with the interface of the reference's replay_memory.py.

Reference surface kept (paths relative to /root/reference): `Batch` (replay_memory.py:9) and
`ReplayMemory(buffer_size, state_shape, action_dim, load_factor=1.5)` with `add_episode`, `size`,
`random_indexes`, `batch`, `current_stats`, `dump` and the public attributes `insert`, `full`,
`state_1_idx`, `state_2_idx`, `action`, `reward`, `terminal_mask`, `state`, `state_free_slots`,
`stats` (replay_memory.py:11-163), plus `batch(idxs=...)` / `Batch.state_1_idx` that the reference's
own test still expects (replay_memory_test.py:84,129).

What is different underneath: the f16 state store (replay_memory.py:32) lives in HBM behind
cpp_replay_*; only the slot bookkeeping (FIFO free list, eviction -- replay_memory.py:66,84,90,104)
runs on the host so its order is exactly the reference's.  `batch()` is the fused sample + gather
kernel; the returned Batch keeps the minibatch on the device and only copies a column to the host
when that column is read.
"""
import collections
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import lib, check, ptr

_FIELDS = ("state_1", "action", "reward", "terminal_mask", "state_2")


class Batch(object):
    """replay_memory.py:9 `Batch = namedtuple("Batch", "state_1 action reward terminal_mask state_2")`.
    Tuple protocol preserved (len 5, indexing, iteration, attribute names); columns are fetched from
    the device on first access (fresh host copies the caller owns, like np.copy in :134-138)."""
    _fields = _FIELDS

    def __init__(self, device_batch, state_shape, idxs=None, s1_idx=None, s2_idx=None, empty=False, memory=None):
        self._dev, self._shape, self._cache, self._empty = device_batch, tuple(state_shape), {}, empty
        self.idxs, self.state_1_idx, self.state_2_idx = idxs, s1_idx, s2_idx
        # the device buffer behind this Batch is shared with the next batch() of the same size: `memory` detaches this
        # Batch (see _detach) before it resamples, so that the columns read later are still THIS draw's (np.copy semantics
        # of replay_memory.py:134-138)
        self._memory, self._detached, self._small, self._store_gen = memory, False, None, None

    @classmethod
    def empty(cls, state_shape, action_dim):
        b = cls(None, state_shape, empty=True)
        b._cache = {"state_1": np.empty((0,) + tuple(state_shape), np.float16),
                    "action": np.empty((0, action_dim), np.float32),
                    "reward": np.empty((0, 1), np.float32),
                    "terminal_mask": np.empty((0, 1), np.float32),
                    "state_2": np.empty((0,) + tuple(state_shape), np.float16)}
        return b

    def _detach(self):
        """the owner is about to overwrite the shared device buffer: keep this draw readable.  The three small columns are
        copied now (a few KB); the state columns are re-read on demand from the replay store through the slots recorded at
        sample time -- valid as long as no state has been written to the memory since (otherwise reading them raises)."""
        if self._cache or self._empty or self._detached:
            return
        d = self._dev
        B = d.size
        a, r, m = np.empty((B, d.action_dim), np.float32), np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
        check(lib.cpp_batch_download(d.handle, None, None, ptr(a), ptr(r), ptr(m)))
        self._small = {"action": a, "reward": r, "terminal_mask": m}
        self._store_gen = self._memory._write_gen
        self._detached, self._dev = True, None

    def _fetch(self):
        if self._cache or self._empty:
            return
        if self._detached:
            rm = self._memory
            if rm._write_gen != self._store_gen or rm.handle is None:
                raise RuntimeError("this Batch was not read before its device buffer was resampled AND states have been "
                                   "written to the replay memory since: its state columns are gone (read a column, or "
                                   "finish with the Batch, before the next batch() / add_episode())")
            self._cache = dict(self._small, state_1=rm.state[self.state_1_idx], state_2=rm.state[self.state_2_idx])
            return
        d = self._dev
        B = d.size
        dt = np.float16 if d.state_dtype == _lib.CPP_F16 else np.float32
        s1 = np.empty((B,) + self._shape, dt)
        s2 = np.empty((B,) + self._shape, dt)
        a = np.empty((B, d.action_dim), np.float32)
        r = np.empty((B, 1), np.float32)
        m = np.empty((B, 1), np.float32)
        check(lib.cpp_batch_download(d.handle, ptr(s1), ptr(s2), ptr(a), ptr(r), ptr(m)))
        self._cache = {"state_1": s1, "action": a, "reward": r, "terminal_mask": m, "state_2": s2}

    def __getattr__(self, name):
        if name in _FIELDS:
            self._fetch()
            return self._cache[name]
        raise AttributeError(name)

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return getattr(self, _FIELDS[i])

    def __iter__(self):
        return (getattr(self, f) for f in _FIELDS)

    @property
    def device(self):
        """the DeviceBatch behind this Batch (None for the empty batch)."""
        return self._dev


class DeviceBatch(object):
    """A minibatch resident in HBM (cpp_batch)."""

    def __init__(self, max_batch, state_elems, action_dim, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self.max_batch, self.state_elems, self.action_dim = int(max_batch), int(state_elems), int(action_dim)
        h = C.c_void_p()
        check(lib.cpp_batch_create(self.ctx.handle, self.max_batch, self.state_elems, self.action_dim, C.byref(h)))
        self.handle = h

    @property
    def size(self):
        return lib.cpp_batch_size(self.handle)

    @property
    def state_dtype(self):
        return lib.cpp_batch_state_dtype(self.handle)

    def upload(self, state_1, action=None, reward=None, terminal_mask=None, state_2=None):
        s1, dt = _lib.as_state_array(state_1)
        B = s1.shape[0]
        s2 = None
        if state_2 is not None:
            s2 = np.ascontiguousarray(state_2, dtype=s1.dtype)
        f32 = lambda x, cols: None if x is None else np.ascontiguousarray(
            np.asarray(x, dtype=np.float32).reshape(B, cols))
        a, r, m = f32(action, self.action_dim), f32(reward, 1), f32(terminal_mask, 1)
        assert s1.size == B * self.state_elems, (s1.shape, self.state_elems)
        check(lib.cpp_batch_upload(self.handle, B, ptr(s1), ptr(s2), dt, ptr(a), ptr(r), ptr(m)))
        return self

    def close(self):
        if self.handle:
            lib.cpp_batch_destroy(self.handle)
            self.handle = None


class _StateStoreView(object):
    """`rm.state[slot]` / `rm.state[[slots]]` reads the f16 payload back from HBM (debug / dump /
    the reference test's `state[0][0][0] == 11` style checks, replay_memory_test.py:52-56)."""

    def __init__(self, rm):
        self._rm = rm
        self.dtype = np.dtype(np.float16)
        self.shape = (rm.state_buffer_size,) + tuple(rm.state_shape)

    def __getitem__(self, key):
        scalar = np.isscalar(key)
        slots = np.ascontiguousarray(np.atleast_1d(np.asarray(key)), dtype=np.int32)
        out = np.empty((len(slots),) + tuple(self._rm.state_shape), np.float16)
        if len(slots):
            check(lib.cpp_replay_read_states(self._rm.handle, ptr(slots), len(slots), ptr(out)))
        return out[0] if scalar else out

    def __len__(self):
        return self.shape[0]


class ReplayMemory(object):
    def __init__(self, buffer_size, state_shape, action_dim, load_factor=1.5, ctx=None, store_dtype="f16"):
        """store_dtype "f16" is the reference's store (replay_memory.py:32).  "u8" keeps 8-bit pixel codes k that read
        back as f16(k/255) -- bit-identical batches for the reference's renders (bullet_cartpole.py:239-243) in half
        the HBM and half the gather traffic; it refuses states that are not such images."""
        assert load_factor >= 1.5, "load_factor has to be at least 1.5"      # replay_memory.py:13
        self.ctx = ctx or _lib.default_context()
        self.buffer_size = int(buffer_size)
        self.state_shape = tuple(int(d) for d in state_shape)
        self.action_dim = int(action_dim)
        self.insert, self.full = 0, False
        n = self.buffer_size
        # host mirrors of the event columns (the device copies feed the gather kernel)
        self.state_1_idx = np.empty(n, dtype=np.int32)
        self.action = np.empty((n, self.action_dim), dtype=np.float32)
        self.reward = np.empty((n, 1), dtype=np.float32)
        self.terminal_mask = np.empty((n, 1), dtype=np.float32)
        self.state_2_idx = np.empty(n, dtype=np.int32)
        self.state_buffer_size = int(n * load_factor)                         # replay_memory.py:30
        self.state_elems = int(np.prod(self.state_shape))
        self.state_free_slots = collections.deque(range(self.state_buffer_size))
        self.stats = collections.Counter()
        h = C.c_void_p()
        assert store_dtype in ("f16", "u8"), store_dtype
        self.store_dtype = store_dtype
        check(lib.cpp_replay_create_ex(self.ctx.handle, n, self.state_buffer_size, self.state_elems, self.action_dim,
                                       _lib.CPP_U8 if store_dtype == "u8" else _lib.CPP_F16, C.byref(h)))
        self.handle = h
        self.state = _StateStoreView(self)
        self._batches = {}
        self._live = {}              # batch size -> weakref of the Batch that currently shares that DeviceBatch
        self._write_gen = 0          # bumped by every write of states (add_episode, fill_synthetic)
        self._adhoc_counter = 0      # sample_on_device draws (separate from the train steps' device counter)
        # pixel states (H, W, 3, cameras, repeats): channel count for the fused whitening statistics
        self.channels = int(np.prod(self.state_shape[2:])) if len(self.state_shape) == 5 else 0

    # --- host-side slot bookkeeping: exactly replay_memory.py:63-118 ---------------------------
    def _pop_slot(self):
        if not self.state_free_slots:
            raise RuntimeError("replay memory state store exhausted (load_factor too small for this "
                               "many short episodes; SURVEY appendix B9)")
        return self.state_free_slots.popleft()

    def add_episode(self, initial_state, action_reward_state_sequence):
        self.stats[">add_episode"] += 1
        seq = list(action_reward_state_sequence)
        assert len(seq) > 0
        n = len(seq)
        slots = np.empty(n + 1, np.int32)
        rows = np.empty(n, np.int32)
        s1 = np.empty(n, np.int32)
        s2 = np.empty(n, np.int32)
        # host bookkeeping is committed only once the device writes have succeeded: a slot store that runs dry mid-episode
        # or a write the device refuses (a non-image state for the 8-bit store) leaves the memory exactly as it was
        saved = dict(insert=self.insert, full=self.full, free=collections.deque(self.state_free_slots), rows=[],
                     stats=collections.Counter(self.stats))
        try:
            self._add_episode_locked(initial_state, seq, n, slots, rows, s1, s2, saved)
        except Exception:
            for row, vals in reversed(saved["rows"]):
                (self.state_1_idx[row], self.action[row], self.reward[row], self.terminal_mask[row], self.state_2_idx[row]) = vals
            self.state_free_slots = saved["free"]
            self.insert, self.full, self.stats = saved["insert"], saved["full"], saved["stats"]
            self.stats[">add_episode"] += 1
            raise

    def _add_episode_locked(self, initial_state, seq, n, slots, rows, s1, s2, saved):
        slots[0] = self._pop_slot()
        for k in range(n):
            self.stats[">add"] += 1
            row = self.insert
            saved["rows"].append((row, (int(self.state_1_idx[row]), self.action[row].copy(), self.reward[row].copy(),
                                       self.terminal_mask[row].copy(), int(self.state_2_idx[row]))))
            if self.full:
                self.state_free_slots.append(int(self.state_1_idx[row]))            # :84
                if self.terminal_mask[row] == 0:                                     # :89-91
                    self.state_free_slots.append(int(self.state_2_idx[row]))
                    self.stats["cache_evicted_s2"] += 1
            action, reward, _state_2 = seq[k]
            self.state_1_idx[row] = slots[k]
            self.action[row] = action
            self.reward[row] = reward
            self.terminal_mask[row] = 0.0 if k == n - 1 else 1.0                      # :101
            slots[k + 1] = self._pop_slot()                                          # :104
            self.state_2_idx[row] = slots[k + 1]
            rows[k], s1[k], s2[k] = row, slots[k], slots[k + 1]
            self.insert += 1
            if self.insert >= self.buffer_size:
                self.insert, self.full = 0, True
        # --- payload to HBM: n+1 states (cast to f16 = numpy's RNE, :32) and n event rows
        first = np.asarray(initial_state)
        if first.dtype == np.uint8:            # raw camera bytes: the device applies the /255 table
            first, dt = np.ascontiguousarray(first), _lib.CPP_U8
        else:
            first, dt = _lib.as_state_array(first)
        states = np.empty((n + 1, self.state_elems), first.dtype)
        states[0] = first.reshape(-1)
        for k in range(n):
            states[k + 1] = np.asarray(seq[k][2]).reshape(-1)
        if dt == _lib.CPP_F32:
            states, dt = states.astype(np.float16), _lib.CPP_F16
        check(lib.cpp_replay_write_states(self.handle, ptr(slots), n + 1, ptr(states), dt))
        check(lib.cpp_replay_write_rows(self.handle, ptr(rows), n, ptr(s1), ptr(s2),
                                        ptr(np.ascontiguousarray(self.action[rows])),
                                        ptr(np.ascontiguousarray(self.reward[rows])),
                                        ptr(np.ascontiguousarray(self.terminal_mask[rows]))))
        check(lib.cpp_replay_set_size(self.handle, self.size()))
        self._write_gen += 1

    def size(self):
        return self.buffer_size if self.full else self.insert

    def random_indexes(self, n=1):                                                   # :123-129
        if self.full:
            return np.random.randint(0, self.buffer_size, n)
        elif self.insert == 0:
            return []
        return np.random.randint(0, self.insert, n)

    def _device_batch(self, B):
        if B not in self._batches:
            self._batches[B] = DeviceBatch(B, self.state_elems, self.action_dim, self.ctx)
        prev = self._live.pop(B, None)
        prev = prev() if prev is not None else None
        if prev is not None:         # an earlier Batch of this size is still alive: it keeps its own draw
            prev._detach()
        return self._batches[B]

    def _new_batch(self, dev, idxs):
        b = Batch(dev, self.state_shape, idxs, self.state_1_idx[idxs], self.state_2_idx[idxs], memory=self)
        self._live[len(idxs)] = weakref.ref(b)
        return b

    def batch(self, batch_size=None, idxs=None):
        """replay_memory.py:131-138.  Rows come from numpy's global RNG exactly like the reference
        (`idxs=` overrides, as replay_memory_test.py:84 expects); the gather runs on the device."""
        self.stats[">batch"] += 1
        if idxs is None:
            idxs = self.random_indexes(batch_size)
        idxs = np.ascontiguousarray(np.asarray(idxs, dtype=np.int64).astype(np.int32))
        if len(idxs) == 0:
            return Batch.empty(self.state_shape, self.action_dim)
        dev = self._device_batch(len(idxs))
        check(lib.cpp_replay_sample(self.handle, len(idxs), ptr(idxs), 0, 0, self.channels, dev.handle))
        return self._new_batch(dev, idxs)

    def sample_on_device(self, batch_size, seed=0, counter=None):
        """Device-side Philox draw (no host RNG, no host copies) -- the sampler of the fused train step, for inspection:
        it has its own device counter word, so calling it between train steps does not rewind or advance the training
        sampler.  counter=None: an auto-incrementing host count (successive calls draw different rows); an explicit
        counter reproduces a draw."""
        self.stats[">batch"] += 1
        if counter is None:
            counter = self._adhoc_counter
            self._adhoc_counter += 1
        dev = self._device_batch(int(batch_size))
        check(lib.cpp_replay_sample(self.handle, int(batch_size), None, int(seed), int(counter),
                                    self.channels, dev.handle))
        idxs = np.empty(int(batch_size), np.int32)
        check(lib.cpp_replay_last_indexes(self.handle, int(batch_size), ptr(idxs)))
        return self._new_batch(dev, idxs)

    def fill_synthetic(self, n_rows, seed=1234):
        """bench/test helper: synthetic transitions generated on the device.  The host
        bookkeeping is advanced to match (fixed 50-step episodes, chain slot layout)."""
        n_rows = int(n_rows)
        check(lib.cpp_replay_fill_synthetic(self.handle, n_rows, int(seed)))
        self._write_gen += 1
        i = np.arange(n_rows)
        self.state_1_idx[:n_rows] = i + i // 50
        self.state_2_idx[:n_rows] = i + i // 50 + 1
        self.reward[:n_rows] = 1.0
        self.terminal_mask[:n_rows, 0] = np.where(i % 50 == 49, 0.0, 1.0)
        used = n_rows + n_rows // 50 + 1
        self.state_free_slots = collections.deque(range(used, self.state_buffer_size))
        self.insert, self.full = (0, True) if n_rows == self.buffer_size else (n_rows, False)

    def dump(self):
        print(">>>> dump")
        print("insert", self.insert)
        print("full?", self.full)
        print("state free slots", len(self.state_free_slots))
        if self.insert == 0 and not self.full:
            print("EMPTY!")
        else:
            for idx in range(self.size()):
                print("idx", idx, "state_1_idx", self.state_1_idx[idx], "action", self.action[idx],
                      "reward", self.reward[idx], "terminal_mask", self.terminal_mask[idx],
                      "state_2_idx", self.state_2_idx[idx])
        print("<<<< dump")

    def current_stats(self):                                                         # :160-163
        out = dict(self.stats)
        out["free_slots"] = len(self.state_free_slots)
        return out

    def reset_from_event_log(self, log_file):
        """prime the memory from an event log (replay_memory.py:40-61): the first event of an episode carries only
        the initial state, every later one (action, reward, state_2); stop once the memory is full."""
        import sys
        import time
        from . import event_log
        elr = event_log.EventLogReader(log_file)
        num_episodes = num_events = 0
        start = time.time()
        for episode in elr.entries():
            initial_state, seq = None, []
            for event_id, event in enumerate(episode.event):
                if event_id == 0:
                    assert len(event.action) == 0
                    assert not event.HasField("reward")
                    initial_state = event_log.read_state_from_event(event)
                else:
                    seq.append((np.asarray(event.action, np.float32), event.reward,
                                event_log.read_state_from_event(event)))
                num_events += 1
            num_episodes += 1
            self.add_episode(initial_state, seq)
            if self.full:
                break
        sys.stderr.write("reset_from_event_log \"%s\" num_episodes=%d num_events=%d took %s sec\n"
                         % (log_file, num_episodes, num_events, time.time() - start))

    def close(self):
        for b in self._batches.values():
            b.close()
        self._batches = {}
        if self.handle:
            lib.cpp_replay_destroy(self.handle)
            self.handle = None
