"""Device-resident replay memory with the interface of the reference's replay_memory.py.

Reference surface kept (paths relative to /root/reference): `Batch` (replay_memory.py:9) and
`ReplayMemory(buffer_size, state_shape, action_dim, load_factor=1.5)` with `add_episode`, `size`,
`random_indexes`, `batch`, `current_stats`, `dump` and the public attributes `insert`, `full`,
`state_1_idx`, `state_2_idx`, `action`, `reward`, `terminal_mask`, `state`, `state_free_slots`,
`stats` (replay_memory.py:11-163), plus `batch(idxs=...)` / `Batch.state_1_idx` that the reference's
own test still expects (replay_memory_test.py:84,129).

What is different underneath: the f16 state store (replay_memory.py:32) lives in HBM behind
cpp_replay_*; only the slot bookkeeping (FIFO free list, eviction -- replay_memory.py:66,84,90,104)
runs on the host so its order is exactly the reference's.  `batch()` draws the rows (numpy's RNG, as the
reference) and returns a Batch whose two state columns are `StateColumn`s: still in HBM, downloaded only
if numpy reads them, and recognised by the agents' train ops, which run the fused sample + gather +
update sequence on the device rows (the reference's literal loop ddpg_cartpole.py:331-334 moves B row
indexes over PCIe per minibatch, not 75 MB of pixels).
"""
import collections
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import lib, check, ptr

_FIELDS = ("state_1", "action", "reward", "terminal_mask", "state_2")


def _delegate(name):
    def op(self, *args):
        return getattr(self._array(), name)(*args)
    op.__name__ = name
    return op


class StateColumn(object):
    """`batch.state_1` / `batch.state_2`: the (B, *state_shape) f16 column of a Batch, still resident in HBM.

    The reference's Batch holds host copies (replay_memory.py:134-138) that its training loop hands straight back to the device
    (`actor.train(batch.state_1)`, ddpg_cartpole.py:333 -- 2 x 37.7 MB per minibatch at 64x64x18).  This column has the array's
    `shape` / `dtype` / `len`, converts on demand (`np.asarray(col)`, indexing, arithmetic, any ndarray attribute: ONE download, then
    cached), and is recognised by the network classes, which take its Batch's device rows instead: nothing crosses PCIe unless
    numpy actually looks at the pixels."""
    __slots__ = ("_batch", "_field")
    __array_priority__ = 100.0

    def __init__(self, batch, field):
        self._batch, self._field = batch, field

    @property
    def batch(self):
        return self._batch

    @property
    def field(self):
        return self._field

    @property
    def shape(self):
        return (len(self._batch.idxs),) + self._batch._shape

    dtype = np.dtype(np.float16)

    @property
    def ndim(self):
        return 1 + len(self._batch._shape)

    @property
    def size(self):
        return int(np.prod(self.shape))

    def _array(self):
        return self._batch._host_states()[self._field]

    def __array__(self, dtype=None, copy=None):
        a = self._array()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __len__(self):
        return len(self._batch.idxs)

    def __getattr__(self, name):          # any other ndarray attribute / method (astype, reshape, T, mean, ...)
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._array(), name)

    def __repr__(self):
        return "<StateColumn %s %s f16 of a device-resident Batch>" % (self._field, self.shape)


for _n in ("__getitem__", "__iter__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__add__", "__radd__", "__sub__",
           "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__neg__", "__abs__", "__pow__", "__matmul__"):
    setattr(StateColumn, _n, _delegate(_n))
StateColumn.__hash__ = None


class Batch(object):
    """replay_memory.py:9 `Batch = namedtuple("Batch", "state_1 action reward terminal_mask state_2")`.
    Tuple protocol preserved (len 5, indexing, iteration, attribute names).  A Batch is a DRAW: the row indexes, the state slots
    they pointed to and host copies of the three small columns, all taken at batch() time.  The two state columns stay in the
    replay store (`StateColumn`) until somebody reads them; `device` gathers the draw into a device minibatch for the train ops
    that want one.  A Batch that is still alive and unread when the memory is about to be written (add_episode) or closed is
    preserved first -- a device-to-device gather into the minibatch buffer of its size, a download only when that buffer is needed
    by a second such Batch -- so it keeps the np.copy semantics of replay_memory.py:134-138 whatever the caller does with it."""
    _fields = _FIELDS

    def __init__(self, memory, state_shape, idxs, s1_idx, s2_idx, small):
        self._memory, self._shape = memory, tuple(state_shape)
        self.idxs, self.state_1_idx, self.state_2_idx = idxs, s1_idx, s2_idx
        self._small = small                       # {"action", "reward", "terminal_mask"}: host arrays this Batch owns
        self._states = None                       # {"state_1", "state_2"} once downloaded
        self._gen = memory._write_gen if memory is not None else None

    @classmethod
    def empty(cls, state_shape, action_dim):
        b = cls(None, state_shape, np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.int32),
                {"action": np.empty((0, action_dim), np.float32), "reward": np.empty((0, 1), np.float32),
                 "terminal_mask": np.empty((0, 1), np.float32)})
        b._states = {"state_1": np.empty((0,) + tuple(state_shape), np.float16),
                     "state_2": np.empty((0,) + tuple(state_shape), np.float16)}
        return b

    def in_replay(self):
        """True while the rows of this draw are still what the replay memory holds (no write since batch())."""
        rm = self._memory
        return rm is not None and rm.handle is not None and rm._write_gen == self._gen

    def _preserve(self, to_host=False):
        """called by the memory before the rows of this draw can change: keep the draw readable (see the class docstring)."""
        if self._states is not None or len(self.idxs) == 0:
            return
        if to_host:
            self._host_states()
            return
        rm = self._memory
        dev = rm._device_batch(len(self.idxs))
        other = dev.owner()
        if other is self:
            return
        if other is not None and other._states is None:
            other._host_states()                  # the buffer's current draw moves to the host, this one takes the buffer
        self.device                               # (gathers: this Batch owns the buffer from here on)

    def _gone(self):
        return RuntimeError("states have been written to the replay memory since this Batch was drawn and its state columns were "
                            "never read: they are gone (read a column, or finish with the Batch, before the next add_episode())")

    def _host_states(self):
        if self._states is None:
            rm = self._memory
            dev = rm._batches.get(len(self.idxs)) if rm is not None and rm.handle is not None else None
            if dev is not None and dev.owner() is self:          # gathered already: one copy of both columns
                B = len(self.idxs)
                s1, s2 = np.empty((B,) + self._shape, np.float16), np.empty((B,) + self._shape, np.float16)
                check(lib.cpp_batch_download(dev.handle, ptr(s1), ptr(s2), None, None, None))
                self._states = {"state_1": s1, "state_2": s2}
            elif self.in_replay():
                self._states = {"state_1": rm.state[self.state_1_idx], "state_2": rm.state[self.state_2_idx]}
            else:
                raise self._gone()
        return self._states

    def __getattr__(self, name):
        if name.startswith("_"):                  # (never resolve private names here: copy / pickle probe them before __init__ ran)
            raise AttributeError(name)
        if name in ("state_1", "state_2"):
            if self._states is not None:
                return self._states[name]
            # a fresh column per access: the column refers to its Batch, so a Batch that kept its columns would be a reference
            # cycle -- alive until a gc pass, found by the next add_episode() in ReplayMemory._drawn and preserved (a device gather,
            # and a 2 x 37.7 MB download for every such Batch but the last) although nobody can read it any more
            return StateColumn(self, name)
        if name in ("action", "reward", "terminal_mask"):
            return self._small[name]
        raise AttributeError(name)

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return getattr(self, _FIELDS[i])

    def __iter__(self):
        return (getattr(self, f) for f in _FIELDS)

    @property
    def device(self):
        """this draw as a DeviceBatch (None for the empty batch).  The device buffer is shared by all Batches of one size and
        remembers whose draw it holds: a Batch that is not the owner gathers (or uploads) its rows again."""
        rm = self._memory
        if rm is None or len(self.idxs) == 0:
            return None
        if rm.handle is None:
            raise self._gone()
        B = len(self.idxs)
        dev = rm._device_batch(B)
        if dev.owner() is not self:
            dev.evict_owner()
            if self.in_replay():
                check(lib.cpp_replay_sample(rm.handle, B, ptr(self.idxs), 0, 0, rm.channels, dev.handle))
            elif self._states is not None:
                dev.upload(self._states["state_1"], self._small["action"], self._small["reward"], self._small["terminal_mask"],
                           self._states["state_2"])
            else:
                raise self._gone()
            dev.set_owner(self)
        return dev


class DeviceBatch(object):
    """A minibatch resident in HBM (cpp_batch)."""

    def __init__(self, max_batch, state_elems, action_dim, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self.max_batch, self.state_elems, self.action_dim = int(max_batch), int(state_elems), int(action_dim)
        h = C.c_void_p()
        check(lib.cpp_batch_create(self.ctx.handle, self.max_batch, self.state_elems, self.action_dim, C.byref(h)))
        self.handle = h
        self._owner = None           # weakref of the Batch whose draw the buffer holds (replay_memory.Batch.device)

    def owner(self):
        return self._owner() if self._owner is not None else None

    def set_owner(self, batch):
        self._owner = weakref.ref(batch) if batch is not None else None

    def evict_owner(self):
        """the buffer is about to take another draw: if it is the ONLY copy of its current owner's states (a Batch preserved here when
        the memory was written, never read), that Batch moves to the host first."""
        o = self.owner()
        if o is not None and o._states is None and not o.in_replay():
            o._host_states()
        self._owner = None

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
        self._owner = None
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
        self._write_gen = 0          # bumped by every write of states (add_episode, fill_synthetic)
        self._drawn = weakref.WeakSet()   # Batches drawn since the last write (preserved before the next one, see Batch)
        self._adhoc_counter = 0      # sample_on_device draws (separate from the train steps' device counter)
        # pixel states (H, W, 3, cameras, repeats): channel count for the fused whitening statistics
        self.channels = int(np.prod(self.state_shape[2:])) if len(self.state_shape) == 5 else 0
        if self.channels > 0:        # per-state whitening sums, kept by the store: sampling never re-reads the pixels for them
            check(lib.cpp_replay_set_stats_channels(self.handle, self.channels))

    # --- host-side slot bookkeeping: exactly replay_memory.py:63-118 ---------------------------
    def _pop_slot(self):
        if not self.state_free_slots:
            raise RuntimeError("replay memory state store exhausted (load_factor too small for this "
                               "many short episodes; SURVEY appendix B9)")
        return self.state_free_slots.popleft()

    def add_episode(self, initial_state, action_reward_state_sequence):
        self._preserve_draws()
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
            self._write_gen += 1         # (a device write may have happened: Batches drawn before this call do not re-read the store)
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
        if self.store_dtype == "u8" and dt == _lib.CPP_F16:
            # the 8-bit store only holds f16(k/255) images: refuse anything else BEFORE a slot is written (the slots popped above may
            # be the ones this very episode's evictions freed -- a device-side refusal would come after they were overwritten)
            codes = (np.arange(256) / 255.0).astype(np.float16).view(np.uint16)
            if not np.isin(states.view(np.uint16), codes).all():
                raise RuntimeError("replay memory (8-bit store): pixel images only -- a state holds a value that is not f16(k/255)")
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
        return self._batches[B]

    def _new_batch(self, idxs):
        """the draw `idxs` as a Batch: slots and the three small columns are copied from the host mirrors now (np.copy semantics of
        replay_memory.py:134-138 for everything that is cheap), the states stay where they are."""
        small = {"action": self.action[idxs], "reward": self.reward[idxs], "terminal_mask": self.terminal_mask[idxs]}
        b = Batch(self, self.state_shape, idxs, self.state_1_idx[idxs], self.state_2_idx[idxs], small)
        self._drawn.add(b)
        return b

    def _preserve_draws(self, to_host=False):
        for b in list(self._drawn):
            if b.in_replay():
                b._preserve(to_host)
        self._drawn.clear()

    def batch(self, batch_size=None, idxs=None):
        """replay_memory.py:131-138.  Rows come from numpy's global RNG exactly like the reference (`idxs=` overrides, as
        replay_memory_test.py:84 expects).  No device work happens here: the returned Batch is the draw, its state columns are
        read (or trained on) where they lie."""
        self.stats[">batch"] += 1
        if idxs is None:
            idxs = self.random_indexes(batch_size)
        idxs = np.ascontiguousarray(np.asarray(idxs, dtype=np.int64).astype(np.int32))
        if len(idxs) == 0:
            return Batch.empty(self.state_shape, self.action_dim)
        if len(idxs) and (idxs.min() < 0 or idxs.max() >= max(self.size(), 1)):
            raise RuntimeError("batch: index outside [0,%d)" % self.size())
        return self._new_batch(idxs)

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
        dev.evict_owner()
        check(lib.cpp_replay_sample(self.handle, int(batch_size), None, int(seed), int(counter),
                                    self.channels, dev.handle))
        idxs = np.empty(int(batch_size), np.int32)
        check(lib.cpp_replay_last_indexes(self.handle, int(batch_size), ptr(idxs)))
        b = self._new_batch(idxs)
        dev.set_owner(b)
        return b

    def fill_synthetic(self, n_rows, seed=1234):
        """bench/test helper: synthetic transitions generated on the device.  The host bookkeeping is advanced to
        match (fixed 50-step episodes, chain slot layout) and the host mirrors of the event columns are read back."""
        n_rows = int(n_rows)
        self._preserve_draws()
        check(lib.cpp_replay_fill_synthetic(self.handle, n_rows, int(seed)))
        self._write_gen += 1
        rows = np.arange(n_rows, dtype=np.int32)
        s1, s2 = np.empty(n_rows, np.int32), np.empty(n_rows, np.int32)
        a, r, m = np.empty((n_rows, self.action_dim), np.float32), np.empty((n_rows, 1), np.float32), np.empty((n_rows, 1), np.float32)
        check(lib.cpp_replay_read_rows(self.handle, ptr(rows), n_rows, ptr(s1), ptr(s2), ptr(a), ptr(r), ptr(m)))
        self.state_1_idx[:n_rows], self.state_2_idx[:n_rows] = s1, s2
        self.action[:n_rows], self.reward[:n_rows], self.terminal_mask[:n_rows] = a, r, m
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
        if self.handle:
            self._preserve_draws(to_host=True)
        for b in self._batches.values():
            b.close()
        self._batches = {}
        if self.handle:
            lib.cpp_replay_destroy(self.handle)
            self.handle = None
