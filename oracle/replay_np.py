"""Host-numpy restatement of the reference replay memory.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/replay_memory.py (HEAD) statement by statement:

  constructor / column dtypes ........ replay_memory.py:11-38
  add_episode ........................ replay_memory.py:63-71
  _add (evict, write row, pop slot) .. replay_memory.py:73-118
  size ............................... replay_memory.py:120-121
  random_indexes ..................... replay_memory.py:123-129
  batch .............................. replay_memory.py:131-138
  current_stats ...................... replay_memory.py:160-163

Differences that are not behavioural: Python 3, `collections.deque` for the FIFO
free-slot list (the reference uses list.pop(0)/append -- same order), no event-log
priming, `batch(idxs=...)` accepted as the reference's own (stale) test expects
(replay_memory_test.py:84).  Pinned by tests/test_oracle_replay.py against the
reference's known answers.
"""
import collections

import numpy as np

OracleBatch = collections.namedtuple(
    "OracleBatch", "state_1 action reward terminal_mask state_2")


class OracleReplayMemory(object):
    def __init__(self, buffer_size, state_shape, action_dim, load_factor=1.5):
        # replay_memory.py:13
        assert load_factor >= 1.5, "load_factor has to be at least 1.5"
        self.buffer_size = int(buffer_size)
        self.state_shape = tuple(state_shape)
        self.insert = 0
        self.full = False
        n = self.buffer_size
        # replay_memory.py:21-25 -- one row per event
        self.state_1_idx = np.empty(n, dtype=np.int32)
        self.action = np.empty((n, action_dim), dtype=np.float32)
        self.reward = np.empty((n, 1), dtype=np.float32)
        self.terminal_mask = np.empty((n, 1), dtype=np.float32)
        self.state_2_idx = np.empty(n, dtype=np.int32)
        # replay_memory.py:30-32 -- de-duplicated f16 state store
        self.state_buffer_size = int(n * load_factor)
        self.state = np.empty((self.state_buffer_size,) + self.state_shape,
                              dtype=np.float16)
        # replay_memory.py:35 -- FIFO of free state slots
        self.state_free_slots = collections.deque(range(self.state_buffer_size))
        self.stats = collections.Counter()

    # replay_memory.py:63-71
    def add_episode(self, initial_state, action_reward_state_sequence):
        self.stats[">add_episode"] += 1
        seq = list(action_reward_state_sequence)
        assert len(seq) > 0
        s1_idx = self.state_free_slots.popleft()
        self.state[s1_idx] = initial_state
        last = len(seq) - 1
        for n, (action, reward, state_2) in enumerate(seq):
            s1_idx = self._add(s1_idx, action, reward, n == last, state_2)

    # replay_memory.py:73-118
    def _add(self, s1_idx, a, r, t, s2):
        self.stats[">add"] += 1
        assert 0 <= s1_idx < self.state_buffer_size, s1_idx
        row = self.insert
        if self.full:
            # :84 the row being clobbered always gives back its state_1 slot ...
            self.state_free_slots.append(int(self.state_1_idx[row]))
            # :89-91 ... and its state_2 slot too when it ended an episode
            if self.terminal_mask[row] == 0:
                self.state_free_slots.append(int(self.state_2_idx[row]))
                self.stats["cache_evicted_s2"] += 1
        self.state_1_idx[row] = s1_idx            # :94
        self.action[row] = a                      # :95 (broadcasts (1,A) rows)
        self.reward[row] = r                      # :96
        self.terminal_mask[row] = 0.0 if t else 1.0   # :101
        s2_idx = self.state_free_slots.popleft()  # :104
        self.state_2_idx[row] = s2_idx
        self.state[s2_idx] = s2                   # :106 (f32 -> f16, RNE)
        self.insert += 1                          # :109-112
        if self.insert >= self.buffer_size:
            self.insert = 0
            self.full = True
        return s2_idx

    def size(self):
        return self.buffer_size if self.full else self.insert

    # replay_memory.py:123-129
    def random_indexes(self, n=1):
        if self.full:
            return np.random.randint(0, self.buffer_size, n)
        if self.insert == 0:
            return []
        return np.random.randint(0, self.insert, n)

    # replay_memory.py:131-138 (+ idxs= of replay_memory_test.py:84)
    def batch(self, batch_size=None, idxs=None):
        self.stats[">batch"] += 1
        if idxs is None:
            idxs = self.random_indexes(batch_size)
        idxs = np.asarray(idxs, dtype=np.int64)
        return OracleBatch(np.copy(self.state[self.state_1_idx[idxs]]),
                           np.copy(self.action[idxs]),
                           np.copy(self.reward[idxs]),
                           np.copy(self.terminal_mask[idxs]),
                           np.copy(self.state[self.state_2_idx[idxs]]))

    # replay_memory.py:160-163
    def current_stats(self):
        out = dict(self.stats)
        out["free_slots"] = len(self.state_free_slots)
        return out
