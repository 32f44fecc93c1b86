"""Pins oracle/replay_np.py to the reference's own known answers.

Known answers come from /root/reference/replay_memory_test.py (values typed in here as data):
  test_empty_memory  :19-30     test_adds_to_full :32-56     test_adds_over_full :58-86
and the soak invariant of /root/reference/replay_memory.py:166-200.
"""
import random

import numpy as np

from oracle.replay_np import OracleReplayMemory


def make():
    return OracleReplayMemory(buffer_size=3, state_shape=(2, 3), action_dim=2, load_factor=2)


def test_empty_memory():
    rm = make()
    assert rm.size() == 0
    assert list(rm.random_indexes()) == []
    b = rm.batch(4)
    assert len(b) == 5
    for col in b:
        assert len(col) == 0
    assert rm.insert == 0 and rm.full is False


def test_adds_to_full():
    rm = make()
    rm.add_episode([[11, 12, 13], [14, 15, 16]],
                   [(17, 18, [[21, 22, 23], [24, 25, 26]]),
                    (27, 28, [[31, 32, 33], [34, 35, 36]]),
                    (37, 38, [[41, 42, 43], [44, 45, 46]])])
    assert rm.size() == 3
    idxs = rm.random_indexes(n=100)
    assert len(idxs) == 100 and sorted(set(idxs)) == [0, 1, 2]
    assert rm.insert == 0 and rm.full is True
    for slot, first in enumerate([11, 21, 31, 41]):
        assert rm.state[slot][0][0] == first


def s_for(i):
    return (np.array(range(1, 7)) + (10 * i)).reshape(2, 3)


def test_adds_over_full():
    rm = make()
    rm.add_episode(s_for(0), [((i * 10) + 7, (i * 10) + 8, s_for(i)) for i in range(1, 5)])
    rm.add_episode(s_for(5), [((i * 10) + 7, (i * 10) + 8, s_for(i)) for i in range(6, 9)])
    assert rm.size() == 3
    idxs = rm.random_indexes(n=100)
    assert sorted(set(idxs)) == [0, 1, 2]
    b = rm.batch(idxs=[0, 1, 2])
    assert np.array_equal(b.reward, [[88], [68], [78]])
    assert np.array_equal(b.terminal_mask, [[0], [1], [1]])


def soak(rm_factory, episodes, seed=0):
    """replay_memory.py:166-200, terminated after `episodes` (the reference loops forever)."""
    rm = rm_factory(buffer_size=43, state_shape=(2, 3), action_dim=2)
    rnd = random.Random(seed)

    def s(i):
        i = (i * 10) % 199
        return [[i + 1, 0, 0], [0, 0, 0]]

    terminals, i = set(), 0
    for _ in range(episodes):
        initial = s(i)
        seq = []
        for _ in range(int(3 + rnd.random() * 5)):
            i += 1
            seq.append(((i, 0), i, s(i)))
        rm.add_episode(initial, seq)
        terminals.add(i)
        for _ in range(7):
            b = rm.batch(13)
            for k in range(3):
                r = int(b.reward[k][0])
                assert b.state_1[k][0][0] == (((r - 1) * 10) % 199) + 1
                assert b.action[k][0] == r
                assert b.terminal_mask[k] == (0 if r in terminals else 1)
                assert b.state_2[k][0][0] == ((r * 10) % 199) + 1
        i += 1
    return rm


def test_soak_invariant():
    np.random.seed(1)
    rm = soak(OracleReplayMemory, 400)
    st = rm.current_stats()
    assert st[">add_episode"] == 400 and st["cache_evicted_s2"] > 0
    # every slot is either free or referenced, never both
    used = set(rm.state_1_idx.tolist()) | set(rm.state_2_idx.tolist())
    assert not (used & set(rm.state_free_slots))


def test_f16_store_rounds_to_nearest_even():
    rm = OracleReplayMemory(4, (3,), 1, 1.5)
    x = np.array([1.0 + 2 ** -11, 1.0 + 3 * 2 ** -11, 0.1], np.float32)
    rm.add_episode(x, [([0.5], 1.0, x)])
    assert rm.state.dtype == np.float16
    assert np.array_equal(rm.state[0], x.astype(np.float16))
    assert float(rm.state[0][0]) == 1.0 and float(rm.state[0][1]) == 1.0 + 2 ** -9
