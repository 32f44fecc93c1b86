"""GPU tests of the device-resident replay memory against the reference's own known answers
(replay_memory_test.py, values typed in as data) and against the oracle replay.  Bit-exact."""
import numpy as np
import pytest

from oracle.replay_np import OracleReplayMemory
from tests.test_oracle_replay import s_for, soak

pytestmark = pytest.mark.gpu


def make(**kw):
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    args = dict(buffer_size=3, state_shape=(2, 3), action_dim=2, load_factor=2)
    args.update(kw)
    return ReplayMemory(**args)


def test_empty_memory():            # replay_memory_test.py:19-30
    rm = make()
    assert rm.size() == 0
    assert list(rm.random_indexes()) == []
    b = rm.batch(4)
    assert len(b) == 5
    for i in range(5):
        assert len(b[i]) == 0
    assert rm.insert == 0 and rm.full is False
    rm.close()


def test_adds_to_full():            # replay_memory_test.py:32-56
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
    rm.close()


def test_adds_over_full():          # replay_memory_test.py:58-86
    rm = make()
    rm.add_episode(s_for(0), [((i * 10) + 7, (i * 10) + 8, s_for(i)) for i in range(1, 5)])
    rm.add_episode(s_for(5), [((i * 10) + 7, (i * 10) + 8, s_for(i)) for i in range(6, 9)])
    assert rm.size() == 3
    assert sorted(set(rm.random_indexes(n=100))) == [0, 1, 2]
    b = rm.batch(idxs=[0, 1, 2])
    assert np.array_equal(b.reward, [[88], [68], [78]])
    assert np.array_equal(b.terminal_mask, [[0], [1], [1]])
    assert np.array_equal(b.state_1[:, 0, 0], [71, 51, 61]) and np.array_equal(b.state_2[:, 0, 0], [81, 61, 71])
    rm.close()


def test_soak_invariant():          # replay_memory.py:166-200
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    np.random.seed(3)
    rm = soak(ReplayMemory, 120)
    assert rm.current_stats()[">add_episode"] == 120
    rm.close()


def test_gather_and_statistics_match_oracle_bitwise():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    shape = (16, 16, 3, 2, 3)
    rng = np.random.default_rng(4)
    rm, orm = ReplayMemory(40, shape, 2), OracleReplayMemory(40, shape, 2)
    for _ in range(14):
        n = int(rng.integers(1, 8))
        mk = lambda: rng.uniform(0, 1, shape).astype(np.float32)       # f32 in: cast to f16 on store (RNE)
        s0, seq = mk(), [(rng.uniform(-1, 1, (1, 2)), float(rng.integers(0, 9)), mk()) for _ in range(n)]
        rm.add_episode(s0, seq); orm.add_episode(s0, seq)
    assert np.array_equal(rm.state_1_idx, orm.state_1_idx) and np.array_equal(rm.state_2_idx, orm.state_2_idx)
    assert list(rm.state_free_slots) == list(orm.state_free_slots)
    idxs = rng.integers(0, 40, 33)
    got, want = rm.batch(idxs=idxs), orm.batch(idxs=idxs)
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and np.array_equal(g, w)
    assert np.array_equal(got.state_1_idx, orm.state_1_idx[idxs])
    rm.close()


def _philox4x32_10(ctr, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def test_device_philox_rows_are_the_published_philox_and_uniform():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rm = ReplayMemory(1000, (4, 2), 2)
    rm.fill_synthetic(777, seed=5)
    B, seed, counter = 256, 0x1234567890, 42
    b = rm.sample_on_device(B, seed=seed, counter=counter)
    want = [(_philox4x32_10([i, 0, counter & 0xFFFFFFFF, counter >> 32],
                            [seed & 0xFFFFFFFF, seed >> 32])[0] * 777) >> 32 for i in range(B)]
    assert np.array_equal(b.idxs, want)
    assert b.idxs.min() >= 0 and b.idxs.max() < 777 and len(set(b.idxs.tolist())) > 150
    # the gathered rows are the rows of those indexes
    assert np.array_equal(b.reward, rm.reward[b.idxs]) and np.array_equal(b.terminal_mask, rm.terminal_mask[b.idxs])
    s1 = rm.state[rm.state_1_idx[b.idxs]]
    assert np.array_equal(b.state_1, s1)
    rm.close()


def test_reset_from_event_log_primes_the_device_memory(tmp_path):
    """--event-log-in (replay_memory.py:40-61): episodes written by EventLog land in HBM exactly like add_episode."""
    from cartpoleplusplus_amd import event_log as E
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rng = np.random.default_rng(9)
    shape = (16, 16, 3, 1, 2)
    path = str(tmp_path / "events")
    log = E.EventLog(path, use_raw_pixels=True)
    orm = OracleReplayMemory(12, shape, 2)
    for ep in range(4):
        log.reset()
        n = int(rng.integers(2, 5))
        fr = [(rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)).astype(np.float32) for _ in range(n + 1)]
        acts = [rng.uniform(-1, 1, (1, 2)).astype(np.float32) for _ in range(n)]
        log.add_just_state(fr[0])
        for k in range(n):
            log.add(fr[k + 1], acts[k], 1.0)
        if not orm.full:
            orm.add_episode(fr[0], [(acts[k], 1.0, fr[k + 1]) for k in range(n)])
    log.close()
    rm = ReplayMemory(12, shape, 2)
    rm.reset_from_event_log(path)
    assert rm.size() == orm.size() and rm.full == orm.full
    idxs = np.arange(rm.size())
    got, want = rm.batch(idxs=idxs), orm.batch(idxs=idxs)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    rm.close()


# ---- 8-bit store (the reference's own TODO, bullet_cartpole.py:237-239: "could just store this as uint8") -------------
def _render(rng, shape):
    """A frame exactly as the reference renders it: uint8 codes -> float16, /= 255 (bullet_cartpole.py:239-243)."""
    f = rng.integers(0, 256, shape).astype(np.float16)
    f /= 255
    return f


def test_u8_store_is_bit_identical_to_the_f16_store_for_rendered_frames():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    shape = (16, 16, 3, 2, 3)
    rng = np.random.default_rng(21)
    a, b = ReplayMemory(40, shape, 2), ReplayMemory(40, shape, 2, store_dtype="u8")
    orm = OracleReplayMemory(40, shape, 2)
    for ep in range(14):
        n = int(rng.integers(1, 8))
        cast = (lambda x: x) if ep % 2 else (lambda x: x.astype(np.float32))        # f16 and f32 payloads both
        s0 = cast(_render(rng, shape))
        seq = [(rng.uniform(-1, 1, (1, 2)), float(rng.integers(0, 9)), cast(_render(rng, shape))) for _ in range(n)]
        for m in (a, b, orm):
            m.add_episode(s0, seq)
    idxs = rng.integers(0, 40, 33)
    ga, gb, want = a.batch(idxs=idxs), b.batch(idxs=idxs), orm.batch(idxs=idxs)
    for x, y, w in zip(ga, gb, want):
        assert x.dtype == y.dtype == w.dtype and np.array_equal(x, y) and np.array_equal(y, w)
    slots = np.arange(a.state_buffer_size)
    used = sorted(set(slots) - set(a.state_free_slots))
    assert np.array_equal(a.state[used], b.state[used])
    # device-side sampling + the fused whitening statistics: same rows, same bits
    sa, sb = a.sample_on_device(32, seed=9, counter=3), b.sample_on_device(32, seed=9, counter=3)
    assert np.array_equal(sa.idxs, sb.idxs)
    assert np.array_equal(sa.state_1, sb.state_1) and np.array_equal(sa.state_2, sb.state_2)
    a.close(); b.close()


def test_raw_camera_bytes_take_the_reference_conversion_on_the_device():
    """uint8 frames handed over as they leave the renderer: both stores must hold what `f16(codes); /= 255` gives."""
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    shape = (8, 8, 3, 1, 2)
    rng = np.random.default_rng(5)
    codes = [rng.integers(0, 256, shape).astype(np.uint8) for _ in range(4)]
    codes[0].reshape(-1)[:256] = np.arange(256)              # every table entry
    want = []
    for c in codes:
        f = c.astype(np.float16); f /= 255
        want.append(f)
    for store in ("f16", "u8"):
        rm = ReplayMemory(6, shape, 2, store_dtype=store)
        rm.add_episode(codes[0], [(np.zeros((1, 2)), 1.0, codes[k]) for k in range(1, 4)])
        assert np.array_equal(rm.state[[0, 1, 2, 3]], np.stack(want)), store
        rm.close()


def test_u8_store_refuses_states_that_are_not_pixel_images():
    from cartpoleplusplus_amd import _lib
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rm = ReplayMemory(6, (8, 8, 3, 1, 2), 2, store_dtype="u8")
    rng = np.random.default_rng(6)
    bad = rng.uniform(0, 1, (8, 8, 3, 1, 2)).astype(np.float32)
    with pytest.raises(RuntimeError, match="pixel images only"):
        rm.add_episode(bad, [(np.zeros((1, 2)), 1.0, bad)])
    rm.close()
    with pytest.raises(RuntimeError, match="state_elems % 8"):
        ReplayMemory(6, (7,), 2, store_dtype="u8")


def test_u8_synthetic_fill_matches_the_f16_fill():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    shape = (16, 16, 3, 2, 2)
    a, b = ReplayMemory(64, shape, 2), ReplayMemory(64, shape, 2, store_dtype="u8")
    a.fill_synthetic(64, seed=11); b.fill_synthetic(64, seed=11)
    sa, sb = a.sample_on_device(48, seed=2, counter=1), b.sample_on_device(48, seed=2, counter=1)
    assert np.array_equal(sa.idxs, sb.idxs) and np.array_equal(sa.state_1, sb.state_1) and np.array_equal(sa.state_2, sb.state_2)
    a.close(); b.close()


def test_two_batches_of_one_size_are_independent_snapshots():
    """replay_memory.py:134-138 returns np.copy snapshots: a Batch must keep ITS draw when the next batch() of the same size
    reuses the shared device buffer -- whether its columns were read before or only after the second draw."""
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rng = np.random.default_rng(3)
    rm = ReplayMemory(60, (4, 4, 3, 1, 2), 2)
    orm = OracleReplayMemory(60, (4, 4, 3, 1, 2), 2)
    for _ in range(8):
        n = int(rng.integers(3, 8))
        mk = lambda: rng.integers(0, 256, (4, 4, 3, 1, 2)).astype(np.float16) / np.float16(255)
        s0, seq = mk(), [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.integers(0, 9)), mk()) for _ in range(n)]
        rm.add_episode(s0, seq); orm.add_episode(s0, seq)
    i1, i2 = rng.integers(0, rm.size(), 13), rng.integers(0, rm.size(), 13)
    b1 = rm.batch(idxs=i1)
    early = b1.reward.copy()
    b2 = rm.batch(idxs=i2)
    assert b2.device is not None                  # b2's draw is gathered into the shared device buffer ...
    b3 = rm.batch(idxs=i1)
    assert b3.device is not None                  # ... which b3 takes over before b2 has been read
    for b, idx in ((b1, i1), (b2, i2), (b3, i1)):
        ob = orm.batch(idxs=idx)
        for f in ("state_1", "action", "reward", "terminal_mask", "state_2"):
            assert np.array_equal(getattr(b, f), getattr(ob, f)), f
        assert np.array_equal(b.idxs, idx)
    assert np.array_equal(early, b1.reward)
    # Batches that are alive and unread when the memory is written keep their draw (np.copy semantics, replay_memory.py:134-138):
    # the last one by a device-side gather into the minibatch buffer, an earlier one of the same size by a download
    b4 = rm.batch(idxs=i1)
    b5 = rm.batch(idxs=i2)
    o4, o5 = orm.batch(idxs=i1), orm.batch(idxs=i2)
    rm.add_episode(s0, seq); orm.add_episode(s0, seq)
    assert not b4.in_replay() and not b5.in_replay()
    assert (b4._states is None) != (b5._states is None)      # one lives on in the device buffer, the other on the host
    for b, ob in ((b4, o4), (b5, o5)):
        for f in ("state_1", "action", "reward", "terminal_mask", "state_2"):
            assert np.array_equal(getattr(b, f), getattr(ob, f)), f
    b6 = rm.batch(idxs=i1)
    o6 = orm.batch(idxs=i1)
    rm.close()                                    # closing the memory moves unread draws to the host
    assert np.array_equal(np.asarray(b6.state_2), o6.state_2)


def test_inspection_draws_do_not_move_the_training_sampler():
    """sample_on_device between train steps must neither rewind nor advance the sampler of cpp_ddpg_train_step, and
    successive inspection draws differ (auto-incrementing counter) unless a counter is given."""
    import ctypes
    from cartpoleplusplus_amd import _lib
    from tests.helpers import make_pair
    shape, B = (16, 16, 3, 1, 2), 16
    seen = {}
    for peek in (False, True):
        agent, _ref, _ = make_pair(shape, B, True, replay_size=300)
        try:
            rm = agent.replay_memory
            rm.fill_synthetic(250, seed=5)
            rows = []
            for _ in range(4):
                agent.train_step(B, 2)
                idxs = np.empty(B, np.int32)
                _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
                rows.append(idxs.copy())
                if peek:
                    a, b = rm.sample_on_device(B, seed=9), rm.sample_on_device(B, seed=9)
                    assert not np.array_equal(a.idxs, b.idxs)
                    assert np.array_equal(rm.sample_on_device(B, seed=9, counter=0).idxs, rm.sample_on_device(B, seed=9, counter=0).idxs)
            seen[peek] = (np.concatenate(rows), agent.actor.get_params())
        finally:
            agent.close()
    assert np.array_equal(seen[False][0], seen[True][0]) and np.array_equal(seen[False][1], seen[True][1])


def test_failed_add_episode_leaves_the_memory_unchanged():
    """a write the device refuses (a non-image state offered to the 8-bit store) must not leave the host bookkeeping
    (insert, free list, index columns) ahead of the device."""
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rng = np.random.default_rng(1)
    shape = (4, 4, 3, 1, 2)
    rm = ReplayMemory(20, shape, 2, store_dtype="u8")
    mk = lambda: rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)
    good = [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), 1.0, mk()) for _ in range(5)]
    rm.add_episode(mk(), good)
    before = (rm.insert, rm.full, list(rm.state_free_slots), rm.state_1_idx.copy(), rm.state_2_idx.copy(), rm.size())
    bad = list(good)
    bad[3] = (bad[3][0], 1.0, np.full(shape, 0.123, np.float16))          # not f16(k/255)
    with pytest.raises(RuntimeError):
        rm.add_episode(mk(), bad)
    assert (rm.insert, rm.full, list(rm.state_free_slots), rm.size()) == (before[0], before[1], before[2], before[5])
    assert np.array_equal(rm.state_1_idx[:5], before[3][:5]) and np.array_equal(rm.state_2_idx[:5], before[4][:5])
    rm.add_episode(mk(), good)                     # and the memory keeps working
    assert rm.size() == 10
    b = rm.batch(idxs=np.arange(10))
    assert np.array_equal(b.state_2[:4], np.stack([g[2] for g in good[:4]]))
    rm.close()


@pytest.mark.parametrize("buffer_size,load_factor,seed", [(3, 2.0, 0), (7, 1.5, 1), (20, 1.5, 2), (43, 1.5, 3), (5, 3.0, 4)])
def test_random_episodes_differential_against_the_oracle_memory(buffer_size, load_factor, seed):
    """random episode streams (lengths 1..12, so longer than the small buffers: several wraps inside one add_episode) into the device
    memory and into oracle/replay_np.py (the line-by-line restatement of replay_memory.py:63-118): after every episode the host
    mirrors -- insert, full, both index columns, action / reward / mask, the FIFO of free state slots -- and the batch of ALL rows
    gathered on the device must be identical."""
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    rng = np.random.default_rng(100 + seed)
    shape, A = (4, 4, 3, 1, 2), 2
    rm = ReplayMemory(buffer_size=buffer_size, state_shape=shape, action_dim=A, load_factor=load_factor)
    orm = OracleReplayMemory(buffer_size, shape, A, load_factor=load_factor)
    try:
        for ep in range(60):
            n = int(rng.integers(1, 13))
            mk = lambda: (rng.integers(0, 256, shape).astype(np.float16) / np.float16(255))
            s0 = mk()
            seq = [(rng.uniform(-1, 1, (1, A)).astype(np.float32), float(rng.integers(-3, 4)), mk()) for _ in range(n)]
            try:
                orm.add_episode(s0, seq)
            except (AssertionError, IndexError):
                # the reference fails (assert / pop from the empty free list, replay_memory.py:65,77-79,104) when an episode needs
                # more state slots than are free -- half way through, so its own state is no longer comparable.  The device memory
                # must refuse the episode too (and, unlike the reference, stays as it was: test_failed_add_episode_...)
                with pytest.raises(Exception):
                    rm.add_episode(s0, seq)
                break
            rm.add_episode(s0, seq)
            assert (rm.insert, rm.full, rm.size()) == (orm.insert, orm.full, orm.size())
            k = orm.size()
            assert np.array_equal(rm.state_1_idx[:k], orm.state_1_idx[:k]) and np.array_equal(rm.state_2_idx[:k], orm.state_2_idx[:k])
            assert np.array_equal(rm.action[:k], orm.action[:k]) and np.array_equal(rm.reward[:k], orm.reward[:k])
            assert np.array_equal(rm.terminal_mask[:k], orm.terminal_mask[:k])
            assert list(rm.state_free_slots) == list(orm.state_free_slots)
            idxs = np.arange(k)
            b, ob = rm.batch(idxs=idxs), orm.batch(idxs=idxs)
            for x, y in zip((b.state_1, b.action, b.reward, b.terminal_mask, b.state_2), (ob.state_1, ob.action, ob.reward, ob.terminal_mask, ob.state_2)):
                assert np.array_equal(np.asarray(x), np.asarray(y))
    finally:
        rm.close()


def test_stored_whitening_sums_equal_a_pass_over_the_pixels():
    """cpp_replay_set_stats_channels: the store keeps sum(x), sum(x^2) per state and channel; a sampled minibatch's whitening tables
    built from those rows (the fused steps' sample pass) must be bit for bit the tables of a gather that reads the images --
    for states written by add_episode (f16 and camera bytes), by the synthetic fill, and in the 8-bit store."""
    import ctypes as C
    from cartpoleplusplus_amd import _lib, ddpg_cartpole as D
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    from tests.helpers import FakeEnv, make_opts
    shape, B = (16, 16, 3, 2, 3), 16
    rng = np.random.default_rng(8)
    for store in ("f16", "u8"):
        make_opts(D, shape, B, True, replay_memory_size=120, replay_store=store)
        agent = D.DeepDeterministicPolicyGradientAgent(FakeEnv(shape))
        agent.initialise_variables(seed=1); agent.post_var_init_setup()
        rm = agent.replay_memory
        rm.fill_synthetic(60, seed=3)
        for ep in range(3):          # episodes over the synthetic rows: f16 frames, then raw camera bytes
            mk = (lambda: rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)) if ep < 2 else (lambda: rng.integers(0, 256, shape).astype(np.uint8))
            rm.add_episode(mk(), [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), 1.0, mk()) for _ in range(7)])
        idx = rng.integers(0, rm.size(), 2 * B).astype(np.int32)

        def params_after(stats_on):
            a2 = D.DeepDeterministicPolicyGradientAgent(FakeEnv(shape))
            a2.initialise_variables(seed=1); a2.post_var_init_setup()
            _lib.check(_lib.lib.cpp_replay_set_stats_channels(rm.handle, rm.channels if stats_on else 0))
            _lib.check(_lib.lib.cpp_ddpg_train_step(a2.trainer.handle, rm.handle, B, 2, idx.ctypes.data_as(C.c_void_p), 0))
            out = [n.get_params() for n in a2.networks()]
            a2.close()
            return out
        with_sums, from_pixels = params_after(True), params_after(False)
        for x, y in zip(with_sums, from_pixels):
            assert np.array_equal(x, y), store
        agent.close()
