"""Replay stores at the capacities BASELINE.json's configs name, on one GPU's 288 GB of HBM: one GPU's shard of configs[4]
(125 000 rows of 128x128x30 states = 187 500 slots, 92 GB as 8-bit codes -- replay_memory.py:30-32 with the reference's own
"store uint8" TODO) and the exps/run_98.sh:7 memory (200 000 rows of 64x64x18 f16 states, 300 000 slots, 44 GB).  Everything
is sampled from the TOP of the store: slot offsets beyond 2^32 bytes in the gather kernel, in the per-slot read-back and in the
f16-pipe conv1 kernels that read the store through the sampled slots."""
import numpy as np
import pytest

from tests.helpers import make_pair, synthetic_state_codes, assert_flat_close

pytestmark = pytest.mark.gpu

LEVELS = np.arange(256).astype(np.float16) / np.float16(255)          # f16(k/255): bullet_cartpole.py:239-243


def test_cfg5_shard_as_8_bit_store_gathers_exactly_from_the_top():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    shape, rows, B, seed = (128, 128, 3, 2, 5), 125000, 512, 77
    rm = ReplayMemory(rows, shape, 2, store_dtype="u8")
    try:
        elems = rm.state_elems
        assert rm.state_buffer_size == 187500 and rm.state_buffer_size * elems > 90e9
        rm.fill_synthetic(rows, seed=seed)
        assert rm.size() == rows and rm.full
        idxs = np.concatenate([np.arange(rows - B + 8, rows), [0, 1, rows // 2, 49, 50, 99999, 65535, 65536]]).astype(np.int64)
        b = rm.batch(idxs=idxs)
        s1, s2 = b.state_1.reshape(B, -1), b.state_2.reshape(B, -1)
        # host-regenerated payload of a handful of the sampled states (two of them in the last slots of the store)
        for j in (0, B - 9, B - 8, B - 5, B - 1, 200):
            for col, slot in ((s1, int(rm.state_1_idx[idxs[j]])), (s2, int(rm.state_2_idx[idxs[j]]))):
                assert np.array_equal(col[j], LEVELS[synthetic_state_codes(slot, elems, seed)]), (j, slot)
        top_slot = int(rm.state_2_idx[rows - 1])
        assert top_slot * elems > 2 ** 35                                # byte offset far beyond 32 bits
        assert np.array_equal(rm.state[top_slot].reshape(-1), LEVELS[synthetic_state_codes(top_slot, elems, seed)])
        assert np.array_equal(b.terminal_mask[:, 0], rm.terminal_mask[idxs, 0])
        # the device sampler covers the whole shard
        seen = np.concatenate([rm.sample_on_device(B, seed=5).idxs for _ in range(4)])
        assert seen.min() >= 0 and seen.max() < rows and seen.max() > 0.9 * rows and seen.min() < 0.1 * rows
    finally:
        rm.close()


def test_cfg5_fused_step_trains_from_the_full_shard():
    """B = 512 fused steps (hipGraph, device Philox rows) on the 92 GB shard: same parameters as the same steps on caller-fed
    copies of the rows the sampler drew (the op-by-op path through a gathered batch)."""
    import ctypes
    from cartpoleplusplus_amd import _lib
    shape, rows, B = (128, 128, 3, 2, 5), 125000, 512
    agent, _ref, (aspec, cspec) = make_pair(shape, B, True, replay_size=rows, replay_store="u8")
    try:
        rm = agent.replay_memory
        rm.fill_synthetic(rows, seed=3)
        P0 = [n.get_params() for n in agent.networks()]
        agent.train_step(B, 1)
        idxs = np.empty(B, np.int32)
        _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
        assert idxs.max() > 100000
        fused = [n.get_params() for n in agent.networks()]
        for n, p in zip(agent.networks(), P0):
            n.set_params(p)
        batch = rm.batch(idxs=idxs)
        agent.actor.train(batch); agent.critic.train(batch)
        agent.target_actor.update_weights(); agent.target_critic.update_weights()
        unfused = [n.get_params() for n in agent.networks()]
        for spec, a, b in ((aspec, fused[0], unfused[0]), (cspec, fused[1], unfused[1]), (aspec, fused[2], unfused[2]), (cspec, fused[3], unfused[3])):
            assert_flat_close(spec, a, b, rel=1e-5, what="fused step on the shard vs train ops on the gathered rows")
            assert np.isfinite(a).all()
        assert np.abs(fused[1] - P0[1]).max() > 0
    finally:
        agent.close()


def test_run_98_memory_of_200000_rows_feeds_conv1_through_slots_at_the_top():
    """44 GB f16 store (exps/run_98.sh:7).  The fused step reads conv1's images straight from the store through the sampled
    slots (no gathered copy): rows at the top of the store against the op-by-op path on a gathered, bit-checked copy."""
    shape, rows, B, seed = (64, 64, 3, 2, 3), 200000, 256, 9
    agent, _ref, (aspec, cspec) = make_pair(shape, B, True, replay_size=rows)
    try:
        rm = agent.replay_memory
        elems = rm.state_elems
        assert rm.state_buffer_size == 300000 and rm.state_buffer_size * elems * 2 > 44e9
        rm.fill_synthetic(rows, seed=seed)
        idxs = np.concatenate([np.arange(rows - B + 4, rows), [0, 7, 123456, 65536]]).astype(np.int32)
        batch = rm.batch(idxs=idxs)
        for j in (0, B - 5, B - 1):
            slot = int(rm.state_2_idx[idxs[j]])
            assert slot * elems * 2 > 2 ** 32 or j == B - 1
            assert np.array_equal(batch.state_2.reshape(B, -1)[j], LEVELS[synthetic_state_codes(slot, elems, seed)])
        P0 = [n.get_params() for n in agent.networks()]
        agent.train_step(B, 1, idxs=idxs)                                 # conv1 reads the store through the slots
        fused = [n.get_params() for n in agent.networks()]
        for n, p in zip(agent.networks(), P0):
            n.set_params(p)
        agent.actor.train(batch); agent.critic.train(batch)               # conv1 reads the gathered copy
        agent.target_actor.update_weights(); agent.target_critic.update_weights()
        unfused = [n.get_params() for n in agent.networks()]
        assert_flat_close(aspec, fused[0], unfused[0], rel=1e-5, what="actor")
        assert_flat_close(cspec, fused[1], unfused[1], rel=1e-5, what="critic")
        assert np.abs(fused[1] - P0[1]).max() > 0 and np.isfinite(fused[0]).all()
    finally:
        agent.close()


def test_cfg5_shard_as_the_reference_f16_store_allocates_and_trains_from_the_top():
    """one GPU's share of configs[4]'s 10^6-row replay as the reference's OWN store type (f16,
    replay_memory.py:32): 187 500 states x 983 040 B = 184 GB of the 288 GB.  The fused step reads conv1's images straight from the
    store through the sampled slots (no materialised f16 minibatch, unlike the 8-bit shard): it must allocate, sample the whole
    range, read the top of the store bit-exactly and train -- equal to the op-by-op path on a gathered copy of the same rows."""
    import ctypes
    from cartpoleplusplus_amd import _lib
    shape, rows, B, seed = (128, 128, 3, 2, 5), 125000, 512, 5
    agent, _ref, (aspec, cspec) = make_pair(shape, B, True, replay_size=rows)
    try:
        rm = agent.replay_memory
        elems = rm.state_elems
        assert rm.state_buffer_size == 187500 and rm.state_buffer_size * elems * 2 > 180e9
        rm.fill_synthetic(rows, seed=seed)
        top_slot = int(rm.state_2_idx[rows - 1])
        assert top_slot * elems * 2 > 120e9 > 2 ** 36                   # byte offset of the last filled state: 125 GB (the fill uses
                                                                        # rows + rows/50 slots of the 187 500 allocated)
        assert np.array_equal(rm.state[top_slot].reshape(-1), LEVELS[synthetic_state_codes(top_slot, elems, seed)])
        P0 = [n.get_params() for n in agent.networks()]
        agent.train_step(B, 1)
        idxs = np.empty(B, np.int32)
        _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
        assert idxs.max() > 0.9 * rows and idxs.min() < 0.1 * rows
        fused = [n.get_params() for n in agent.networks()]
        for n, p in zip(agent.networks(), P0):
            n.set_params(p)
        batch = rm.batch(idxs=idxs)
        agent.actor.train(batch); agent.critic.train(batch)
        agent.target_actor.update_weights(); agent.target_critic.update_weights()
        unfused = [n.get_params() for n in agent.networks()]
        for spec, a, b in ((aspec, fused[0], unfused[0]), (cspec, fused[1], unfused[1]), (aspec, fused[2], unfused[2]), (cspec, fused[3], unfused[3])):
            assert_flat_close(spec, a, b, rel=1e-5, what="fused step on the 184 GB f16 shard vs train ops on the gathered rows")
            assert np.isfinite(a).all()
        assert np.abs(fused[1] - P0[1]).max() > 0
    finally:
        agent.close()
