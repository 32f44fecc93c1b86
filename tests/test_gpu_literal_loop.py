"""The reference's inner loop, written as the reference writes it (ddpg_cartpole.py:331-337, naf_cartpole.py:365-373), must be the
fused device sequence: `batch.state_1` is a device-resident column, `actor.train(batch.state_1)` waits for `critic.train(batch)`
and both run as cpp_ddpg_train_rows on the draw's rows -- bit-identical to `agent.train_step(..., idxs=)` on the same rows, with
no state bytes over PCIe.  Anything that looks at the actor in between gets the unfused update first."""
import collections

import numpy as np
import pytest

from tests.helpers import make_pair

pytestmark = pytest.mark.gpu


def _twin_agents(shape, B, pixel, rows, seed=3, **kw):
    out = []
    for _ in range(2):
        agent, _ref, _ = make_pair(shape, B, pixel, seed=seed, replay_size=rows + 40, **kw)
        agent.replay_memory.fill_synthetic(rows, seed=21)
        out.append(agent)
    return out


def _maxrel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("shape,B,pixel", [((32, 32, 3, 2, 3), 32, True), ((16, 16, 3, 1, 2), 8, True), ((2, 2, 7), 16, False),
                                           ((64, 64, 3, 2, 3), 256, True)], ids=["32x32x18-B32", "16x16x6-B8", "lowdim-B16", "cfg3-B256"])
def test_reference_loop_verbatim_is_the_fused_step(shape, B, pixel):
    """ddpg_cartpole.py:331-337 verbatim for 10 minibatches: every actor.train / critic.train pair runs as ONE fused device sequence
    (cpp_ddpg_train_rows) and no state column crosses PCIe.  With --batches-per-step 1 the loop is bit-identical to
    agent.train_step(B, 1, idxs) call by call (the same kernels on the same rows, graph replay included); with 5 minibatches per step
    it equals train_step(B, 5, idxs) to f32 rounding (that call sends minibatch i + 1's sample pass along with minibatch i's
    backward kernels, which moves last bits: profiles/debug_literal_bits.py)."""
    for batches_per_step, steps in ((1, 10), (5, 2)):
        lit, fused = _twin_agents(shape, B, pixel, rows=200)
        try:
            np.random.seed(1234)
            drawn = []
            for _step in range(steps):
                # ---- ddpg_cartpole.py:331-337, verbatim (self -> lit)
                for _ in range(batches_per_step):
                    batch = lit.replay_memory.batch(B)
                    lit.actor.train(batch.state_1)
                    lit.critic.train(batch)
                    drawn.append(batch)
                lit.target_actor.update_weights()
                lit.target_critic.update_weights()
            assert lit.trainer.fused_pairs == 10
            assert all(b._states is None for b in drawn), "a state column crossed PCIe"
            idxs = np.concatenate([b.idxs for b in drawn])
            n = batches_per_step * B
            for k in range(steps):
                fused.train_step(B, batches_per_step, idxs=idxs[k * n:(k + 1) * n])
            for a, b in zip(lit.networks(), fused.networks()):
                pa, pb = a.get_params(), b.get_params()
                if batches_per_step == 1:
                    assert np.array_equal(pa, pb), a.namespace
                else:
                    assert _maxrel(pa, pb) < 2e-6, (a.namespace, _maxrel(pa, pb))
            if batches_per_step == 1:
                assert np.array_equal(lit.trainer.last_stats(), fused.trainer.last_stats())
            # the columns are still readable afterwards (one download, from the replay store) and are the draw's pixels
            s1 = np.asarray(drawn[-1].state_1)
            assert s1.shape == (B,) + tuple(shape) and s1.dtype == np.float16
            assert np.array_equal(s1, lit.replay_memory.state[drawn[-1].state_1_idx])
        finally:
            lit.close(); fused.close()


def test_a_deferred_actor_update_lands_before_anything_observes_the_actor():
    shape, B = (16, 16, 3, 1, 2), 8
    a1, a2 = _twin_agents(shape, B, True, rows=100)
    try:
        idx = np.arange(3, 3 + B)
        # reference order with host arrays: the unfused ops
        hb = a2.replay_memory.batch(idxs=idx)
        s1_host = np.asarray(hb.state_1)
        a2.actor.train(s1_host)
        want_actor = a2.actor.get_params()
        # deferred, then observed
        b = a1.replay_memory.batch(idxs=idx)
        before = a1.actor._handle
        a1.actor.train(b.state_1)
        assert a1.trainer._pending is not None and a1.actor._handle == before
        got = a1.actor.get_params()                         # flushes
        assert a1.trainer._pending is None and a1.trainer.fused_pairs == 0
        assert np.array_equal(got, want_actor)
        # a critic.train on ANOTHER draw does not fuse with the pending actor update
        b1, b2 = a1.replay_memory.batch(idxs=idx), a1.replay_memory.batch(idxs=idx + 20)
        h1, h2 = a2.replay_memory.batch(idxs=idx), a2.replay_memory.batch(idxs=idx + 20)
        a1.actor.train(b1.state_1); a1.critic.train(b2)
        a2.actor.train(np.asarray(h1.state_1)); a2.critic.train(h2)
        assert a1.trainer.fused_pairs == 0
        for x, y in zip(a1.networks(), a2.networks()):
            assert np.array_equal(x.get_params(), y.get_params()), x.namespace
        # action_given (the rollout) between the two calls sees the updated actor
        b3, h3 = a1.replay_memory.batch(idxs=idx + 40), a2.replay_memory.batch(idxs=idx + 40)
        a1.actor.train(b3.state_1)
        a2.actor.train(np.asarray(h3.state_1))
        st = np.asarray(h3.state_1)[0]
        assert np.array_equal(a1.actor.action_given(st), a2.actor.action_given(st))
    finally:
        a1.close(); a2.close()


def test_an_older_batch_trains_on_its_own_rows_after_a_second_draw():
    """b1 read, then b2 = batch() of the same size -- training on b1 must use b1's rows, not the shared device buffer's."""
    shape, B = (16, 16, 3, 1, 2), 8
    a1, a2 = _twin_agents(shape, B, True, rows=100)
    try:
        HostBatch = collections.namedtuple("HostBatch", "state_1 action reward terminal_mask state_2")      # replay_memory.py:9
        i1, i2 = np.arange(B), np.arange(50, 50 + B)
        b1 = a1.replay_memory.batch(idxs=i1)
        _ = b1.reward, np.asarray(b1.state_1)               # read
        b2 = a1.replay_memory.batch(idxs=i2)
        assert b2.device is not None                        # the shared buffer now holds b2's draw
        a1.actor.train(b1); a1.critic.train(b1)             # whole-Batch form: the unfused ops on b1.device
        h1 = a2.replay_memory.batch(idxs=i1)
        a2.actor.train(h1); a2.critic.train(h1)
        for x, y in zip(a1.networks(), a2.networks()):
            assert np.array_equal(x.get_params(), y.get_params()), x.namespace
        # ... and after a write to the memory b1 trains from its cached host columns, an unread Batch from its preserved device copy
        b3 = a1.replay_memory.batch(idxs=i2)
        h3 = a2.replay_memory.batch(idxs=i2)
        h3_cols = HostBatch(np.asarray(h3.state_1), h3.action, h3.reward, h3.terminal_mask, np.asarray(h3.state_2))
        frame = np.asarray(b1.state_1)[0]
        a1.replay_memory.add_episode(frame, [(np.zeros((1, 2), np.float32), 1.0, frame)])
        a2.replay_memory.add_episode(frame, [(np.zeros((1, 2), np.float32), 1.0, frame)])
        a1.critic.train(b1)
        a2.critic.train(HostBatch(np.asarray(b1.state_1), b1.action, b1.reward, b1.terminal_mask, np.asarray(b1.state_2)))
        assert np.array_equal(a1.critic.get_params(), a2.critic.get_params())
        assert not b3.in_replay()          # (preserved: in the device buffer of its size, or on the host if another unread Batch took that)
        a1.critic.train(b3)
        a2.critic.train(h3_cols)
        assert np.array_equal(a1.critic.get_params(), a2.critic.get_params())
    finally:
        a1.close(); a2.close()


def test_naf_reference_loop_verbatim_trains_on_the_device_rows():
    from cartpoleplusplus_amd import naf_cartpole as N
    from tests.helpers import FakeEnv, make_opts
    shape, B = (32, 32, 3, 2, 3), 32
    agents = []
    for _ in range(4):
        make_opts(N, shape, B, True, replay_memory_size=240, share_input_state_representation=True)
        ag = N.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
        ag.initialise_variables(seed=5)
        ag.post_var_init_setup()
        ag.replay_memory.fill_synthetic(200, seed=4)
        agents.append(ag)
    try:
        for k, (batches_per_step, steps) in enumerate(((1, 4), (5, 1))):
            lit, fused = agents[2 * k], agents[2 * k + 1]
            np.random.seed(7)
            losses, drawn = [], []
            for _step in range(steps):
                # ---- naf_cartpole.py:365-373, verbatim
                for _ in range(batches_per_step):
                    batch = lit.replay_memory.batch(B)
                    losses.append(lit.naf.train(batch))
                    drawn.append(batch)
                lit.target_value_net.update_weights()
            assert all(b._states is None for b in drawn)
            idxs = np.concatenate([b.idxs for b in drawn])
            n = batches_per_step * B
            for j in range(steps):
                fused.train_step(B, batches_per_step, idxs=idxs[j * n:(j + 1) * n])
            # one minibatch per step: the same kernels on the same rows, bit for bit; five: train_step sends minibatch i + 1's sample
            # pass along with minibatch i's backward kernels, which moves last bits (as in the DDPG loop above)
            for a, b in zip(lit.networks(), fused.networks()):
                pa, pb = a.get_params(), b.get_params()
                if batches_per_step == 1:
                    assert np.array_equal(pa, pb), a.namespace
                else:
                    assert _maxrel(pa, pb) < 2e-6, (a.namespace, _maxrel(pa, pb))
            assert np.isfinite(losses).all()
            last = float(fused.naf.last_stats()[0])
            assert abs(losses[-1] - last) <= (0.0 if batches_per_step == 1 else 1e-5 * abs(last))
    finally:
        for ag in agents:
            ag.close()


def test_finished_batches_of_the_literal_loop_cost_the_next_add_episode_nothing():
    """a Batch that cached its StateColumns was a reference cycle -- after the loop body's `batch` name was rebound the old
    draw stayed alive until a gc pass, add_episode() found it in ReplayMemory._drawn and PRESERVED it (a device gather each, and a
    2 x state-column download for all but the last).  With the cycle gone, reference counting frees a finished draw at once: the
    reference's loop followed by add_episode() -- no gc pass in between -- makes no gather and no download."""
    import gc
    from cartpoleplusplus_amd import _lib
    shape, B = (32, 32, 3, 2, 3), 32
    lit, other = _twin_agents(shape, B, True, rows=200)
    other.close()
    calls = collections.Counter()
    real = {name: getattr(_lib.lib, name) for name in ("cpp_batch_download", "cpp_replay_sample")}

    def counting(name):
        def f(*args):
            calls[name] += 1
            return real[name](*args)
        return f
    gc.collect()
    gc.disable()
    try:
        for _ in range(5):                                    # ddpg_cartpole.py:331-334
            batch = lit.replay_memory.batch(B)
            lit.actor.train(batch.state_1)
            lit.critic.train(batch)
        del batch
        assert len(lit.replay_memory._drawn) == 0, "finished draws are still alive without a gc pass"
        for name in real:
            setattr(_lib.lib, name, counting(name))
        rng = np.random.default_rng(0)
        frames = [(rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)) for _ in range(4)]
        lit.replay_memory.add_episode(frames[0], [(np.zeros((1, 2), np.float32), 1.0, f) for f in frames[1:]])
        assert not calls, dict(calls)
        # ... while a draw somebody still holds IS preserved (one gather, no download) and stays readable
        held = lit.replay_memory.batch(B)
        want = lit.replay_memory.state[held.state_1_idx]
        lit.replay_memory.add_episode(frames[0], [(np.zeros((1, 2), np.float32), 1.0, f) for f in frames[1:]])
        assert calls["cpp_replay_sample"] == 1 and calls["cpp_batch_download"] == 0, dict(calls)
        assert np.array_equal(np.asarray(held.state_1), want)
    finally:
        gc.enable()
        for name, fn in real.items():
            setattr(_lib.lib, name, fn)
        lit.close()
