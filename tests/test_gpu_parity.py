"""GPU parity tests: the HIP path, called through the package's public surface (every call lands in the
C ABI of include/cartpolepp_abi.h), against the oracle on identical seeded inputs.

Tolerances: Q-values / actions / TD within 1e-5 absolute (north_star); gradients and updated
parameters within 2e-5 relative L2 per variable (f32 MFMA accumulation vs the f64 oracle); replay
payloads, indices and gathers bit-exact.
"""
import numpy as np
import pytest

from oracle import ddpg_np as O
from oracle.replay_np import OracleReplayMemory
from tests.helpers import make_pair, assert_flat_close, per_var_report

pytestmark = pytest.mark.gpu

ATOL = 1e-5

PIXEL_CASES = [
    pytest.param((8, 8, 3, 1, 2), 4, id="8x8x6-B4"),
    pytest.param((12, 10, 3, 1, 3), 3, id="12x10x9-B3-oddpool"),
    pytest.param((50, 50, 3, 1, 2), 3, id="50x50x6-B3-default-render"),
    pytest.param((64, 64, 3, 2, 3), 2, id="64x64x18-B2-cfg3-shape"),
    pytest.param((64, 64, 3, 1, 3), 5, id="64x64x9-B5-cfg2-shape"),
    # further geometries of the f16-pipe conv1 kernels (conv_k16.h / conv_dw16.h): strips, images per workgroup, chunks
    pytest.param((50, 50, 3, 2, 3), 3, id="50x50x18-B3-run_98-render"),
    pytest.param((16, 16, 3, 2, 3), 5, id="16x16x18-B5"),
    pytest.param((32, 32, 3, 2, 3), 3, id="32x32x18-B3"),
    pytest.param((24, 40, 3, 2, 2), 3, id="24x40x12-B3"),
]
LOWDIM_CASE = pytest.param((2, 2, 7), 5, id="lowdim-28-B5-cfg1")
LOWDIM_FULL = pytest.param((2, 2, 7), 256, id="lowdim-28-B256-cfg1-full-size")      # BASELINE configs[0] at its own batch size


def _batch(rng, B, shape, pixel):
    return O.synthetic_batch(rng, B, shape, 2, pixel)


class HostBatch(object):
    def __init__(self, t):
        self.state_1, self.action, self.reward, self.terminal_mask, self.state_2 = t


@pytest.mark.parametrize("shape,B", PIXEL_CASES + [LOWDIM_CASE])
def test_forward_actions_q_and_pools(shape, B):
    pixel = len(shape) == 5
    agent, ref, _ = make_pair(shape, B, pixel)
    rng = np.random.default_rng(11)
    s1, a, r, m, s2 = _batch(rng, B, shape, pixel)
    try:
        for dtype in (np.float16, np.float32):
            got_a = agent.actor.forward(s1.astype(dtype))
            want = ref.actor.forward(s1)
            assert np.abs(got_a - want["out"]).max() < ATOL
            got_q = agent.critic.forward(s1.astype(dtype), a)
            wq = ref.critic.forward(s1, action=a)
            assert np.abs(got_q - wq["out"]).max() < ATOL
            if pixel:
                for i, name in enumerate(("conv1", "conv2", "conv3")):
                    pool = getattr(agent.critic, "pool%d" % (i + 1)).eval(B)
                    assert np.abs(pool - wq[name][1]).max() < ATOL, name
        # batch-of-one inference whitens with that image's own statistics (base_network.py:95-99)
        one = agent.actor.action_given(s1[0].astype(np.float32))
        assert one.shape == (1, 2)
        assert np.abs(one - ref.action_given(s1[0])).max() < ATOL
    finally:
        agent.close()


@pytest.mark.parametrize("shape,B", PIXEL_CASES + [LOWDIM_CASE])
def test_gradients_dq_da_and_check_loss(shape, B):
    pixel = len(shape) == 5
    agent, ref, (aspec, cspec) = make_pair(shape, B, pixel)
    rng = np.random.default_rng(5)
    t = _batch(rng, B, shape, pixel)
    try:
        hb = HostBatch(t)
        ag = ref.actor_gradients(t[0])
        cg = ref.critic_gradients(t)
        dq = agent.critic.q_gradients_wrt_actions(hb)
        assert np.abs(dq - ag["dq_da"]).max() < ATOL
        loss, td, q = agent.critic.check_loss(hb)
        assert np.abs(q - cg["q"]).max() < ATOL and np.abs(td - cg["td"]).max() < ATOL
        assert abs(loss - cg["loss"]) < ATOL * max(1.0, abs(cg["loss"]))
        # pre-clip gradients of the train ops (lr 0 so parameters stay put)
        from cartpoleplusplus_amd import ddpg_cartpole as D
        p_a, p_c = agent.actor.get_params(), agent.critic.get_params()
        agent.actor.train(hb)
        assert_flat_close(aspec, agent.actor.get_grads(), ag["grads"], what="actor grads")
        agent.actor.set_params(p_a)
        agent.critic.train(hb)
        assert_flat_close(cspec, agent.critic.get_grads(), cg["grads"], what="critic grads")
        st = agent.trainer.last_stats()
        assert abs(st[2] - np.linalg.norm(cg["grads"])) < 1e-4 * max(1.0, np.linalg.norm(cg["grads"]))
    finally:
        agent.close()


@pytest.mark.parametrize("shape,B", PIXEL_CASES[:2] + PIXEL_CASES[3:4] + [LOWDIM_CASE, LOWDIM_FULL])
def test_train_ops_update_parameters_like_the_oracle(shape, B):
    """actor.train + critic.train + both target updates (ddpg_cartpole.py:331-337) vs the oracle."""
    pixel = len(shape) == 5
    agent, ref, (aspec, cspec) = make_pair(shape, B, pixel)
    rng = np.random.default_rng(9)
    try:
        for _ in range(2):
            t = _batch(rng, B, shape, pixel)
            hb = HostBatch(t)
            agent.actor.train(hb.state_1)
            agent.critic.train(hb)
            ref.train_minibatch(t)
        agent.target_actor.update_weights()
        agent.target_critic.update_weights()
        ref.update_targets()
        assert_flat_close(aspec, agent.actor.get_params(), ref.actor.flat(), rel=1e-5, what="actor params")
        assert_flat_close(cspec, agent.critic.get_params(), ref.critic.flat(), rel=1e-5, what="critic params")
        assert_flat_close(aspec, agent.target_actor.get_params(), ref.target_actor.flat(), rel=1e-6, what="target actor")
        assert_flat_close(cspec, agent.target_critic.get_params(), ref.target_critic.flat(), rel=1e-6, what="target critic")
    finally:
        agent.close()


def test_gradient_clip_engages():
    """scale the critic so its gradient norm exceeds 5: update must have norm lr*5 (util.py:47-50)."""
    shape, B = (8, 8, 3, 1, 2), 4
    agent, ref, (aspec, cspec) = make_pair(shape, B, True)
    rng = np.random.default_rng(2)
    s1, a, r, m, s2 = _batch(rng, B, shape, True)
    try:
        r = r * 1000.0
        t = (s1, a, r, m, s2)
        before = agent.critic.get_params()
        agent.critic.train(HostBatch(t))
        cg = ref.critic_gradients(t)
        norm = np.linalg.norm(cg["grads"])
        assert norm > 5.0
        delta = agent.critic.get_params() - before
        want = -0.01 * cg["grads"] * (5.0 / norm)
        assert abs(np.linalg.norm(delta) - 0.05) < 1e-5
        assert np.linalg.norm(delta - want) / np.linalg.norm(want) < 2e-5
    finally:
        agent.close()


def _fill_pair(agent, oracle_rm, rng, shape, episodes, pixel=True):
    for _ in range(episodes):
        n = int(rng.integers(2, 7))
        mk = (lambda: rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)) if pixel else \
             (lambda: rng.standard_normal(shape).astype(np.float32))
        s0 = mk()
        seq = [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.integers(0, 5)), mk()) for _ in range(n)]
        agent.replay_memory.add_episode(s0, seq)
        oracle_rm.add_episode(s0, seq)


@pytest.mark.parametrize("shape,B", [PIXEL_CASES[0], PIXEL_CASES[3], LOWDIM_CASE, LOWDIM_FULL])
def test_fused_train_step_matches_oracle_and_unfused_ops(shape, B):
    """cpp_ddpg_train_step with caller rows: replay gather + both updates x n_batches + target updates."""
    pixel = len(shape) == 5
    agent, ref, (aspec, cspec) = make_pair(shape, B, pixel, replay_size=24)
    rng = np.random.default_rng(21)
    orm = OracleReplayMemory(24, shape, 2)
    try:
        _fill_pair(agent, orm, rng, shape, 9, pixel)          # wraps the 24-row buffer
        assert agent.replay_memory.size() == orm.size() == 24
        nb = 3
        idxs = rng.integers(0, 24, nb * B)
        batches = []
        for i in range(nb):
            ob = orm.batch(idxs=idxs[i * B:(i + 1) * B])
            batches.append((ob.state_1, ob.action, ob.reward, ob.terminal_mask, ob.state_2))
        outs = ref.train_step(batches)
        agent.train_step(B, nb, idxs=idxs)
        assert_flat_close(aspec, agent.actor.get_params(), ref.actor.flat(), rel=1e-5, what="actor params")
        assert_flat_close(cspec, agent.critic.get_params(), ref.critic.flat(), rel=1e-5, what="critic params")
        assert_flat_close(aspec, agent.target_actor.get_params(), ref.target_actor.flat(), rel=1e-6, what="target actor")
        assert_flat_close(cspec, agent.target_critic.get_params(), ref.target_critic.flat(), rel=1e-6, what="target critic")
        st = agent.trainer.last_stats()
        assert abs(st[0] - outs[-1]["loss"]) < ATOL * max(1.0, abs(outs[-1]["loss"]))
    finally:
        agent.close()


def test_graph_replay_is_deterministic_and_equals_eager():
    """idxs=None: device Philox rows; the first call runs eagerly and captures a hipGraph, later calls
    replay it.  Two agents with the same seeds must agree bit for bit, and the profiled (eager) path
    must agree with the graph path."""
    shape, B = (16, 16, 3, 1, 2), 8
    params = []
    for mode in ("graph", "graph", "eager"):
        agent, _ref, _ = make_pair(shape, B, True, replay_size=200)
        try:
            agent.replay_memory.fill_synthetic(150, seed=77)
            if mode == "eager":
                agent.actor.ctx.prof_enable(True)
            for _ in range(4):
                agent.train_step(B, 2)
            agent.actor.ctx.sync()
            agent.actor.ctx.prof_enable(False)
            params.append((agent.actor.get_params(), agent.critic.get_params(), agent.target_critic.get_params()))
        finally:
            agent.close()
    for k in range(3):
        assert np.array_equal(params[0][k], params[1][k])
        assert np.array_equal(params[0][k], params[2][k])
    assert np.isfinite(params[0][0]).all() and np.isfinite(params[0][1]).all()


def test_data_parallel_half_steps_equal_the_fused_step():
    """the data-parallel learner's split step (cpp_ddpg_sample_and_compute -> [all-reduce] -> cpp_ddpg_apply_gradients, then
    the target updates) at world size 1 must walk the same minibatches to the same parameters as the fused train step."""
    from cartpoleplusplus_amd.distributed import DataParallelLearner, GradAllReducer, AgentOps
    import torch
    shape, B = (16, 16, 3, 1, 2), 16
    params = []
    for mode in ("fused", "dp"):
        agent, _ref, _ = make_pair(shape, B, True, replay_size=300)
        try:
            agent.replay_memory.fill_synthetic(200, seed=11)
            if mode == "fused":
                for _ in range(5):
                    agent.train_step(B, 3)
            else:
                from cartpoleplusplus_amd import ddpg_cartpole as D
                learner = DataParallelLearner(AgentOps(agent, B, int(D.opts.sample_seed)), GradAllReducer(torch.zeros(1)))
                for _ in range(5):
                    learner.train_step(3)
            agent.actor.ctx.sync()
            params.append((agent.actor.get_params(), agent.critic.get_params(), agent.target_actor.get_params()))
        finally:
            agent.close()
    for k in range(3):     # (the fused step's dW reductions may sum in 4 slices instead of 16: last-bit differences only)
        assert np.abs(params[0][k] - params[1][k]).max() < 1e-6, float(np.abs(params[0][k] - params[1][k]).max())


def test_replayed_graph_samples_the_grown_replay_memory():
    """the sampler's range (rows currently in the memory) is an argument of the captured sample kernel: when episodes have
    been added since the capture, the next train step must draw from the whole memory, not replay the old range."""
    import ctypes
    from cartpoleplusplus_amd import _lib
    shape, B = (16, 16, 3, 1, 2), 64
    agent, _ref, _ = make_pair(shape, B, True, replay_size=700)
    try:
        rm = agent.replay_memory
        rm.fill_synthetic(100, seed=3)
        for _ in range(3):                                 # eager + capture + replay
            agent.train_step(B, 2)
        idxs = np.empty(B, np.int32)
        _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
        assert idxs.min() >= 0 and idxs.max() < 100
        rm.fill_synthetic(600, seed=3)
        seen = []
        for _ in range(3):
            agent.train_step(B, 2)
            _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
            seen.append(idxs.copy())
        seen = np.concatenate(seen)
        assert seen.min() >= 0 and seen.max() < 600
        assert (seen >= 100).sum() > len(seen) // 2, seen    # ~5/6 of uniform draws over 600 rows
    finally:
        agent.close()


def test_training_from_the_8_bit_replay_store_is_bit_identical():
    """--replay-store u8: the gathered minibatches are the same f16 bits, so whole training steps (device sampling,
    f16-pipe conv1 kernels, fused heads, updates) must leave exactly the same parameters as the f16 store."""
    shape, B = (16, 16, 3, 2, 3), 8
    params = []
    for store in ("f16", "u8"):
        agent, _ref, _ = make_pair(shape, B, True, replay_size=200, replay_store=store)
        try:
            assert agent.replay_memory.store_dtype == store
            agent.replay_memory.fill_synthetic(150, seed=31)
            for _ in range(3):
                agent.train_step(B, 2)
            params.append((agent.actor.get_params(), agent.critic.get_params(), agent.target_actor.get_params()))
        finally:
            agent.close()
    for k in range(3):
        assert np.array_equal(params[0][k], params[1][k])
    assert np.isfinite(params[0][0]).all()


def test_end_to_end_cli_with_checkpoints_and_event_log(tmp_path, capsys):
    """the agent's own main(): rollouts on the stand-in env, replay in HBM, fused train steps, STATS lines, a
    checkpoint that restores bit for bit, then offline training from the event log it wrote (--event-log-in
    --dont-do-rollouts, as exps/run_81-84 do)."""
    import json
    from cartpoleplusplus_amd import ddpg_cartpole as D, event_log as E
    from cartpoleplusplus_amd.synthetic_env import SyntheticCartpole
    ck = str(tmp_path / "ckpts")
    common = ["--synthetic-env", "--use-raw-pixels", "--render-width", "16", "--render-height", "16", "--batch-size", "8",
              "--replay-memory-size", "120", "--replay-memory-burn-in", "20", "--max-episode-len", "12", "--ckpt-dir", ck]
    D.main(common + ["--max-num-actions", "60"])
    out = capsys.readouterr().out
    stats = [json.loads(l.split("\t", 1)[1]) for l in out.splitlines() if l.startswith("STATS")]
    assert len(stats) >= 4 and stats[-1]["replay_memory_stats"][">add"] >= 60
    assert any(np.isfinite(s["mean_losses"]) for s in stats)          # training ran once the burn-in was passed
    # restore: a fresh agent built from the checkpoint has the saved parameters
    D.set_opts(D.build_parser().parse_args(common))
    agent = D.DeepDeterministicPolicyGradientAgent(SyntheticCartpole(D.opts))
    from cartpoleplusplus_amd import util
    saver = util.SaverUtil(agent, ck, 3600)
    latest = [l for l in open(ck + "/checkpoint")][0].split('"')[1]
    blob = np.load("%s/%s.npz" % (ck, latest))
    for net in agent.networks():
        assert np.array_equal(net.get_params(), blob[net.namespace])
    # an event log written by EventLog primes the replay memory for rollout-free training
    path = str(tmp_path / "events")
    log = E.EventLog(path, use_raw_pixels=True)
    env = SyntheticCartpole(D.opts, seed=3)
    for _ in range(4):
        log.reset()
        log.add_just_state(env.reset())
        done = False
        while not done:
            a = np.random.uniform(-1, 1, (1, 2)).astype(np.float32)
            s2, r, done, _ = env.step(a)
            log.add(s2, a, r)
    log.close()
    agent.close()
    D.main(common[:-2] + ["--event-log-in", path, "--dont-do-rollouts"])
    out = capsys.readouterr().out
    last = json.loads([l for l in out.splitlines() if l.startswith("STATS")][-1].split("\t", 1)[1])
    assert last["episode_len"] == 0 and last["replay_memory_stats"][">add_episode"] == 4 and np.isfinite(last["mean_losses"])


@pytest.mark.parametrize("shape,B", [PIXEL_CASES[0], PIXEL_CASES[2], PIXEL_CASES[3], LOWDIM_CASE])
def test_actions_given_equals_one_action_given_per_row(shape, B):
    """cpp_net_forward_each: a batch of independent states, each whitened with its own statistics, gives bit for bit
    the actions of B separate batch-of-one forwards (ddpg_cartpole.py:121-126), and matches the oracle per row."""
    pixel = len(shape) == 5
    agent, ref, _ = make_pair(shape, B, pixel)
    rng = np.random.default_rng(11)
    t = O.synthetic_batch(rng, B, shape, 2, pixel)
    try:
        got = agent.actor.actions_given(t[0])
        one = np.concatenate([agent.actor.action_given(t[0][i]) for i in range(B)], axis=0)
        assert got.shape == (B, 2) and np.array_equal(got, one)
        want = np.concatenate([ref.actor.forward(t[0][i:i + 1])["out"] for i in range(B)], axis=0)
        assert np.abs(got - want).max() < 1e-5
        if pixel:       # and it is NOT the batch-statistics forward
            assert np.abs(got - agent.actor.forward(t[0])).max() > 0
    finally:
        agent.close()
