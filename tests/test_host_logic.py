"""CPU tests of host-side logic that needs no GPU: flags, exploration noise, helper functions."""
import numpy as np
import pytest

from cartpoleplusplus_amd import util


def test_flag_defaults_match_the_reference():
    from cartpoleplusplus_amd import ddpg_cartpole as D
    o = D.default_opts()
    # ddpg_cartpole.py:18-57, util.py:10-20
    assert (o.batch_size, o.batches_per_step, o.target_update_rate) == (128, 5, 0.0001)
    assert (o.actor_learning_rate, o.critic_learning_rate, o.discount) == (0.001, 0.01, 0.99)
    assert (o.replay_memory_size, o.replay_memory_burn_in) == (22000, 1000)
    assert (o.action_noise_theta, o.action_noise_sigma) == (0.01, 0.05)
    assert o.gradient_clip == 5 and o.actor_hidden_layers == "100,100,50"
    assert util.gradient_clip_value(o) == 5.0


def test_ou_noise_reproduces_the_clip_quirk():
    np.random.seed(0)
    n = util.OrnsteinUhlenbeckNoise(2, theta=0.5, sigma=10.0)
    xs = np.array([n.sample() for _ in range(300)])
    assert xs.max() <= 1.5 and xs.min() < -1.5      # util.py:155: only the upper bound is enforced
    from oracle.ddpg_np import OUNoise
    np.random.seed(7); a = util.OrnsteinUhlenbeckNoise(3, 0.01, 0.2)
    b = OUNoise(3, 0.01, 0.2, rng=np.random.RandomState(7))
    for _ in range(20):
        assert np.array_equal(a.sample(), b.sample())


def test_collapsed_successive_ranges():
    assert util.collapsed_successive_ranges([2, 3, 4, 5, 13, 14, 15]) == "2-5, 13-15"
    assert util.collapsed_successive_ranges([7]) == "7-7"


def test_synthetic_env_shapes():
    from cartpoleplusplus_amd import ddpg_cartpole as D
    from cartpoleplusplus_amd.synthetic_env import SyntheticCartpole
    o = D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3)
    env = SyntheticCartpole(o)
    s = env.reset()
    assert s.shape == (64, 64, 3, 2, 3) and s.dtype == np.float32
    assert np.array_equal(s, s.astype(np.float16).astype(np.float32))   # f16(k/255) values
    s2, r, done, _ = env.step(np.zeros((1, 2)))
    assert s2.shape == s.shape and r == 1.0


def test_bench_flop_accounting_matches_the_survey():
    """conv FLOPs per step = 2 B (4 F + 2 Bk) with F / Bk the unpadded per-image MACs -- cfg2 39.74, cfg3 68.05,
    cfg5 846.4 GFLOP (the numerators of every roofline figure bench.py prints)."""
    import bench
    want = {"cfg2": (12006400, 14796800, 39.74), "cfg3": (21222400, 24012800, 68.05), "cfg5": (134041600, 145203200, 846.4)}
    for wl, (F_, Bk_, gf) in want.items():
        shape, B, kind = bench.WORKLOADS[wl]
        F, Bk, per_layer = bench.conv_macs(shape)
        assert (F, Bk) == (F_, Bk_) and kind == "ddpg"
        assert abs(2.0 * B * (4 * F + 2 * Bk) / 1e9 - gf) < 0.05 * max(1.0, gf / 100)
    F50 = bench.conv_macs((50, 50, 3, 1, 2))[0]                          # the default render: 5.44 MMAC per image
    assert abs(F50 / 1e6 - 5.44) < 0.01
    assert bench.PIPES["f16x2"][0] == bench.PEAK_F16_MFMA_TFLOPS / 2.0 and bench.PIPES["f16x3"][0] == bench.PEAK_F16_MFMA_TFLOPS / 3.0 and bench.PIPES["bf16x6"][0] == bench.PEAK_F16_MFMA_TFLOPS / 6.0


def test_host_gradient_helpers_follow_util_py():
    """util.py:33-58 on host arrays: l2_norm, standardise, clip_and_debug_gradients (tf.clip_by_global_norm's rule, None skipped)."""
    from cartpoleplusplus_amd import util

    class Opts(object):
        gradient_clip, print_gradients = 5.0, False
    g1, g2 = np.full((3, 4), 2.0, np.float32), np.full(5, -3.0, np.float32)
    norm = np.sqrt(12 * 4.0 + 5 * 9.0)
    assert abs(util.l2_norm(g1) - np.sqrt(48.0)) < 1e-12
    out = util.clip_and_debug_gradients([(g1, "a"), (None, "b"), (g2, "c")], Opts)
    assert out[1][0] is None and [v for _g, v in out] == ["a", "b", "c"]
    np.testing.assert_allclose(out[0][0], g1 * 5.0 / norm, rtol=1e-6)
    np.testing.assert_allclose(out[2][0], g2 * 5.0 / norm, rtol=1e-6)
    Opts.gradient_clip = 100.0                                        # norm below the clip: unchanged
    np.testing.assert_allclose(util.clip_and_debug_gradients([(g1, "a")], Opts)[0][0], g1)
    Opts.gradient_clip = None
    assert util.clip_and_debug_gradients([(g1, "a")], Opts)[0][0] is g1
    z = util.standardise(np.arange(10.0))
    assert abs(z.mean()) < 1e-12 and abs((z ** 2).mean() - 1.0) < 1e-12


# ---------------------------------------------------------------------------------------------------------------------
# training_loop.py: the agents' outer loop (ddpg_cartpole.py:291-383) with stubs -- no device, no process group
# ---------------------------------------------------------------------------------------------------------------------
class _LoopAgent(object):
    def __init__(self, episode_len):
        import collections

        class Env(object):
            def __init__(self):
                self.t = 0

            def reset(self):
                self.t = 0
                return np.zeros(3, np.float32)

            def step(self, action):
                self.t += 1
                return np.full(3, self.t, np.float32), 1.0, self.t >= episode_len, {}

        class Mem(object):
            def __init__(self):
                self.rows, self.stats = 0, collections.Counter()

            def add_episode(self, s0, seq):
                self.rows += len(seq)
                self.stats[">add_episode"] += 1

            def size(self):
                return self.rows

            def current_stats(self):
                return dict(self.stats)
        self.env, self.replay_memory, self.evals, self.trained = Env(), Mem(), 0, []

    def run_eval(self, n, add_noise=False):
        self.evals += n


def _loop_opts(**kw):
    import types
    o = types.SimpleNamespace(dont_do_rollouts=False, replay_memory_burn_in=10, async_rollouts=False)
    o.__dict__.update(kw)
    return o


def test_training_loop_follows_the_reference_order_of_events():
    import io
    from cartpoleplusplus_amd.training_loop import TrainingLoop
    agent, out = _LoopAgent(4), io.StringIO()

    def train(batch_size, batches_per_step):
        agent.trained.append(agent.replay_memory.size())
        return [0.5]
    loop = TrainingLoop(agent, _loop_opts(), act=lambda s: np.zeros((1, 2), np.float32), train=train, out=out)
    loop.run(max_num_actions=45, max_run_time=0, batch_size=8, batches_per_step=5, saver_util=None)
    # 4 actions per episode: training starts once size() > 10 (3rd episode), the loop leaves once actions > 45 (12th episode)
    assert loop.iterations == 12 and agent.trained == [12 + 4 * k for k in range(10)]
    lines = [l for l in out.getvalue().splitlines() if l.startswith("STATS")]
    assert len(lines) == 12 and agent.evals == 1                                  # eval after the 10th episode (n % 10 == 0)
    import json
    first, last = json.loads(lines[0].split("\t")[1]), json.loads(lines[-1].split("\t")[1])
    assert np.isnan(first["mean_losses"]) and last["mean_losses"] == 0.5 and last["episode_len"] == 4 and last["n"] == 11


def test_training_loop_without_rollouts_runs_exactly_one_iteration_when_no_budget_is_given():
    import io
    from cartpoleplusplus_amd.training_loop import TrainingLoop
    agent = _LoopAgent(4)
    agent.replay_memory.rows = 100
    loop = TrainingLoop(agent, _loop_opts(dont_do_rollouts=True), act=None, train=lambda b, n: [1.0], out=io.StringIO())
    loop.run(0, 0, 8, 5, None)
    assert loop.iterations == 1 and loop.train_calls == 1                         # ddpg_cartpole.py: --dont-do-rollouts one-shot


def test_async_rollouts_keep_training_while_episodes_are_played_and_surface_env_errors():
    import io
    import time as _time
    from cartpoleplusplus_amd.training_loop import TrainingLoop
    agent = _LoopAgent(5)
    slow_step = agent.env.step

    def step(action):
        _time.sleep(0.002)
        return slow_step(action)
    agent.env.step = step
    loop = TrainingLoop(agent, _loop_opts(async_rollouts=True), act=lambda s: np.zeros((1, 2), np.float32),
                        train=lambda b, n: [0.0], out=io.StringIO())
    loop.run(max_num_actions=60, max_run_time=0, batch_size=8, batches_per_step=5, saver_util=None)
    assert agent.replay_memory.rows > 60
    assert loop.train_calls > agent.replay_memory.stats[">add_episode"]           # the learner did not wait for episodes
    # an exception in the environment reaches the caller of run()
    bad = _LoopAgent(5)

    def boom(action):
        raise ValueError("physics exploded")
    bad.env.step = boom
    loop = TrainingLoop(bad, _loop_opts(async_rollouts=True), act=lambda s: 0, train=lambda b, n: [0.0], out=io.StringIO())
    with pytest.raises(ValueError):
        loop.run(60, 0, 8, 5, None)


def test_fair_lock_serves_in_arrival_order():
    import threading
    import time as _time
    from cartpoleplusplus_amd.training_loop import FairLock
    lock, order = FairLock(), []

    def greedy():
        for _ in range(50):
            with lock:
                order.append("L")
                _time.sleep(0.0005)

    def occasional():
        for _ in range(5):
            _time.sleep(0.002)
            with lock:
                order.append("R")
    ts = [threading.Thread(target=greedy), threading.Thread(target=occasional)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert order.count("R") == 5 and order.index("R") < 15 and "R" in order[:40]


def test_deferred_naf_loss_behaves_like_the_float_the_reference_logs(monkeypatch):
    """naf.train(batch) on a replay draw returns a DeferredLoss: resolved once, on first use; usable wherever the reference uses the
    float (losses.append(...), np.mean(losses), json, formatting, comparisons); a check_numerics hit raises when looked at."""
    import json
    from cartpoleplusplus_amd import naf_cartpole as N

    class FakeLib(object):
        def __init__(self):
            self.waits = []

        def cpp_naf_loss_wait(self, handle, ticket, loss_ref):
            self.waits.append(int(ticket))
            loss_ref._obj.value = 0.5 * (int(ticket) + 1)
            return 4 if int(ticket) == 7 else 0

        def cpp_last_error(self):
            return b"check_numerics: loss is not finite"
    fake = FakeLib()
    monkeypatch.setattr(N, "lib", fake)
    net = type("Net", (), {"handle": "h"})()
    losses = [N.DeferredLoss(net, t) for t in range(3)]
    assert fake.waits == []                                   # nothing fetched yet
    assert float(np.mean(losses)) == 1.0 and sorted(fake.waits) == [0, 1, 2]
    assert float(losses[1]) == 1.0 and sorted(fake.waits) == [0, 1, 2]      # cached
    assert losses[0] + 1 == 1.5 and 2 * losses[2] == 3.0 and losses[0] < losses[1] and "%.2f" % losses[2] == "1.50"
    assert json.dumps({"mean_losses": float(np.mean(losses))}) == '{"mean_losses": 1.0}'
    bad = N.DeferredLoss(net, 7)
    with pytest.raises(FloatingPointError):
        float(bad)
    with pytest.raises(FloatingPointError):                   # and again: the error sticks to the minibatch
        bad + 1


def test_bench_self_launch_builds_the_drivers_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE re-runs itself as N ranks: python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <the same flags>."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, cwd=None, env=None):
        seen.update(cmd=cmd, cwd=cwd, env=env)
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert bench.self_launch(8) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_fair_lock_skips_the_ticket_of_an_interrupted_waiter():
    """a waiter that is interrupted (Ctrl-C) between taking its ticket and being served left a ticket nobody would ever
    serve: every later __enter__ -- AsyncRollouts.stop()'s included -- waited for ever."""
    import threading
    from cartpoleplusplus_amd.training_loop import FairLock
    lock = FairLock()
    lock.__enter__()                                   # ticket 0 is being served
    hit = []

    class Interrupt(BaseException):
        pass
    real_wait = lock._cv.wait

    def wait_then_interrupt(*a, **k):                  # the queued waiter is interrupted inside its wait
        hit.append(1)
        raise Interrupt()
    lock._cv.wait = wait_then_interrupt
    try:
        lock.__enter__()                               # ticket 1: queued behind ticket 0, interrupted
    except Interrupt:
        pass
    lock._cv.wait = real_wait
    assert hit
    lock.__exit__(None, None, None)                    # ticket 0 leaves: ticket 1 was abandoned and must be skipped
    done = threading.Event()

    def later():
        with lock:
            done.set()
    t = threading.Thread(target=later, daemon=True)
    t.start()
    assert done.wait(5.0), "a ticket taken by an interrupted waiter blocks the lock"


@pytest.mark.parametrize("mod", ["ddpg_cartpole", "naf_cartpole"])
def test_data_parallel_refuses_host_rng_sampling(mod):
    """--data-parallel --host-rng-sampling ran the local literal loop with no all-reduce while LoopAgreement kept the ranks'
    train / stop decisions collective: replicas that silently diverge.  The combination is an error before anything is built."""
    import importlib
    m = importlib.import_module("cartpoleplusplus_amd." + mod)
    with pytest.raises(SystemExit) as e:
        m.main(["--data-parallel", "--host-rng-sampling", "--synthetic-env"])
    assert "host-rng-sampling" in str(e.value)
