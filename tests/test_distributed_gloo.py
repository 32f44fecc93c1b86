"""world_size-2 gloo tests of the data-parallel learners (run on CPU).

Two layers are covered:
  * the product code above the C ABI -- communicator id hand-over, `make_learner` / `learner_for_agent` / `setup_data_parallel`,
    replica sync, `NativeLearner`, `LoopAgreement` and the agents' own `run_training` control flow (training_loop.TrainingLoop) with
    unequal episode lengths, unequal budgets and a rank that reaches burn-in late -- with tests/fake_abi.py standing in for the
    library (the only double);
  * the host-written protocol `DataParallelLearner` (the diagnostic learner's base): every rank computes its own gradients, ONE flat
    buffer is sum-all-reduced, every rank applies clip+SGD to the mean and the replicas stay bit-identical."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cartpoleplusplus_amd.distributed import DataParallelLearner, GradAllReducer


class FakeOps(object):
    """stands in for AgentOps: a quadratic model whose per-rank gradient is known in closed form."""

    def __init__(self, rank, grad_tensor, halves=False):
        self.rank, self.g, self.halves = rank, grad_tensor, halves
        self.params = np.linspace(-1, 1, grad_tensor.numel()).astype(np.float32) * np.float32(1.0 + 0.0 * rank)
        self.params_t = torch.from_numpy(self.params)               # shares memory: the reducer averages it in place
        self.target = self.params.copy()
        self.calls = []

    def sample_and_compute(self):
        self.calls.append("compute")
        data = np.full_like(self.params, float(self.rank + 1))       # the rank's "replay shard"
        if self.halves:                                               # (directions that differ between ranks after the clip)
            data[len(data) // 2:] = 1.0
        self.g.copy_(torch.from_numpy(self.params * data))

    def apply(self, grad_scale):
        self.calls.append("apply")
        g = self.g.numpy() * np.float32(grad_scale)
        norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        g = g * np.float32(5.0 / max(norm, 5.0))                      # util.py:47-50
        self.params -= np.float32(0.01) * g                           # in place: params_t sees it

    def update_targets(self):
        self.calls.append("targets")
        self.target = self.target - np.float32(1e-4) * (self.target - self.params)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.zeros(1000, dtype=torch.float32)
    ops = FakeOps(rank, g)
    learner = DataParallelLearner(ops, GradAllReducer(g))
    for _ in range(3):
        learner.train_step(5)
    out[rank] = (ops.params.copy(), ops.target.copy(), list(ops.calls))
    dist.destroy_process_group()


def test_two_learners_stay_identical_and_average_gradients():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (p0, t0, c0), (p1, t1, c1) = out[0], out[1]
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1)
    assert c0 == (["compute", "apply"] * 5 + ["targets"]) * 3
    # single-process reference: mean gradient of the two shards = params * 1.5
    p = np.linspace(-1, 1, 1000).astype(np.float32)
    for _ in range(15):
        g = (p * np.float32(1.0) + p * np.float32(2.0)) * np.float32(0.5)
        norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        g = g * np.float32(5.0 / max(norm, 5.0))
        p = p - np.float32(0.01) * g
    assert np.allclose(p0, p, rtol=1e-6, atol=1e-7)


def _periodic_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.zeros(1000, dtype=torch.float32)
    ops = FakeOps(rank, g, halves=True)
    learner = DataParallelLearner(ops, GradAllReducer(g, param_tensors=[ops.params_t]), sync_every=3)
    snaps = []
    for _ in range(4):
        learner.train_step(5)                # 20 minibatches: averagings after the 3rd, 6th, ..., 18th
        snaps.append(ops.params.copy())
    out[rank] = (snaps, list(ops.calls))
    dist.destroy_process_group()


def test_periodic_mode_takes_local_steps_and_meets_at_the_parameter_mean():
    """sync_every = 3 (SURVEY 8e "periodic"): no gradient all-reduce; each learner applies its own gradients, and after every
    third minibatch -- counted across train steps -- the parameters are averaged.  Same sequence as cpp_ddpg_dp_train_step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_periodic_worker, args=(2, port, out), nprocs=2, join=True)
    (s0, c0), (s1, c1) = out[0], out[1]
    assert c0 == (["compute", "apply"] * 5 + ["targets"]) * 4
    # single-process reference of both replicas
    P = [np.linspace(-1, 1, 1000).astype(np.float32) for _ in range(2)]
    want, local = [], 0
    for mb in range(20):
        for r in range(2):
            data = np.full(1000, np.float32(r + 1)); data[500:] = 1.0
            g = P[r] * data
            norm = np.sqrt((g.astype(np.float64) ** 2).sum())
            P[r] = P[r] - np.float32(0.01) * (g * np.float32(5.0 / max(norm, 5.0)))
        local += 1
        if local >= 3:
            m = (P[0] + P[1]) * np.float32(0.5)
            P, local = [m.copy(), m.copy()], 0
        if mb % 5 == 4:
            want.append([P[0].copy(), P[1].copy()])
    for k in range(4):
        assert np.allclose(s0[k], want[k][0], rtol=1e-6, atol=1e-7) and np.allclose(s1[k], want[k][1], rtol=1e-6, atol=1e-7)
    assert not np.array_equal(s0[0], s1[0])          # after 5 minibatches the replicas are 2 local steps apart
    assert np.array_equal(s0[2], s1[2])              # after 15 (a multiple of 3) they have just met


class _FakeNet(object):
    def __init__(self, rng, n):
        self.p = rng.standard_normal(n).astype(np.float32)

    def get_params(self):
        return self.p.copy()

    def set_params(self, p):
        assert p.shape == self.p.shape
        self.p = np.asarray(p, np.float32).copy()


class _FakeNaf(object):
    def __init__(self, rng):
        self.state = {"m": rng.standard_normal(7).astype(np.float32), "v": rng.random(7).astype(np.float32), "step": np.uint64(rng.integers(1, 99))}

    def get_optimiser_state(self):
        return dict(self.state)

    def set_optimiser_state(self, s):
        self.state = dict(s)


class _FakeAgent(object):
    def __init__(self, seed, with_opt):
        rng = np.random.default_rng(seed)
        self.nets = [_FakeNet(rng, n) for n in (11, 5, 11, 5)]
        if with_opt:
            self.naf = _FakeNaf(rng)

    def networks(self):
        return self.nets


def _sync_worker(rank, world, port, out):
    from cartpoleplusplus_amd.distributed import sync_replicas_from_rank0
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = []
    for with_opt in (False, True):
        agent = _FakeAgent(100 + rank, with_opt)              # every process draws its own weights, as the CLI's agents do
        sync_replicas_from_rank0(agent, dist)
        res.append(([n.get_params() for n in agent.nets], agent.naf.state if with_opt else None))
    out[rank] = res
    dist.destroy_process_group()


def test_replicas_start_from_rank_zeros_parameters_and_optimiser_slots():
    """the CLI's --data-parallel agents draw their own initial weights (and only rank 0 restores the checkpoint): before the first
    collective step every rank takes rank 0's networks and, for NAF, the optimiser slots and step count."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sync_worker, args=(2, port, out), nprocs=2, join=True)
    want = _FakeAgent(100, True)
    for k, with_opt in enumerate((False, True)):
        ref = _FakeAgent(100, with_opt)
        for r in range(2):
            params, st = out[r][k]
            assert all(np.array_equal(a, n.p) for a, n in zip(params, ref.nets))
            if with_opt:
                assert np.array_equal(st["m"], ref.naf.state["m"]) and np.array_equal(st["v"], ref.naf.state["v"])
                assert int(st["step"]) == int(ref.naf.state["step"])
    assert not np.array_equal(_FakeAgent(101, True).nets[0].p, want.nets[0].p)


# ---------------------------------------------------------------------------------------------------------------------
# the product code above the ABI, at world size 2 (tests/fake_abi.py is the library)
# ---------------------------------------------------------------------------------------------------------------------
class _Space(object):
    def __init__(self, shape):
        self.shape = tuple(shape)


class _ToyEnv(object):
    """episodes of a fixed, per-rank length."""

    def __init__(self, episode_len):
        self.episode_len, self.t, self.resets = episode_len, 0, 0
        self.observation_space, self.action_space = _Space((2, 3)), _Space((1, 2))

    def reset(self):
        self.t, self.resets = 0, self.resets + 1
        return np.zeros((2, 3), np.float32)

    def step(self, action):
        self.t += 1
        return np.full((2, 3), self.t, np.float32), 1.0, self.t >= self.episode_len, {}


class _ToyNet(object):
    def __init__(self, rng, n):
        self.p = rng.standard_normal(n).astype(np.float32)

    def get_params(self):
        return self.p.copy()

    def set_params(self, p):
        self.p = np.asarray(p, np.float32).copy()


def _toy_ddpg_agent(rank, episode_len, opts, halves=False):
    """a DeepDeterministicPolicyGradientAgent with every device-backed member replaced: the class's own run_training /
    train_step / _dp_learner / _train_once / _action run unchanged."""
    from cartpoleplusplus_amd import ddpg_cartpole as D
    from oracle.replay_np import OracleReplayMemory
    from tests.fake_abi import ToyTrainer
    D.set_opts(opts)
    agent = object.__new__(D.DeepDeterministicPolicyGradientAgent)
    agent.env = _ToyEnv(episode_len)
    agent.replay_memory = OracleReplayMemory(400, (2, 3), 2)
    agent.replay_memory.handle = "replay"
    trainer = ToyTrainer(rank, halves=halves)
    rng = np.random.default_rng(100 + rank)                    # every process draws its own weights, as the CLI's agents do
    nets = [_ToyNet(rng, n) for n in (11, 5, 11, 5)]

    class _Actor(object):
        def action_given(self, state, add_noise=False):
            return np.zeros((1, 2), np.float32)

    class _Critic(object):
        def _trainer(self):
            return trainer
    agent.actor, agent.critic = _Actor(), _Critic()
    agent.networks = lambda: nets
    agent.run_eval = lambda n, add_noise=False: None
    agent.train_steps = 0
    return agent, trainer, nets


def _loop_worker(rank, world, port, out, mode):
    import io
    from cartpoleplusplus_amd import ddpg_cartpole as D, distributed
    from tests import fake_abi
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = fake_abi.install(distributed)
    opts = D.default_opts(data_parallel=True, replay_memory_burn_in=20, batch_size=8, sample_seed=5,
                          sync_every=3 if mode == "periodic" else 1, async_rollouts=(mode == "async"))
    # rank 0: episodes of 10 actions, budget 60; rank 1: episodes of 4 actions (past burn-in THREE iterations after rank 0), budget 45
    agent, trainer, nets = _toy_ddpg_agent(rank, (10, 4)[rank], opts, halves=(mode == "periodic"))
    learner = distributed.setup_data_parallel(agent, opts, opts.batch_size)          # what main() does, eagerly
    assert isinstance(learner, distributed.NativeLearner) and learner.world == world and learner.seed == 5 + rank
    assert distributed.setup_data_parallel(agent, opts, opts.batch_size) is learner   # idempotent
    synced = [n.get_params() for n in nets]
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        loop = agent.run_training((60, 45)[rank], 0, opts.batch_size, 5, None)
    comm = fake.comms[learner.comm.handle.value]
    out[rank] = dict(params=trainer.params.copy(), target=trainer.target.copy(), calls=list(trainer.calls), synced=synced,
                     train_calls=loop.train_calls, iterations=loop.iterations, agree_calls=loop.agreement.calls,
                     n_allreduce=comm.n_allreduce, n_max=comm.n_max, uid=comm.uid, resets=agent.env.resets,
                     added=agent.replay_memory.stats[">add"], stats_lines=buf.getvalue().count("STATS "))
    agent._learner.close()
    assert comm.destroyed
    distributed.shutdown_data_parallel()
    assert not dist.is_initialized()


def _run_world2(worker, *args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(worker, args=(2, port, out) + args, nprocs=2, join=True)
    return out[0], out[1]


def test_agents_loop_with_unequal_learners_issues_equal_collectives_and_exits_on_both_ranks():
    """ddpg_cartpole.py:329 (burn-in) and :379-383 (exit) are per-process facts in front of a collective step.  Rank 1 plays shorter
    episodes (burn-in three iterations later) and has a smaller action budget: both ranks must train in exactly the same
    iterations, leave in the same iteration, and end with identical replicas."""
    r0, r1 = _run_world2(_loop_worker, "sync")
    assert r0["uid"] == r1["uid"] and len(r0["uid"]) == 128                        # one communicator id reached both ranks
    for a, b in zip(r0["synced"], r1["synced"]):
        assert np.array_equal(a, b)                                                # every rank started from rank 0's networks
    assert r0["iterations"] == r1["iterations"] == 12                              # rank 1 passes 45 actions in its 12th episode
    assert r0["train_calls"] == r1["train_calls"] == 7                             # both train from iteration 6 on (rank 1: 24 > 20 rows)
    assert r0["agree_calls"] == r1["agree_calls"] == 24 and r0["n_max"] == r1["n_max"] == 24
    assert r0["n_allreduce"] == r1["n_allreduce"] == 7 * 5
    assert r0["calls"] == r1["calls"] == (["compute", "apply"] * 5 + ["targets"]) * 7
    assert np.array_equal(r0["params"], r1["params"]) and np.array_equal(r0["target"], r1["target"])
    assert (r0["added"], r1["added"]) == (120, 48) and r0["stats_lines"] == r1["stats_lines"] == 12
    # single-process reference: mean gradient of the two shards = params * 1.5
    p = np.linspace(-1, 1, 1000).astype(np.float32)
    for _ in range(35):
        g = (p * np.float32(1.0) + p * np.float32(2.0)) * np.float32(0.5)
        norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        p = p - np.float32(0.01) * (g * np.float32(5.0 / max(norm, 5.0)))
    assert np.allclose(r0["params"], p, rtol=1e-6, atol=1e-7)


def test_agents_loop_in_periodic_mode_counts_local_steps_across_train_calls():
    r0, r1 = _run_world2(_loop_worker, "periodic")
    assert r0["train_calls"] == r1["train_calls"] == 7
    assert r0["n_allreduce"] == r1["n_allreduce"] == 35 // 3                       # parameter averagings after every 3rd minibatch
    assert not np.array_equal(r0["params"], r1["params"])                          # 35 % 3 = 2 local steps since the last meeting


def test_agents_loop_with_rollout_threads_never_waits_for_an_episode_after_burn_in():
    """--async-rollouts: the environments run on rollout threads; the learners iterate (and agree) at their own pace, so the
    iteration count no longer equals the episode count, yet both ranks still issue the same collectives and stop together."""
    r0, r1 = _run_world2(_loop_worker, "async")
    assert r0["iterations"] == r1["iterations"] and r0["train_calls"] == r1["train_calls"] >= 1
    assert r0["agree_calls"] == r1["agree_calls"] == 2 * r0["iterations"]
    assert r0["n_allreduce"] == r1["n_allreduce"] == 5 * r0["train_calls"]
    assert np.array_equal(r0["params"], r1["params"])
    assert r0["added"] > 60 and r1["added"] > 45


def _bad_rank_worker(rank, world, port, out):
    from cartpoleplusplus_amd import ddpg_cartpole as D, distributed
    from tests import fake_abi
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = fake_abi.install(distributed)
    if rank == 1:                                     # this rank cannot form its communicator
        real = fake.cpp_comm_create

        def failing(*a):
            real(*a)                                  # (takes part in the id check so that rank 0 is not left waiting in the double)
            return 3
        fake.cpp_comm_create = failing
        lib, _ = distributed._abi()
        distributed._abi = lambda: (lib, lambda rc: (_ for _ in ()).throw(RuntimeError("ncclCommInitRank -> unhandled system error")) if rc else None)
    opts = D.default_opts(data_parallel=True, batch_size=8)
    agent, _trainer, _nets = _toy_ddpg_agent(rank, 5, opts)
    try:
        distributed.setup_data_parallel(agent, opts, 8)
        out[rank] = "no error"
    except RuntimeError as e:
        out[rank] = str(e)
    dist.destroy_process_group()


def test_a_rank_that_cannot_form_its_communicator_fails_every_rank_instead_of_a_split_fallback():
    """the fallback used to be decided per rank -- one rank on torch's all_reduce, the others inside RCCL: a deadlock."""
    m0, m1 = _run_world2(_bad_rank_worker)
    for m in (m0, m1):
        assert "failed on 1 rank(s)" in m and "rank 1" in m and "ncclCommInitRank" in m
