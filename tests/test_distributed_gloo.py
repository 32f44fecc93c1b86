"""world_size-2 gloo test of the data-parallel learner protocol (runs on CPU): every rank computes its
own gradients, ONE flat buffer is sum-all-reduced, every rank applies clip+SGD to the mean and the
replicas stay bit-identical -- the structure bench.py uses over RCCL/xGMI with the HIP ops."""
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
