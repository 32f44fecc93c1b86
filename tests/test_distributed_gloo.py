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

    def __init__(self, rank, grad_tensor):
        self.rank, self.g = rank, grad_tensor
        self.params = np.linspace(-1, 1, grad_tensor.numel()).astype(np.float32)
        self.target = self.params.copy()
        self.calls = []

    def sample_and_compute(self):
        self.calls.append("compute")
        data = np.full_like(self.params, float(self.rank + 1))       # the rank's "replay shard"
        self.g.copy_(torch.from_numpy(self.params * data))

    def apply(self, grad_scale):
        self.calls.append("apply")
        g = self.g.numpy() * np.float32(grad_scale)
        norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        g = g * np.float32(5.0 / max(norm, 5.0))                      # util.py:47-50
        self.params = self.params - np.float32(0.01) * g

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
