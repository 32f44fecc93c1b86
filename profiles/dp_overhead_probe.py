#!/usr/bin/env python
"""What the data-parallel protocol costs on ONE GPU (world size 1): the learner's split step (gradient graph -> all-reduce ->
clip + SGD) with and without the RCCL call, and the host time to enqueue a step.  Multi-GPU runs are the driver's."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import numpy as np
from cartpoleplusplus_amd import _lib, ddpg_cartpole as D
from cartpoleplusplus_amd.distributed import GradAllReducer, DataParallelLearner, AgentOps
shape, B = (64, 64, 3, 2, 3), 256
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream(device=0)
ctx = _lib.Context(0, stream=stream.cuda_stream); _lib.set_default_context(ctx)
class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S(shape), S((1, 2))
D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3, batch_size=B,
                          replay_memory_size=22000, sample_seed=1234))
agent = D.DeepDeterministicPolicyGradientAgent(Env()); agent.initialise_variables(seed=42); agent.post_var_init_setup()
agent.replay_memory.fill_synthetic(22000, seed=1234)
reducer = GradAllReducer.for_trainer(agent.trainer, stream)
learner = DataParallelLearner(AgentOps(agent, B, 1234), reducer)
def bench(n=40):
    for _ in range(4): learner.train_step(5)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): learner.train_step(5)
    ctx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * 5) * 1e3
for always in (False, True, False, True):
    reducer.always = always
    print("collective" if always else "no collective", round(bench(), 4), "ms/step")
# host-side cost only: time to enqueue without waiting
reducer.always = True
ctx.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): learner.train_step(5)
t1 = time.perf_counter()
ctx.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.4f ms/step, drained after %.4f ms/step" % ((t1 - t0) / 100 * 1e3, (t2 - t0) / 100 * 1e3))
dist.destroy_process_group()
