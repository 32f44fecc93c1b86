"""How much of the step is idle GPU?  Two independent DDPG learners (own context, own HIP stream, own replay store) in ONE process on
one GPU, their fused steps (hipGraph replays) enqueued alternately: if the whole-job rate is above one learner's, the second stream's
kernels found CUs the first left idle (the latency-bound tail, the half-filled launches)."""
import json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from cartpoleplusplus_amd import _lib
from cartpoleplusplus_amd import ddpg_cartpole as D

SHAPE, B, ROWS = (64, 64, 3, 2, 3), 256, 6000


class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S(SHAPE), S((1, 2))


def make(seed):
    stream = torch.cuda.Stream(device=0)
    ctx = _lib.Context(0, stream=stream.cuda_stream)
    _lib.set_default_context(ctx)
    D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3, batch_size=B,
                              replay_memory_size=ROWS, sample_seed=seed))
    ag = D.DeepDeterministicPolicyGradientAgent(Env())
    ag.initialise_variables(seed=42)
    ag.post_var_init_setup()
    ag.replay_memory.fill_synthetic(ROWS, seed=seed)
    return ag, ctx, stream


def timed(agents, ctxs, groups):
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(groups):
        for ag in agents:
            ag.train_step(B, 5)
    for c in ctxs: c.sync()
    return 5 * groups * len(agents) / (time.perf_counter() - t0)


a1, c1, s1 = make(1)
a2, c2, s2 = make(2)
for ag, c in ((a1, c1), (a2, c2)):
    _lib.set_default_context(c)
    for _ in range(40): ag.train_step(B, 5)
    c.sync()
out = {}
for rep in range(2):
    out["one_%d" % rep] = round(timed([a1], [c1], 60), 1)
    out["two_%d" % rep] = round(timed([a1, a2], [c1, c2], 60), 1)
print(json.dumps(out))
