#!/usr/bin/env python
"""conv1 dW ablation (see ablate_conv1.py): CARTPOLEPP_ABLATION=<variant name> python profiles/ablate_conv1_dw.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartpoleplusplus_amd import ddpg_cartpole as D

shape, B = (64, 64, 3, 2, 3), 256
D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2,
                          action_repeats=3, batch_size=B, replay_memory_size=600))
class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S(shape), S((1, 2))
agent = D.DeepDeterministicPolicyGradientAgent(Env())
agent.initialise_variables(seed=1); agent.post_var_init_setup()
agent.replay_memory.fill_synthetic(500, seed=3)
ctx = agent.actor.ctx
for _ in range(2):
    agent.train_step(B, 1)
ctx.sync(); ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    agent.train_step(B, 1)
ctx.prof_enable(False)
for k, (ms, n) in sorted(ctx.prof_read().items()):
    if "conv1" in k or "conv2_d" in k:
        print("%-16s %8.2f us/launch  (%d launches)" % (k, 1e3 * ms / n, n))
