#!/usr/bin/env python
"""NAF (BASELINE.json configs[3]: naf_cartpole.py pixel obs 64x64x18, batch 256, shared conv trunk) throughput
on one GPU: hipGraph-replayed inner steps (naf_cartpole.py:367-373), device-resident replay, synthetic data.
Not the headline metric (that is bench.py / DDPG); kept next to the profiles as a measured data point."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartpoleplusplus_amd import naf_cartpole as F

shape, B = (64, 64, 3, 2, 3), 256
share = "--own-trunks" not in sys.argv
F.set_opts(F.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3,
                          batch_size=B, replay_memory_size=22000, share_input_state_representation=share,
                          optimiser="Momentum", optimiser_args=json.dumps({"learning_rate": 0.01, "momentum": 0.9})))
class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S(shape), S((1, 2))
agent = F.NormalizedAdvantageFunctionAgent(Env())
agent.initialise_variables(seed=42); agent.post_var_init_setup()
agent.replay_memory.fill_synthetic(22000, seed=1234)
ctx = agent.value_net.ctx
for _ in range(6):
    agent.train_step(B, 5)
ctx.sync()
t0 = time.perf_counter()
groups = 40
for _ in range(groups):
    agent.train_step(B, 5)
ctx.sync()
dt = time.perf_counter() - t0
st = agent.naf.last_stats()
print(json.dumps({"metric": "NAF training steps/sec, 64x64x18 pixel obs, batch=256, %s" % ("shared trunk" if share else "own trunks"),
                  "value": round(groups * 5 / dt, 2), "ms_per_step": round(1e3 * dt / (groups * 5), 4),
                  "loss": float(st[0]), "nonfinite": float(st[2])}))
