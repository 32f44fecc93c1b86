#!/usr/bin/env python
"""The drop-in inner step (ddpg_cartpole.py:331-337) through the host boundary, beside bench.py's device-resident number:
  device : batch = replay.batch(B); actor.train(batch); critic.train(batch)  -- the Batch stays on the device, op-by-op launches
  host   : the same with plain numpy columns handed in (a namedtuple like the reference's Batch): every step uploads
           2 x 37.7 MB of f16 states over PCIe -- the PCIe-inclusive rate DESIGN.md quotes (never bench.py's `value`)
cfg3 shapes (64x64x18, B = 256), synthetic replay."""
import collections, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartpoleplusplus_amd import ddpg_cartpole as D

shape, B = (64, 64, 3, 2, 3), 256
D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3,
                          batch_size=B, replay_memory_size=4000))
class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S(shape), S((1, 2))
agent = D.DeepDeterministicPolicyGradientAgent(Env())
agent.initialise_variables(seed=42); agent.post_var_init_setup()
agent.replay_memory.fill_synthetic(4000, seed=1234)
ctx = agent.trainer.ctx
HostBatch = collections.namedtuple("Batch", "state_1 action reward terminal_mask state_2")

def step_device():
    batch = agent.replay_memory.batch(B)
    agent.actor.train(batch)
    agent.critic.train(batch)

dev = agent.replay_memory.batch(B)
host = HostBatch(np.array(dev.state_1), np.array(dev.action), np.array(dev.reward), np.array(dev.terminal_mask), np.array(dev.state_2))
def step_host():
    agent.actor.train(host.state_1)
    agent.critic.train(host)

out = {}
for name, fn, n in (("device", step_device, 200), ("host", step_host, 40)):
    for _ in range(5):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if (i + 1) % 5 == 0:
            agent.target_actor.update_weights(); agent.target_critic.update_weights()
    ctx.sync()
    dt = time.perf_counter() - t0
    out[name] = {"steps_per_s": round(n / dt, 1), "ms_per_step": round(1e3 * dt / n, 3)}
out["host_bytes_per_step"] = int(2 * host.state_1.nbytes + host.state_1.nbytes)
print(json.dumps(out))
