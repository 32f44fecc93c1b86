#!/usr/bin/env python
"""The reference's inner loop as the reference writes it, beside bench.py's fused number (cfg3: 64x64x18, B = 256; cfg4 for NAF):
  literal : ddpg_cartpole.py:331-337 verbatim -- batch = replay_memory.batch(B); actor.train(batch.state_1); critic.train(batch);
            target updates every 5th minibatch.  `batch.state_1` is a device-resident StateColumn, the actor's update is deferred
            and runs with the critic's as cpp_ddpg_train_rows (one hipGraph replay per minibatch); B row indexes cross PCIe.
  fused   : agent.train_step(B, 5)  (device-drawn rows, one hipGraph replay per 5 minibatches) -- bench.py's path
  unfused : actor.train(batch); critic.train(batch) on the whole Batch: the op-by-op train ops on a gathered device minibatch
  host    : plain numpy columns handed in (a namedtuple like the reference's Batch): 2 x 37.7 MB + 37.7 MB of f16 states over PCIe
            per minibatch -- the PCIe-inclusive rate DESIGN.md quotes (never bench.py's `value`)
  naf_literal / naf_fused : naf_cartpole.py:365-373 verbatim against agent.train_step (cfg4)
`state_bytes_over_pcie` counts the bytes of state columns that were downloaded or uploaded during the timed literal loops."""
import collections, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartpoleplusplus_amd import ddpg_cartpole as D, naf_cartpole as N, replay_memory as R

shape, B, ROWS = (64, 64, 3, 2, 3), 256, 22000
class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S(shape), S((1, 2))
common = dict(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3, batch_size=B, replay_memory_size=ROWS)

# count every state column that crosses PCIe (Batch downloads; DeviceBatch uploads)
pcie = {"bytes": 0}
_hs, _up = R.Batch._host_states, R.DeviceBatch.upload
def _count_down(self):
    fresh = self._states is None
    out = _hs(self)
    if fresh:
        pcie["bytes"] += sum(v.nbytes for v in out.values())
    return out
def _count_up(self, s1, *a, **k):
    pcie["bytes"] += 2 * np.asarray(s1).nbytes
    return _up(self, s1, *a, **k)
R.Batch._host_states, R.DeviceBatch.upload = _count_down, _count_up

def timed(fn, n, sync, warm=10):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    dt = time.perf_counter() - t0
    return {"steps_per_s": round(5 * n / dt, 1), "ms_per_step": round(1e3 * dt / (5 * n), 4)}

out = {}
D.set_opts(D.default_opts(**common))
agent = D.DeepDeterministicPolicyGradientAgent(Env())
agent.initialise_variables(seed=42); agent.post_var_init_setup()
agent.replay_memory.fill_synthetic(ROWS, seed=1234)
ctx = agent.trainer.ctx
HostBatch = collections.namedtuple("Batch", "state_1 action reward terminal_mask state_2")

def literal():                  # ddpg_cartpole.py:331-337
    for _ in range(5):
        batch = agent.replay_memory.batch(B)
        agent.actor.train(batch.state_1)
        agent.critic.train(batch)
    agent.target_actor.update_weights()
    agent.target_critic.update_weights()
def fused():
    agent.train_step(B, 5)
def unfused():
    for _ in range(5):
        batch = agent.replay_memory.batch(B)
        agent.actor.train(batch)
        agent.critic.train(batch)
    agent.target_actor.update_weights(); agent.target_critic.update_weights()
dev = agent.replay_memory.batch(B)
host = HostBatch(np.array(dev.state_1), np.array(dev.action), np.array(dev.reward), np.array(dev.terminal_mask), np.array(dev.state_2))
def hostfed():
    for _ in range(5):
        agent.actor.train(host.state_1)
        agent.critic.train(host)
    agent.target_actor.update_weights(); agent.target_critic.update_weights()

out["fused"] = timed(fused, 60, ctx.sync, warm=45)
pcie["bytes"] = 0
out["literal"] = timed(literal, 60, ctx.sync, warm=45)
out["literal"]["state_bytes_over_pcie"] = pcie["bytes"]
out["literal"]["fused_pairs"] = agent.trainer.fused_pairs
out["literal_over_fused"] = round(out["literal"]["steps_per_s"] / out["fused"]["steps_per_s"], 4)
out["unfused"] = timed(unfused, 20, ctx.sync, warm=4)
out["host"] = timed(hostfed, 6, ctx.sync, warm=1)
out["host_bytes_per_step"] = int(3 * host.state_1.nbytes)
agent.close()

N.set_opts(N.default_opts(share_input_state_representation=True, optimiser="Momentum",
                          optimiser_args=json.dumps({"learning_rate": 0.01, "momentum": 0.9}), **common))
nagent = N.NormalizedAdvantageFunctionAgent(Env())
nagent.initialise_variables(seed=42); nagent.post_var_init_setup()
nagent.replay_memory.fill_synthetic(ROWS, seed=1234)
def naf_literal():              # naf_cartpole.py:365-373
    losses = []
    for _ in range(5):
        batch = nagent.replay_memory.batch(B)
        losses.append(nagent.naf.train(batch))
    nagent.target_value_net.update_weights()
def naf_fused():
    nagent.train_step(B, 5)
nsync = nagent.naf.ctx.sync
out["naf_fused"] = timed(naf_fused, 60, nsync, warm=45)
pcie["bytes"] = 0
out["naf_literal"] = timed(naf_literal, 60, nsync, warm=45)
out["naf_literal"]["state_bytes_over_pcie"] = pcie["bytes"]
out["naf_literal_over_fused"] = round(out["naf_literal"]["steps_per_s"] / out["naf_fused"]["steps_per_s"], 4)
nagent.close()
print(json.dumps(out))
