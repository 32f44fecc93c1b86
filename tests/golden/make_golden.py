#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/.

The reference (Python 2 + TensorFlow 0.x) cannot be imported or run in this environment, so:
  * replay_known_answers.json -- the known answers held by the reference's own test
    (/root/reference/replay_memory_test.py:32-56 and :58-86), typed in as data;
  * ddpg_step_*.npz -- seeded inputs, parameters and the outputs of ONE inner train step
    (ddpg_cartpole.py:331-337 with batches_per_step = 2) computed by the float64 oracle
    (oracle/ddpg_np.py, which tests/test_oracle_vs_torch.py cross-checks against torch autograd).
    They pin the oracle against regressions and give the GPU tests fixed vectors; they are NOT
    reference outputs (network parity is unpinned by the reference, see oracle/__init__.py).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ddpg_np as O  # noqa: E402

CASES = {
    "pixel_8x8x6_B4": dict(shape=(8, 8, 3, 1, 2), B=4, pixel=True),
    "pixel_12x10x9_B3": dict(shape=(12, 10, 3, 1, 3), B=3, pixel=True),
    "lowdim_28_B5": dict(shape=(2, 2, 7), B=5, pixel=False),
}


def make_case(name, shape, B, pixel, seed=1234):
    rng = np.random.default_rng(seed)
    kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:]))) if pixel else \
        dict(pixel=False, state_elems=int(np.prod(shape)))
    aspec, cspec = O.NetSpec("actor", 2, [100, 100, 50], **kw), O.NetSpec("critic", 2, [100, 100, 50], **kw)
    af = O.init_params(aspec, rng) + rng.normal(0, 0.05, aspec.num_params()).astype(np.float32)
    cf = O.init_params(cspec, rng) + rng.normal(0, 0.05, cspec.num_params()).astype(np.float32)
    taf = af + rng.normal(0, 0.01, af.shape).astype(np.float32)
    tcf = cf + rng.normal(0, 0.01, cf.shape).astype(np.float32)
    batches = [O.synthetic_batch(rng, B, shape, 2, pixel) for _ in range(2)]
    agent = O.DDPG(aspec, cspec, af, cf, np.float64)
    agent.set_targets(taf, tcf)
    outs = agent.train_step(batches)
    d = dict(actor=af, critic=cf, target_actor=taf, target_critic=tcf,
             new_actor=agent.actor.flat(), new_critic=agent.critic.flat(),
             new_target_actor=agent.target_actor.flat(), new_target_critic=agent.target_critic.flat())
    for i, (b, o) in enumerate(zip(batches, outs)):
        for k, v in zip(("s1", "a", "r", "mask", "s2"), b):
            d["b%d_%s" % (i, k)] = v
        for k in ("actions", "q_actor", "dq_da", "q", "td", "actor_grads", "critic_grads"):
            d["o%d_%s" % (i, k)] = np.asarray(o[k])
        d["o%d_loss" % i] = np.float64(o["loss"])
    np.savez_compressed(os.path.join(HERE, "ddpg_step_%s.npz" % name), **d)


def main():
    known = {
        "source": "/root/reference/replay_memory_test.py",
        "setup": {"buffer_size": 3, "state_shape": [2, 3], "action_dim": 2, "load_factor": 2},
        "adds_to_full": {
            "lines": "32-56",
            "initial_state": [[11, 12, 13], [14, 15, 16]],
            "action_reward_state": [[17, 18, [[21, 22, 23], [24, 25, 26]]],
                                    [27, 28, [[31, 32, 33], [34, 35, 36]]],
                                    [37, 38, [[41, 42, 43], [44, 45, 46]]]],
            "expect": {"size": 3, "insert": 0, "full": True, "state_first_elements": [11, 21, 31, 41]}},
        "adds_over_full": {
            "lines": "58-86",
            "episodes": [{"first": 0, "steps": [1, 2, 3, 4]}, {"first": 5, "steps": [6, 7, 8]}],
            "state_rule": "s_for(i) = (1..6) + 10*i reshaped (2,3); action = 10*i+7; reward = 10*i+8",
            "expect": {"size": 3, "reward": [[88], [68], [78]], "terminal_mask": [[0], [1], [1]]}},
    }
    with open(os.path.join(HERE, "replay_known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)
    for name, c in CASES.items():
        make_case(name, **c)


if __name__ == "__main__":
    main()
