"""the NAF flavour of op_fuzz.py: random interleavings of episodes (with evictions), fused steps at changing batch sizes, data-parallel
steps (both modes), the reference's op-by-op train call, inference, debug fetches, optimiser-state save / restore."""
import sys, json
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import FakeEnv
from oracle.replay_np import OracleReplayMemory
from cartpoleplusplus_amd import naf_cartpole as F
from cartpoleplusplus_amd.distributed import NativeLearner, Communicator
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
shape = [(16, 16, 3, 2, 3), (20, 20, 3, 1, 2), (32, 32, 3, 1, 3)][seed % 3]
share = bool(seed % 2)
opt = ["GradientDescent", "Momentum", "Adam"][seed % 3]
N = 90
F.set_opts(F.default_opts(batch_size=16, replay_memory_size=N, share_input_state_representation=share, optimiser=opt,
                          optimiser_args=json.dumps({"learning_rate": 0.001} if opt != "Momentum" else {"learning_rate": 0.001, "momentum": 0.9}),
                          use_raw_pixels=True, render_height=shape[0], render_width=shape[1], num_cameras=shape[3], action_repeats=shape[4]))
agent = F.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
agent.initialise_variables(seed=seed); agent.post_var_init_setup()
orm = OracleReplayMemory(N, shape, 2)
rm = agent.replay_memory
learners = {}
def add():
    n = int(rng.integers(1, 15))
    mk = lambda: (rng.integers(0, 256, shape).astype(np.float16) / np.float16(255))
    s0, seq = mk(), [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.integers(0, 3)), mk()) for _ in range(n)]
    rm.add_episode(s0, seq); orm.add_episode(s0, seq)
    k = orm.size()
    assert (rm.insert, rm.full) == (orm.insert, orm.full) and np.array_equal(rm.state_2_idx[:k], orm.state_2_idx[:k])
for _ in range(4):
    add()
ops = []
for step in range(200):
    op = rng.choice(["add", "fused", "dp", "opbyop", "infer", "debug", "optstate"], p=[0.2, 0.25, 0.2, 0.1, 0.1, 0.1, 0.05])
    B = int(rng.choice([1, 3, 8, 16]))
    ops.append((op, B))
    if op == "add":
        add()
    elif op == "fused":
        agent.train_step(B, int(rng.integers(1, 4)))
    elif op == "dp":
        if B not in learners:
            learners[B] = NativeLearner(agent, B, int(F.opts.sample_seed), Communicator.single(agent.naf.ctx) if B != 8 else None, sync_every=2 if B == 3 else 1)
        learners[B].train_step(int(rng.integers(1, 4)))
    elif op == "opbyop":
        agent.naf.train(rm.batch(B))
    elif op == "infer":
        st = rm.state[rm.state_1_idx[int(rng.integers(0, rm.size()))]]
        assert np.isfinite(agent.naf.action_given(st, add_noise=bool(rng.integers(0, 2)))).all()
    elif op == "debug":
        out = agent.naf.debug_values(rm.batch(B))
        assert np.isfinite(out[1])
    else:
        stt = agent.naf.get_optimiser_state(); agent.naf.set_optimiser_state(stt)
    if step % 25 == 24:
        agent.naf.ctx.sync()
        for n in (agent.value_net, agent.naf.mu_net, agent.naf.l_net, agent.target_value_net):
            assert np.isfinite(n.get_params()).all(), (step, ops[-5:])
for l in learners.values():
    l.close()
agent.close()
print("FUZZNAF seed", seed, "shape", shape, "share", share, opt, "ok", len(ops), "ops", flush=True)
