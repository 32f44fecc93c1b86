"""crash / NaN fuzz: random interleavings of the public operations on one agent (episodes added with evictions, fused steps at
changing batch sizes, data-parallel steps, the reference's op-by-op calls, inference, debug fetches) -- nothing may fault, raise
or go non-finite; the replay bookkeeping is checked against the oracle memory as it goes."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import make_pair
from oracle.replay_np import OracleReplayMemory
from cartpoleplusplus_amd import ddpg_cartpole as D
from cartpoleplusplus_amd.distributed import NativeLearner, Communicator
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
shape = [(16, 16, 3, 2, 3), (20, 20, 3, 1, 2), (32, 32, 3, 1, 3)][seed % 3]
N = 90
agent, _ref, _ = make_pair(shape, 16, True, replay_size=N, seed=seed)
orm = OracleReplayMemory(N, shape, 2)
rm = agent.replay_memory
learners = {}
def add():
    n = int(rng.integers(1, 15))
    mk = lambda: (rng.integers(0, 256, shape).astype(np.float16) / np.float16(255))
    s0, seq = mk(), [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.integers(0, 3)), mk()) for _ in range(n)]
    rm.add_episode(s0, seq); orm.add_episode(s0, seq)
    k = orm.size()
    assert (rm.insert, rm.full) == (orm.insert, orm.full) and np.array_equal(rm.state_1_idx[:k], orm.state_1_idx[:k])
for _ in range(4):
    add()
ops = []
for step in range(250):
    op = rng.choice(["add", "fused", "dp", "opbyop", "infer", "debug", "batch"], p=[0.2, 0.25, 0.2, 0.1, 0.1, 0.1, 0.05])
    B = int(rng.choice([1, 3, 8, 16]))
    ops.append((op, B))
    if op == "add":
        add()
    elif op == "fused":
        agent.train_step(B, int(rng.integers(1, 4)))
    elif op == "dp":
        if B not in learners:
            learners[B] = NativeLearner(agent, B, int(D.opts.sample_seed), Communicator.single(agent.trainer.ctx) if B % 2 else None, overlap=bool(B == 3))
        learners[B].train_step(int(rng.integers(1, 4)))
    elif op == "opbyop":
        b = rm.batch(B)
        agent.actor.train(b); agent.critic.train(b)
        agent.target_actor.update_weights(); agent.target_critic.update_weights()
    elif op == "infer":
        st = rm.state[rm.state_1_idx[int(rng.integers(0, rm.size()))]]
        a = agent.actor.action_given(st, add_noise=bool(rng.integers(0, 2)))
        assert np.isfinite(a).all()
    elif op == "debug":
        b = rm.batch(B)
        out = agent.critic.check_loss(b)
        assert np.isfinite(out[0])
    else:
        b = rm.batch(B); ob = orm.batch(idxs=b.idxs if hasattr(b, "idxs") and b.idxs is not None else None) if False else None
        assert np.isfinite(np.asarray(b.state_1, np.float32)).all()
    if step % 25 == 24:
        agent.actor.ctx.sync()
        for n in agent.networks():
            assert np.isfinite(n.get_params()).all(), (step, ops[-5:])
for l in learners.values():
    l.close()
agent.close()
print("FUZZ seed", seed, "shape", shape, "ok", len(ops), "ops", flush=True)
