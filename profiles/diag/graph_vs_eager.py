"""diagnostic: with learning rates 0 the parameters never move, so the gradients of a minibatch are a pure function of its
rows: a hipGraph-replayed fused step (device Philox rows) must leave bit-identical gradients to an eager fused step fed the
same rows.  Prints per-variable differences."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import make_pair, per_var_report
from cartpoleplusplus_amd import _lib

shape, B, rows = (64, 64, 3, 2, 3), 256, 2500
store = sys.argv[1] if len(sys.argv) > 1 else "f16"
agent, _ref, (aspec, cspec) = make_pair(shape, B, True, replay_size=rows + 50, replay_store=store,
                                        actor_learning_rate=0.0, critic_learning_rate=0.0, target_update_rate=0.0)
rm = agent.replay_memory
rm.fill_synthetic(rows, seed=21)

def last_rows():
    idxs = np.empty(B, np.int32)
    _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
    return idxs

def grads():
    return agent.actor.get_grads(), agent.critic.get_grads()

def show(tag, ga, gb):
    for spec, a, b, nm in ((aspec, ga[0], gb[0], "actor"), (cspec, ga[1], gb[1], "critic")):
        bad = [(n, m, r) for n, m, r in per_var_report(spec, a, b) if m > 0]
        print(tag, nm, "IDENTICAL" if not bad else " ".join("%s:%.2e" % (n.split("/")[0] + n.split("/")[1][0], r) for n, m, r in bad))

agent.train_step(B, 1)                      # eager + capture
r0 = last_rows(); g0 = grads()
agent.train_step(B, 1, idxs=r0); show("eager-philox vs eager-rows      ", g0, grads())
for k in range(3):
    agent.train_step(B, 1)                  # graph replay
    rk = last_rows(); gk = grads()
    agent.train_step(B, 1, idxs=rk); ge = grads()
    show("graph replay %d vs eager-rows     " % k, gk, ge)
    agent.train_step(B, 1, idxs=rk); show("eager-rows twice                ", ge, grads())
agent.actor.ctx.prof_enable(True)
agent.train_step(B, 1); rp = last_rows(); gp = grads()
agent.actor.ctx.prof_enable(False)
agent.train_step(B, 1, idxs=rp); show("prof(eager philox) vs eager-rows", gp, grads())
agent.close()
