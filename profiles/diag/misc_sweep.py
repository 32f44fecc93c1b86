"""run-only sweep: large batches, NAF over geometries / sharing, the 8-bit store over geometries"""
import sys, json
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import make_pair, FakeEnv
def ddpg(shape, B, **kw):
    agent, _r, _ = make_pair(shape, B, True, replay_size=max(200, 3 * B), **kw)
    try:
        agent.replay_memory.fill_synthetic(max(150, 2 * B), seed=1)
        agent.train_step(B, 2); agent.train_step(B, 3)
        agent.actor.ctx.sync()
        return bool(np.isfinite(agent.critic.get_params()).all() and np.isfinite(agent.actor.get_params()).all())
    finally:
        agent.close()
def naf(shape, B, share):
    from cartpoleplusplus_amd import naf_cartpole as F
    F.set_opts(F.default_opts(batch_size=B, replay_memory_size=200, share_input_state_representation=share, optimiser="Adam",
                              optimiser_args=json.dumps({"learning_rate": 0.001}), use_raw_pixels=True, render_height=shape[0],
                              render_width=shape[1], num_cameras=shape[3], action_repeats=shape[4]))
    agent = F.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
    try:
        agent.initialise_variables(seed=1); agent.post_var_init_setup()
        agent.replay_memory.fill_synthetic(150, seed=1)
        agent.train_step(B, 2); agent.train_step(B, 3)
        agent.value_net.ctx.sync()
        return bool(np.isfinite(agent.value_net.get_params()).all())
    finally:
        agent.close()
cases = [("ddpg B=%d 64x64x18" % B, lambda B=B: ddpg((64, 64, 3, 2, 3), B)) for B in (512, 1024, 2048, 300, 333)]
cases += [("ddpg u8 %dx%dx%d" % (h, w, 3 * c * r), lambda h=h, w=w, c=c, r=r: ddpg((h, w, 3, c, r), 6, replay_store="u8")) for h, w, c, r in ((50, 50, 1, 2), (40, 30, 2, 2), (128, 128, 2, 5), (20, 20, 1, 5), (64, 64, 1, 3))]
cases += [("naf %s %dx%dx%d" % ("shared" if sh else "own", h, w, 3 * c * r), lambda h=h, w=w, c=c, r=r, sh=sh: naf((h, w, 3, c, r), 6, sh))
          for sh in (True, False) for h, w, c, r in ((50, 50, 1, 2), (40, 30, 2, 2), (64, 64, 1, 3), (20, 20, 1, 5), (128, 128, 2, 5), (32, 32, 2, 3))]
for name, fn in cases:
    try:
        print("CASE %-28s %s" % (name, "ok" if fn() else "NON-FINITE"), flush=True)
    except Exception as e:
        print("CASE %-28s ERR %s" % (name, str(e)[:200]), flush=True)
