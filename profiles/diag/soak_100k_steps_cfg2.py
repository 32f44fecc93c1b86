import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from cartpoleplusplus_amd import _lib
from cartpoleplusplus_amd import ddpg_cartpole as D
class Env(object):
    class S(object):
        def __init__(self, s): self.shape = tuple(s)
    observation_space, action_space = S((64, 64, 3, 1, 3)), S((1, 2))
D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=1, action_repeats=3, batch_size=256, replay_memory_size=22000))
agent = D.DeepDeterministicPolicyGradientAgent(Env())
agent.initialise_variables(seed=42); agent.post_var_init_setup()
agent.replay_memory.fill_synthetic(22000, seed=1234)
t0 = time.time()
for i in range(20000):
    agent.train_step(256, 5)
    if i % 5000 == 4999:
        st = agent.trainer.last_stats()
        p = agent.actor.get_params(); q = agent.critic.get_params()
        print(i + 1, "groups", round(time.time() - t0, 1), "s loss/norms", st, "finite", bool(np.isfinite(p).all() and np.isfinite(q).all()), "|theta|", float(np.abs(p).max()), float(np.abs(q).max()), flush=True)
import subprocess
print(subprocess.run(["rocm-smi", "--showmeminfo", "vram"], stdout=subprocess.PIPE).stdout.decode()[-300:])
agent.close()
