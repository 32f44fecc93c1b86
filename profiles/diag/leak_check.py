import sys, subprocess, re
sys.path.insert(0, ".")
import numpy as np
from tests.helpers import make_pair
from cartpoleplusplus_amd.distributed import NativeLearner, Communicator
from cartpoleplusplus_amd import ddpg_cartpole as D
def used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], stdout=subprocess.PIPE).stdout.decode()
    return int(re.search(r"Used Memory \(B\): (\d+)", out).group(1))
def cycle():
    agent, _r, _ = make_pair((64, 64, 3, 2, 3), 32, True, replay_size=600)
    agent.replay_memory.fill_synthetic(500, seed=1)
    agent.train_step(32, 2); agent.train_step(32, 2)
    lr = NativeLearner(agent, 32, int(D.opts.sample_seed), Communicator.single(agent.trainer.ctx))
    lr.train_step(2); lr.train_step(2); lr.close()
    agent.actor.ctx.sync()
    agent.close()
cycle(); cycle()
u0 = used()
for i in range(60):
    cycle()
u1 = used()
print("LEAK used before %.1f MB after 60 create/train/close cycles %.1f MB delta %.1f MB" % (u0 / 1e6, u1 / 1e6, (u1 - u0) / 1e6), flush=True)
