"""conv1's weight / bias gradient after one fused step, dumped for comparison across switches:
  CARTPOLEPP_ABLATION=1 [CPP_CONV1_DWRS=0] python profiles/diag/dw16rs_diff.py out.npy"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import make_pair
shape, B = (64, 64, 3, 2, 3), 256
agent, _ref, _ = make_pair(shape, B, True, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=11)
idxs = np.arange(B, dtype=np.int32)
agent.train_step(B, 1, idxs=idxs)
ga, gc = agent.actor.get_grads(), agent.critic.get_grads()
np.save(sys.argv[1], np.stack([ga[:4510], gc[:4510]]))
print("saved", [float(np.abs(g[:4510]).sum()) for g in (ga, gc)])
agent.close()
