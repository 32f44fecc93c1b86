"""conv2's dX (the pooled-gradient buffer conv1's dW reads) after one fused step, dumped for comparison across builds / switches:
  CARTPOLEPP_ABLATION=1 [CPP_CONV_DXRS=0 | CPP_CONV2_PAIR=0] python profiles/diag/dxrs_pair_diff.py out.npy"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.helpers import make_pair
from cartpoleplusplus_amd._lib import lib, check, ptr
shape, B = (64, 64, 3, 2, 3), 256
agent, _ref, _ = make_pair(shape, B, True, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=11)
idxs = np.arange(B, dtype=np.int32)
agent.train_step(B, 1, idxs=idxs)
out = {}
for name, net in (("actor", agent.actor), ("critic", agent.critic)):
    d = np.empty((B, 32, 32, 10), np.float32)
    check(lib.cpp_net_get_pool(net.handle, 21, B, ptr(d)))
    out[name] = d
np.save(sys.argv[1], np.stack([out["actor"], out["critic"]]))
print("saved", sys.argv[1], [float(np.abs(v).sum()) for v in out.values()])
agent.close()
