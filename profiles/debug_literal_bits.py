"""where do cpp_ddpg_train_rows (literal loop) and cpp_ddpg_train_step(idxs) part ways?  per-k comparison of the parameters."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import make_pair

shape, B = (32, 32, 3, 2, 3), 32
for k in (1, 2, 3, 5):
    ags = []
    for _ in range(2):
        a, _r, _ = make_pair(shape, B, True, seed=3, replay_size=240)
        a.replay_memory.fill_synthetic(200, seed=21)
        ags.append(a)
    lit, fused = ags
    rng = np.random.default_rng(5)
    idxs = rng.integers(0, 200, k * B).astype(np.int32)
    for i in range(k):
        b = lit.replay_memory.batch(idxs=idxs[i * B:(i + 1) * B])
        lit.actor.train(b.state_1); lit.critic.train(b)
    lit.target_actor.update_weights(); lit.target_critic.update_weights()
    fused.train_step(B, k, idxs=idxs)
    for x, y in zip(lit.networks(), fused.networks()):
        px, py = x.get_params(), y.get_params()
        d = np.abs(px - py)
        print("k=%d %-14s equal=%s maxdiff=%.3e ndiff=%d first=%s" % (k, x.namespace, np.array_equal(px, py), d.max(), int((d > 0).sum()),
                                                                    np.flatnonzero(d > 0)[:5]))
    lit.close(); fused.close()
