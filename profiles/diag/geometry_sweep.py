"""which (render size, cameras, action repeats) run through the fused step (and batch-norm mode)?  Prints ok / the loud error."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import make_pair
sizes = [(50, 50), (64, 64), (32, 32), (40, 30), (84, 84), (96, 96), (100, 100), (128, 128), (28, 28), (20, 20)]
for cams, reps in ((1, 1), (1, 2), (1, 3), (2, 2), (1, 4), (1, 5), (2, 3), (2, 4), (2, 5)):
    row = []
    for H, W in sizes:
        for bn in (False, True):
            shape, B = (H, W, 3, cams, reps), 4
            try:
                agent, _ref, _ = make_pair(shape, B, True, replay_size=60, use_batch_norm=bn)
                try:
                    agent.replay_memory.fill_synthetic(40, seed=1)
                    agent.train_step(B, 2); agent.train_step(B, 2)
                    agent.actor.ctx.sync()
                    ok = bool(np.isfinite(agent.critic.get_params()).all())
                    row.append("%dx%d%s:%s" % (H, W, "+bn" if bn else "", "ok" if ok else "NAN"))
                finally:
                    agent.close()
            except Exception as e:
                row.append("%dx%d%s:ERR(%s)" % (H, W, "+bn" if bn else "", str(e)[:60]))
    print("C=%d" % (3 * cams * reps), " ".join(row), flush=True)
