"""random 64-wide geometries (the widths conv_rs16.h takes) at every channel count and height against the float64 oracle: round 6's
channel instances of conv1 forward on heights other than 64 (the general blocks at both ends of the row walk), odd batches, renders"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import fused_step_against_f64_oracle
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed0)
bad = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    H = int(rng.choice([16, 18, 20, 22, 24, 28, 30, 32, 34, 40, 46, 48, 50, 56, 62, 64, 66, 72, 80, 96]))
    cams, reps = [(1, 1), (1, 2), (1, 3), (2, 2), (1, 4), (2, 3)][int(rng.integers(0, 6))]
    B = int(rng.integers(1, 12))
    shape = (H, 64, 3, cams, reps)
    try:
        rep = fused_step_against_f64_oracle(shape, B, rows=60, graph=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)),
                                            fill=("render" if rng.integers(0, 3) == 0 else "noise"))
        print("GEO", shape, "B", B, "ok errq %.1e relc %.1e" % (rep["err_q"], rep["rel_critic_grads"]), flush=True)
    except Exception as e:
        bad += 1
        print("GEO", shape, "B", B, "FAIL", str(e).replace("\n", " | ")[:1200], flush=True)
print("GEODONE bad", bad, flush=True)
