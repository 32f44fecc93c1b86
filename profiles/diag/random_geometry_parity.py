import sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import fused_step_against_f64_oracle
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed0)
bad = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    H, W = int(rng.integers(8, 73)), int(rng.integers(8, 73))
    cams, reps = int(rng.integers(1, 3)), int(rng.integers(1, 6))
    B = int(rng.integers(1, 10))
    shape = (H, W, 3, cams, reps)
    try:
        rep = fused_step_against_f64_oracle(shape, B, rows=60, graph=bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1000)))
        print("GEO", shape, "B", B, "ok errq %.1e relc %.1e" % (rep["err_q"], rep["rel_critic_grads"]), flush=True)
    except Exception as e:
        bad += 1
        print("GEO", shape, "B", B, "FAIL", str(e).replace("\n", " | ")[:1600], flush=True)
print("GEODONE bad", bad, flush=True)
