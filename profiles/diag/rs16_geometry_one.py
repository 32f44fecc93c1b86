"""re-run ONE draw of rs16_geometry_parity.py (seed0, index) and print the full report or failure text"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.helpers import fused_step_against_f64_oracle
seed0, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for i in range(want + 1):
    H = int(rng.choice([16, 18, 20, 22, 24, 28, 30, 32, 34, 40, 46, 48, 50, 56, 62, 64, 66, 72, 80, 96]))
    cams, reps = [(1, 1), (1, 2), (1, 3), (2, 2), (1, 4), (2, 3)][int(rng.integers(0, 6))]
    B = int(rng.integers(1, 12))
    graph = bool(rng.integers(0, 2)); seed = int(rng.integers(0, 1000)); fill = "render" if rng.integers(0, 3) == 0 else "noise"
    if i < want: continue
    shape = (H, 64, 3, cams, reps)
    print("DRAW", i, shape, "B", B, "graph", graph, "seed", seed, fill, flush=True)
    try:
        rep = fused_step_against_f64_oracle(shape, B, rows=60, graph=graph, seed=seed, fill=fill)
        print("OK", rep)
    except Exception as e:
        print("FAIL", str(e)[:3000])
