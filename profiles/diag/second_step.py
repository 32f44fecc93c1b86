import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import fused_step_against_f64_oracle as f
shape = (64, 64, 3, 2, 3)
for name, kw in (("first step, eager rows", dict(graph=False, warm="none")),
                 ("second step after eager-rows step, eager rows", dict(graph=False, warm="rows")),
                 ("second step after philox step, eager rows", dict(graph=False, warm="philox-eager")),
                 ("graph replay", dict(graph=True))):
    rep = f(shape, 256, rows=2500, seed=0, report_only=True, **kw)
    print(name, rep["events"], "\n   actor ", rep["actor"], "\n   critic", rep["critic"], flush=True)
