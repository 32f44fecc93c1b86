#!/usr/bin/env python
"""conv1 forward ablation: time the kernel (HIP events) for library variants built with
-DCONV_ABLATE_NOMFMA (staging + epilogue only) / -DCONV_ABLATE_NOSTAGE (MFMA phase only).
usage: CARTPOLEPP_ABLATION=<variant name> python profiles/ablate_conv1.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartpoleplusplus_amd import _lib, ddpg_cartpole as D

shape, B = (64, 64, 3, 2, 3), 256
D.set_opts(D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2,
                          action_repeats=3, batch_size=B))
net = D.ActorNetwork("actor", D.base_network.Placeholder([None] + list(shape)), 2)
net.initialise_variables(np.random.RandomState(0))
s = (np.random.RandomState(1).randint(0, 256, (B,) + shape).astype(np.float16) / np.float16(255))
ctx = net.ctx
for _ in range(3):
    net.forward(s)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(10):
    net.forward(s)
ctx.prof_enable(False)
for k, (ms, n) in sorted(ctx.prof_read().items()):
    print("%-16s %8.2f us/launch  (%d launches)" % (k, 1e3 * ms / n, n))
