"""Grids, bands and per-workgroup partial buffers are planned from the device's CU count (ctx->num_cus): whole-image units while they fit
one resident round, bands otherwise, and `conv_dw_partial_floats` sizes the dW partial buffers for num_cus * 4 workgroups per network.
ADVICE r5 (medium): `conv_dw_rs.h`'s launcher did not check its grid against that capacity -- on a 32-CU CPX partition cfg5's B = 512 at
64-wide rows wrote past the buffer.  The ablation build can plan as for a smaller device on the whole chip (`CPP_NUM_CUS`): the same
minibatch with the plan of 32 / 8 CUs and with the real one must give the same gradients up to summation order (different band / unit
splits, the clamping fallbacks where a wave-per-unit grid does not fit), and every guard band must survive (the library registers its
allocations between guard bands and the kernels' range checks are against them)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SNIPPET = r"""
import sys
import numpy as np
from tests.helpers import make_pair
shape, B = eval(sys.argv[1]), int(sys.argv[2])
agent, _ref, _ = make_pair(shape, B, True, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=21)
idxs = np.arange(B, dtype=np.int32)
agent.train_step(B, 1, idxs=idxs)
g = np.concatenate([agent.actor.get_grads(), agent.critic.get_grads()])      # the first minibatch's: same parameters, same rows in both plans
agent.train_step(B, 2)                                   # ... and a graph-replayed step with device-drawn rows behind it
p = np.concatenate([n.get_params() for n in agent.networks()])
np.savez(sys.argv[3], grads=g, params=p)
agent.close()
"""


def _run(tmp_path, name, shape, B, extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / (name + ".npz"))
    r = subprocess.run([sys.executable, "-c", _SNIPPET, repr(shape), str(B), out], cwd=root,
                       env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-1500:]
    return dict(np.load(out))


@pytest.mark.parametrize("shape,B,cus", [
    ((64, 64, 3, 2, 3), 256, 32),          # cfg3 as a CPX partition would plan it: conv2's wave-per-unit dW does not fit 128 partials
    ((64, 64, 3, 2, 3), 256, 8),           # ... and a plan so small that every dW launcher clamps or falls back
    ((64, 64, 3, 1, 3), 64, 32),           # 9 channels
    ((128, 128, 3, 2, 5), 16, 32),         # cfg5's geometry (64-wide conv2 rows: two column units per band)
], ids=["cfg3-B256-32cu", "cfg3-B256-8cu", "9ch-B64-32cu", "cfg5-geometry-B16-32cu"])
def test_a_smaller_devices_plan_gives_the_same_gradients(tmp_path, shape, B, cus):
    small = _run(tmp_path, "small", shape, B, {"CPP_NUM_CUS": str(cus)})
    full = _run(tmp_path, "full", shape, B, {})
    # (gradients of the first minibatch: summation order only; parameters after three updates: those differences times the learning rates,
    # and whatever pooling routes the last bits move in minibatches two and three)
    assert not np.array_equal(small["grads"], full["grads"]), "the smaller plan changed nothing: is CPP_NUM_CUS read?"      # (other unit splits: other roundings)
    for k, rel in (("grads", 3e-6), ("params", 2e-5)):
        a, b = small[k].astype(np.float64), full[k].astype(np.float64)
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        h = len(a) // 2 if k == "grads" else len(a)
        for lo, hi in ((0, h), (h, len(a))) if k == "grads" else ((0, len(a)),):
            d = np.abs(a[lo:hi] - b[lo:hi]).max()
            assert d <= rel * np.abs(b[lo:hi]).max(), (k, lo, d, np.abs(b[lo:hi]).max())
