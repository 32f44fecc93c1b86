"""Round 5's backward kernels (csrc/conv_dx_rs.h, conv_dw_rs.h, conv_dw16_rs.h: one wave per unit, bf16 / f16 pieces on the matrix
pipes; and conv3's forward on conv_fw_rs.h where its rows are 32+ pixels wide) against the kernels they replaced (conv_kyo.h's f32-input dX, conv_dwb16.h, conv_dw16.h's pair kernel; still in the library for
other geometries and selectable in the ablation build): the same minibatch through both, every gradient compared per variable.
The new bodies differ from the old in summation order, in the dX products (three bf16 pieces per operand, six products, instead of
f32 x f32) and in conv1 dW's scale (one 2^S per wave instead of per workgroup): agreement to a few f32 ulps of the gradient's size,
not bit identity.  (Parity with the float64 oracle is asserted by the suites that run on these kernels by default: test_gpu_fullsize.py,
test_gpu_fused_fullsize.py, test_gpu_naf.py, test_gpu_render_inputs.py.)"""
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
shape, B, fused = eval(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "fused"
agent, _ref, _ = make_pair(shape, B, True, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=21)
idxs = np.arange(B, dtype=np.int32)
if fused:
    agent.train_step(B, 1, idxs=idxs)                        # both networks' conv backward in paired launches
else:
    batch = agent.replay_memory.batch(idxs=idxs)
    agent.actor.train(batch); agent.critic.train(batch)      # one network per launch
np.save(sys.argv[4], np.concatenate([agent.actor.get_grads(), agent.critic.get_grads()]))
agent.close()
"""

# (CPP_CONV_FWRS: conv3's forward at rows of 32+ pixels -- the cfg5-geometry case -- back on the f32-input kernel as well)
OLD = {"CPP_CONV_DXRS": "0", "CPP_CONV_DWRS": "0", "CPP_CONV1_DWRS": "0", "CPP_CONV_FWRS": "0"}


def _grads(tmp_path, name, shape, B, mode, extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / (name + ".npy"))
    r = subprocess.run([sys.executable, "-c", _SNIPPET, repr(shape), str(B), mode, out], cwd=root,
                       env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-1500:]
    return np.load(out).astype(np.float64)


@pytest.mark.parametrize("shape,B,mode", [
    ((64, 64, 3, 2, 3), 256, "fused"),        # cfg3: conv1 dW wave-per-unit, conv2's and conv3's pairs on the row-streaming bodies
    ((64, 64, 3, 2, 3), 5, "fused"),          # odd batch: the last workgroup of a dX launch has an image without a wave's work
    ((64, 64, 3, 2, 3), 7, "unfused"),        # one network per launch: stand-alone launches, two bands per image for dX
    ((64, 64, 3, 1, 3), 6, "fused"),          # 9 channels: conv1 stays on conv_dw16.h, conv2 / conv3 on the new bodies
    ((128, 128, 3, 2, 5), 4, "fused"),        # cfg5's geometry: 64-wide conv2 rows (two columns per band, four tiles per row), 32-wide conv3 rows
], ids=["cfg3-B256-fused", "cfg3-B5-fused", "cfg3-B7-op-by-op", "64x64x9-B6-fused", "cfg5-geometry-B4-fused"])
def test_round_5_backward_kernels_agree_with_the_kernels_they_replaced(tmp_path, shape, B, mode):
    new = _grads(tmp_path, "new", shape, B, mode, {})
    old = _grads(tmp_path, "old", shape, B, mode, OLD)
    assert np.isfinite(new).all() and np.abs(new).max() > 0
    # per network half, relative to that half's largest gradient: conv1's gradients sum ~1e6 products each
    h = len(new) // 2
    for lo, hi, what in ((0, h, "first half (actor)"), (h, len(new), "second half (critic)")):
        d = np.abs(new[lo:hi] - old[lo:hi]).max()
        assert d <= 3e-6 * np.abs(old[lo:hi]).max(), (what, d, np.abs(old[lo:hi]).max())
    assert np.linalg.norm(new - old) <= 2e-6 * np.linalg.norm(old)


def test_exact_products_mode_runs_the_nine_product_instances_of_the_new_bodies(tmp_path):
    """cpp_ctx_set_precision(CPP_PRECISION_EXACT): all nine bf16 piece products in conv_dx_rs.h / conv_dw_rs.h (conv1 dW keeps conv_dw16.h's
    three-piece kernel).  dW: the old kernel issues the same nine exact products -- only the summation order differs; dX: nine exact
    products of the three-piece operands = the f32 kernel's products."""
    shape, B = (64, 64, 3, 2, 3), 8
    new = _grads(tmp_path, "new", shape, B, "fused", {"TEST_EXACT_PRODUCTS": "1"})
    old = _grads(tmp_path, "old", shape, B, "fused", dict(OLD, TEST_EXACT_PRODUCTS="1"))
    assert np.isfinite(new).all() and np.abs(new).max() > 0
    assert np.abs(new - old).max() <= 2e-6 * np.abs(old).max() and np.linalg.norm(new - old) <= 1e-6 * np.linalg.norm(old)


def test_the_banded_and_the_whole_image_walk_of_conv_dx_rs_give_the_same_bits(tmp_path):
    """conv_dx_rs.h walks an image in two bands of output rows when a launch would leave CUs without a workgroup; a band starts its walk
    up to UNR - 1 rows early (rows that only reach output rows it does not store).  The same sums in the same order: bit-identical."""
    shape, B = (64, 64, 3, 2, 3), 6
    a = _grads(tmp_path, "bands", shape, B, "unfused", {"CPP_DXRS_BANDS": "2"})
    b = _grads(tmp_path, "whole", shape, B, "unfused", {"CPP_DXRS_BANDS": "0"})
    assert np.array_equal(a, b)
