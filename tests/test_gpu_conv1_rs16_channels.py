"""Round 6: conv1 forward's row-streaming kernel (csrc/conv_rs16.h) at every channel count it is instantiated for -- 3, 6, 9 (cfg2),
12 and 18 (cfg3) channels, 64 pixels wide -- against the (ky, o)-ring kernel it replaces there (csrc/conv_k16.h, still in the library for the
other geometries; `CPP_CONV_RS16_CH=0` in the ablation build puts every count but 18 back on it).

Both kernels issue the same products (raw f16 pixel x two f16 pieces of W s 2^S, the chunks' ones slots) and add them in the same order
(chunk by chunk, ky inside a piece, f32 accumulators), so every pooled output whose 5x5 windows do not touch the image's left or right
border must hold the SAME BITS; at the border columns the data-independent pivot remainder enters the accumulator at its restart instead
of its end: a few f32 ulps of the pre-activation.  An odd channel count (9) reads its odd pixels from a second, displaced
copy of the staged row -- an arrangement, not arithmetic.  (Parity with the float64 oracle on these kernels: test_gpu_fused_fullsize.py,
test_gpu_render_inputs.py, test_gpu_fuzz.py run on them by default.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_SNIPPET = r"""
import sys
import numpy as np
from tests.helpers import make_pair, device_pool_codes
shape, B = eval(sys.argv[1]), int(sys.argv[2])
agent, _ref, _ = make_pair(shape, B, True, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=33)
idxs = np.arange(B, dtype=np.int32)
agent.train_step(B, 1, idxs=idxs)
out = {}
for name, net in (("actor", agent.actor), ("critic", agent.critic), ("tactor", agent.target_actor), ("tcritic", agent.target_critic)):
    if name in ("actor", "critic"):                      # (the target networks' f32 pool1 and codes are never written)
        out[name + "_pool1"] = net.pool1.eval(B)
        out[name + "_code1"] = device_pool_codes(net, B)["conv1"]
    out[name + "_pool3"] = net.pool3.eval(B)
out["grads"] = np.concatenate([agent.actor.get_grads(), agent.critic.get_grads()])
np.savez(sys.argv[3], **out)
agent.close()
"""


def _run(tmp_path, name, shape, B, extra):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / (name + ".npz"))
    r = subprocess.run([sys.executable, "-c", _SNIPPET, repr(shape), str(B), out], cwd=root,
                       env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-1500:]
    return dict(np.load(out))


@pytest.mark.parametrize("shape,B", [
    ((64, 64, 3, 1, 3), 256),        # cfg2: 9 channels, two chunks, the displaced copy
    ((64, 64, 3, 1, 3), 5),          # odd batch: a workgroup with one image
    ((64, 64, 3, 1, 1), 6),          # 3 channels: one chunk (15 of its 30 k), odd
    ((64, 64, 3, 1, 2), 6),          # 6 channels: one full chunk
    ((64, 64, 3, 2, 2), 6),          # 12 channels: two full chunks
    ((32, 64, 3, 1, 3), 4),          # a shorter image (H = 32): the general blocks at both ends meet
], ids=["cfg2-B256", "9ch-B5", "3ch", "6ch", "12ch", "9ch-32x64"])
def test_row_streaming_conv1_at_other_channel_counts_equals_the_ring_kernel_away_from_the_borders(tmp_path, shape, B):
    new = _run(tmp_path, "new", shape, B, {})
    old = _run(tmp_path, "old", shape, B, {"CPP_CONV_RS16_CH": "0"})
    for net in ("actor", "critic"):
        pn, po = new[net + "_pool1"], old[net + "_pool1"]
        assert np.isfinite(pn).all() and np.abs(pn).max() > 0
        inner = (slice(None), slice(None), slice(1, -1))
        assert np.array_equal(pn[inner], po[inner]), (net, np.abs(pn[inner] - po[inner]).max())
        assert np.array_equal(new[net + "_code1"][inner], old[net + "_code1"][inner]), net
        # border columns: the pivot remainder (<= 2^-12 of a term) enters at the restart: a few ulps of pre-activations of size ~10
        d = np.abs(pn - po).max()
        assert d <= 2e-5 * max(1.0, np.abs(po).max()), (net, d, np.abs(po).max())
    # the target networks read conv1 through the bf16 planes only: compare what conv3 made of them
    for net in ("actor", "critic", "tactor", "tcritic"):
        d = np.abs(new[net + "_pool3"] - old[net + "_pool3"]).max()
        assert d <= 2e-5 * max(1.0, np.abs(old[net + "_pool3"]).max()), (net, d)
    g, go = new["grads"].astype(np.float64), old["grads"].astype(np.float64)
    assert np.linalg.norm(g - go) <= 2e-5 * np.linalg.norm(go)


@pytest.mark.parametrize("B", [256, 5])
def test_wave_per_unit_conv1_dw_at_9_channels_agrees_with_the_kernel_it_does_not_replace(tmp_path, B):
    """conv_dw16_rs.h (one wave per (network, 32-pixel column) unit) is templated on the channel count; its 9-channel instance (4 x 4
    accumulator tiles per wave, raw dwords stored half by half at a 12-half pixel pitch) measured 43.0 us against conv_dw16.h's 40.4 at
    cfg2 and stays an opt-in of the ablation build (`CPP_CONV1_DWRS_CH=1`).  Same forward pass, same products (raw pixel x two f16
    pieces of dY 2^S), a different summation order and one 2^S per wave instead of per workgroup: a few f32 ulps of the gradient's
    size, as for 18 channels (tests/test_gpu_backward_rs.py)."""
    shape = (64, 64, 3, 1, 3)
    new = _run(tmp_path, "new", shape, B, {"CPP_CONV1_DWRS_CH": "1"})["grads"].astype(np.float64)
    old = _run(tmp_path, "old", shape, B, {})["grads"].astype(np.float64)
    assert np.isfinite(new).all() and np.abs(new).max() > 0 and not np.array_equal(new, old)
    h = len(new) // 2
    for lo, hi, what in ((0, h, "actor"), (h, len(new), "critic")):
        d = np.abs(new[lo:hi] - old[lo:hi]).max()
        assert d <= 3e-6 * np.abs(old[lo:hi]).max(), (what, d, np.abs(old[lo:hi]).max())
    assert np.linalg.norm(new - old) <= 2e-6 * np.linalg.norm(old)


@pytest.mark.parametrize("H,cams,reps,B,graph,fill", [
    (18, 1, 3, 10, True, "noise"),        # the shortest images: the row walk's first and last general blocks overlap
    (34, 1, 1, 2, False, "render"),       # 3 channels on rendered frames
    (50, 2, 2, 7, True, "noise"),         # 12 channels, a height that is no multiple of the block
    (66, 1, 3, 5, True, "render"),        # 9 channels, taller than wide
    (96, 1, 2, 3, False, "noise"),        # 6 channels
], ids=["18x64x9", "34x64x3-render", "50x64x12", "66x64x9-render", "96x64x6"])
def test_channel_instances_at_other_heights_against_the_f64_oracle(H, cams, reps, B, graph, fill):
    """the row-streaming conv1 kernel's channel instances on 64-wide images of other heights, through the whole fused step against
    oracle.DDPG(float64) at the suite's ordinary bars (a slice of `profiles/diag/rs16_geometry_parity.py`, which ran 150 such draws)."""
    from tests.helpers import fused_step_against_f64_oracle
    rep = fused_step_against_f64_oracle((H, 64, 3, cams, reps), B, rows=60, graph=graph, seed=17, fill=fill)
    assert rep["err_q"] <= 1e-5, rep


@pytest.mark.parametrize("shape,B", [((64, 64, 3, 2, 3), 96), ((64, 64, 3, 1, 3), 31), ((50, 64, 3, 2, 2), 64), ((36, 64, 3, 1, 1), 40)],
                         ids=["18ch-B96", "9ch-B31", "50x64x12-B64", "36x64x3-B40"])
def test_two_bands_of_rows_per_image_are_an_arrangement_not_arithmetic(tmp_path, shape, B):
    """A launch of conv_fwd_rs16_kernel that would put at most one workgroup on a CU (NAF's two trunks at B = 256; here four networks at
    small batches) walks every image as TWO bands of output rows (round 6; `CPP_CONV_BANDS=0` in the ablation build: whole images).  A
    band's walk starts two input rows above its first output row and its unstored first steps are the only difference: every pooled
    value, code and bf16 plane -- and with them conv3's output and both gradient lists -- must hold the SAME BITS."""
    new = _run(tmp_path, "bands", shape, B, {})
    old = _run(tmp_path, "whole", shape, B, {"CPP_CONV_BANDS": "0"})
    assert sorted(new) == sorted(old)
    for k in new:
        assert np.isfinite(new[k]).all() and np.array_equal(new[k], old[k]), (k, np.abs(new[k].astype(np.float64) - old[k]).max())
    assert np.abs(new["actor_pool1"]).max() > 0 and np.abs(new["grads"]).max() > 0
