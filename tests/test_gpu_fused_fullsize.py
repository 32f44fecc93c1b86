"""The fused inner step at the sizes BASELINE.json's metric is quoted on, against the float64 oracle at north_star's bar
(actions / Q / TD within 1e-5; pre-clip gradients of BOTH networks per variable at 2e-5) -- the path bench.py times: default
kernels (f16x2 conv1 reading the replay store through the sampled slots, bf16x6 conv2, fused heads, paired launches), the
hipGraph replay, device-drawn rows.  Match: ddpg_cartpole.py:329-337, base_network.py:95-127."""
import numpy as np
import pytest

from tests.helpers import fused_step_against_f64_oracle

pytestmark = pytest.mark.gpu


def test_cfg3_B256_graph_replayed_fused_step_against_f64_oracle():
    rep = fused_step_against_f64_oracle((64, 64, 3, 2, 3), 256, rows=2500, graph=True)
    print("cfg3 B=256 fused graph step vs f64 oracle:", rep)


def test_cfg3_B256_fused_step_with_caller_rows_against_f64_oracle():
    rep = fused_step_against_f64_oracle((64, 64, 3, 2, 3), 256, rows=2500, graph=False, seed=3)
    print("cfg3 B=256 fused eager step vs f64 oracle:", rep)


def test_cfg2_B256_graph_replayed_fused_step_against_f64_oracle():
    rep = fused_step_against_f64_oracle((64, 64, 3, 1, 3), 256, rows=2500, graph=True, seed=1)
    print("cfg2 B=256 fused graph step vs f64 oracle:", rep)


def test_cfg3_B256_fused_step_from_the_8_bit_store_against_f64_oracle():
    rep = fused_step_against_f64_oracle((64, 64, 3, 2, 3), 256, rows=2500, graph=True, seed=2, replay_store="u8")
    print("cfg3 B=256 (u8 store) fused graph step vs f64 oracle:", rep)


def test_cfg5_B512_graph_replayed_fused_step_against_f64_oracle():
    """BASELINE configs[4] at its own batch size: 128x128x30, B = 512 (the oracle pass is ~100 GFLOP of float64 numpy)."""
    rep = fused_step_against_f64_oracle((128, 128, 3, 2, 5), 512, rows=1500, graph=True, seed=4)
    print("cfg5 B=512 fused graph step vs f64 oracle:", rep)


def test_cfg5_geometry_under_exact_products_against_f64_oracle():
    """cfg5's 128x128x30 geometry with cpp_ctx_set_precision(EXACT): the three-piece instances of the 30-channel kernels -- the ring
    kernel's five-chunk forward, conv_dw16.h's dW with its accumulator tiles divided 6 x 2 over the waves (Dw16Geom::SPLIT, round 6) --
    and all nine bf16 products at 64-wide conv2 rows, through the fused step (a subprocess: the mode is a property of the context).
    (Seed: a draw with |Q| <= 4.2.  At this geometry the absolute 1e-5 bar on TD sits at float32's own distance from float64 -- seed 6
    has |Q| = 14.7 and float32 NUMPY is 1.1e-5 - 1.3e-5 away there, the device 0.9e-5 (fast) / 1.1e-5 (exact): profiles/diag/exact30_probe.py.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    snippet = ("from tests.helpers import fused_step_against_f64_oracle\n"
               "rep = fused_step_against_f64_oracle((128, 128, 3, 2, 5), 96, rows=300, graph=True, seed=8)\n"
               "print('EXACT30', rep['err_q'], rep['rel_actor_grads'], rep['rel_critic_grads'])\n")
    r = subprocess.run([sys.executable, "-c", snippet], cwd=root, env=dict(os.environ, TEST_EXACT_PRODUCTS="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and "EXACT30" in out, out[-2000:]


@pytest.mark.parametrize("shape", [(50, 50, 3, 1, 2), (50, 50, 3, 2, 3)], ids=["50x50x6-all-defaults", "50x50x18"])
def test_reference_default_render_B128_graph_replayed_fused_step_against_f64_oracle(shape):
    """the reference's OWN defaults: 50 x 50 render (bullet_cartpole.py:33-36), one camera x two action repeats = 6 channels
    (:21,:25), batch 128 (ddpg_cartpole.py:29) -- and the same render with the 18 channels of the benchmark: rows that are not
    16-byte multiples and odd pooled sizes (25 -> 12 -> 6: 'VALID' drops the last row and column)."""
    rep = fused_step_against_f64_oracle(shape, 128, rows=1500, graph=True, seed=6)
    print("%dx%dx%d B=128 fused graph step vs f64 oracle:" % (shape[0], shape[1], int(np.prod(shape[2:]))), rep)


@pytest.mark.parametrize("B", [255, 17, 1])
def test_cfg3_geometry_at_odd_batch_sizes_against_f64_oracle(B):
    """batch sizes that do not fill the kernels' image groups (two / four images per workgroup; conv3 can only ride in conv2's
    launch with an even count): the same graph-replayed step, the same bar."""
    rep = fused_step_against_f64_oracle((64, 64, 3, 2, 3), B, rows=1200, graph=True, seed=9)
    print("cfg3 geometry, B=%d:" % B, rep)


@pytest.mark.parametrize("cams,reps", [(1, 1), (1, 2), (1, 3), (2, 2), (1, 4), (1, 5), (2, 3), (2, 4), (2, 5)],
                         ids=lambda v: str(v))
def test_every_channel_count_the_flags_can_produce(cams, reps):
    """--num-cameras 1..2 x --action-repeats 1..5 (bullet_cartpole.py:21-26, :121-123): 3, 6, 9, 12, 15, 18, 24, 30 channels, at a
    render the generic and the f16-pipe kernels both see (40 x 40, odd pooled sizes further down)."""
    rep = fused_step_against_f64_oracle((40, 40, 3, cams, reps), 6, rows=150, graph=True, seed=cams + 3 * reps)
    print("%d channels:" % (3 * cams * reps), rep)


_RIDER_SNIPPET = r"""
import hashlib, sys
import numpy as np
from tests.helpers import make_pair
from tests.test_gpu_naf import make_naf
shape, B = (64, 64, 3, 2, 3), 64
agent, _ref, _ = make_pair(shape, B, True, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=11)
for _ in range(3):
    agent.train_step(B, 3)
agent.actor.ctx.sync()
h = hashlib.sha256()
for n in (agent.actor, agent.critic, agent.target_actor, agent.target_critic):
    h.update(n.get_params().tobytes())
print("DIGEST ddpg", h.hexdigest())
agent.close()
agent, _ref, _ = make_naf(shape, B, True, "Momentum", {"learning_rate": 0.01, "momentum": 0.9}, seed=4, replay_size=4 * B)
agent.replay_memory.fill_synthetic(3 * B, seed=12)
for _ in range(3):
    agent.train_step(B, 3)
h = hashlib.sha256()
for n in (agent.value_net, agent.naf.mu_net, agent.naf.l_net, agent.target_value_net):
    h.update(n.get_params().tobytes())
h.update(np.asarray(agent.naf.get_optimiser_state()["m"]).tobytes())
print("DIGEST naf", h.hexdigest())
"""


def test_the_conv1_image_rider_is_an_arrangement_not_arithmetic():
    """conv1's operand images (conv_rs16.h) are built by the optimiser's launch for the NEXT minibatch -- the rider restates the SGD /
    Momentum update of conv1's own parameters (optim.hip; rt_ddpg.cpp apply, rt_naf.cpp naf_apply) -- or by a launch of their own in
    front of the forward kernel (CPP_RIDE_IMAGE=0, ablation build).  Nine minibatches in three calls (a target update between the calls,
    so both the rider's and the stand-alone launch's images are consumed): every parameter, target parameter and Momentum slot is
    bit-identical between the two arrangements, DDPG (SGD) and NAF (Momentum)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, extra in (("rider", {}), ("own-launch", {"CPP_RIDE_IMAGE": "0"})):
        r = subprocess.run([sys.executable, "-c", _RIDER_SNIPPET], cwd=root, env=dict(os.environ, CARTPOLEPP_ABLATION="1", **extra),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        out = r.stdout.decode()
        assert r.returncode == 0, out[-1500:]
        got[name] = [l for l in out.splitlines() if l.startswith("DIGEST")]
        assert len(got[name]) == 2, out[-1500:]
    assert got["rider"] == got["own-launch"], got
