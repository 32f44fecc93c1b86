"""CPU tests of host-side logic that needs no GPU: flags, exploration noise, helper functions."""
import numpy as np

from cartpoleplusplus_amd import util


def test_flag_defaults_match_the_reference():
    from cartpoleplusplus_amd import ddpg_cartpole as D
    o = D.default_opts()
    # ddpg_cartpole.py:18-57, util.py:10-20
    assert (o.batch_size, o.batches_per_step, o.target_update_rate) == (128, 5, 0.0001)
    assert (o.actor_learning_rate, o.critic_learning_rate, o.discount) == (0.001, 0.01, 0.99)
    assert (o.replay_memory_size, o.replay_memory_burn_in) == (22000, 1000)
    assert (o.action_noise_theta, o.action_noise_sigma) == (0.01, 0.05)
    assert o.gradient_clip == 5 and o.actor_hidden_layers == "100,100,50"
    assert util.gradient_clip_value(o) == 5.0


def test_ou_noise_reproduces_the_clip_quirk():
    np.random.seed(0)
    n = util.OrnsteinUhlenbeckNoise(2, theta=0.5, sigma=10.0)
    xs = np.array([n.sample() for _ in range(300)])
    assert xs.max() <= 1.5 and xs.min() < -1.5      # util.py:155: only the upper bound is enforced
    from oracle.ddpg_np import OUNoise
    np.random.seed(7); a = util.OrnsteinUhlenbeckNoise(3, 0.01, 0.2)
    b = OUNoise(3, 0.01, 0.2, rng=np.random.RandomState(7))
    for _ in range(20):
        assert np.array_equal(a.sample(), b.sample())


def test_collapsed_successive_ranges():
    assert util.collapsed_successive_ranges([2, 3, 4, 5, 13, 14, 15]) == "2-5, 13-15"
    assert util.collapsed_successive_ranges([7]) == "7-7"


def test_synthetic_env_shapes():
    from cartpoleplusplus_amd import ddpg_cartpole as D
    from cartpoleplusplus_amd.synthetic_env import SyntheticCartpole
    o = D.default_opts(use_raw_pixels=True, render_height=64, render_width=64, num_cameras=2, action_repeats=3)
    env = SyntheticCartpole(o)
    s = env.reset()
    assert s.shape == (64, 64, 3, 2, 3) and s.dtype == np.float32
    assert np.array_equal(s, s.astype(np.float16).astype(np.float32))   # f16(k/255) values
    s2, r, done, _ = env.step(np.zeros((1, 2)))
    assert s2.shape == s.shape and r == 1.0


def test_bench_flop_accounting_matches_the_survey():
    """conv FLOPs per step = 2 B (4 F + 2 Bk) with F / Bk the unpadded per-image MACs -- cfg2 39.74, cfg3 68.05,
    cfg5 846.4 GFLOP (the numerators of every roofline figure bench.py prints)."""
    import bench
    want = {"cfg2": (12006400, 14796800, 39.74), "cfg3": (21222400, 24012800, 68.05), "cfg5": (134041600, 145203200, 846.4)}
    for wl, (F_, Bk_, gf) in want.items():
        shape, B, kind = bench.WORKLOADS[wl]
        F, Bk, per_layer = bench.conv_macs(shape)
        assert (F, Bk) == (F_, Bk_) and kind == "ddpg"
        assert abs(2.0 * B * (4 * F + 2 * Bk) / 1e9 - gf) < 0.05 * max(1.0, gf / 100)
    F50 = bench.conv_macs((50, 50, 3, 1, 2))[0]                          # the default render: 5.44 MMAC per image
    assert abs(F50 / 1e6 - 5.44) < 0.01
    assert bench.PIPES["f16x3"][0] == bench.PEAK_F16_MFMA_TFLOPS / 3.0 and bench.PIPES["bf16x9"][0] == bench.PEAK_F16_MFMA_TFLOPS / 9.0


def test_host_gradient_helpers_follow_util_py():
    """util.py:33-58 on host arrays: l2_norm, standardise, clip_and_debug_gradients (tf.clip_by_global_norm's rule, None skipped)."""
    from cartpoleplusplus_amd import util

    class Opts(object):
        gradient_clip, print_gradients = 5.0, False
    g1, g2 = np.full((3, 4), 2.0, np.float32), np.full(5, -3.0, np.float32)
    norm = np.sqrt(12 * 4.0 + 5 * 9.0)
    assert abs(util.l2_norm(g1) - np.sqrt(48.0)) < 1e-12
    out = util.clip_and_debug_gradients([(g1, "a"), (None, "b"), (g2, "c")], Opts)
    assert out[1][0] is None and [v for _g, v in out] == ["a", "b", "c"]
    np.testing.assert_allclose(out[0][0], g1 * 5.0 / norm, rtol=1e-6)
    np.testing.assert_allclose(out[2][0], g2 * 5.0 / norm, rtol=1e-6)
    Opts.gradient_clip = 100.0                                        # norm below the clip: unchanged
    np.testing.assert_allclose(util.clip_and_debug_gradients([(g1, "a")], Opts)[0][0], g1)
    Opts.gradient_clip = None
    assert util.clip_and_debug_gradients([(g1, "a")], Opts)[0][0] is g1
    z = util.standardise(np.arange(10.0))
    assert abs(z.mean()) < 1e-12 and abs((z ** 2).mean() - 1.0) < 1e-12
