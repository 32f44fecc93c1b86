"""Parity on the inputs the reference actually trains on: RENDERS (bullet_cartpole.py:227-257) -- a cart and a pole on flat sky /
ground levels, R action-repeat frames per camera that differ by a few pixels, state_2 of a transition = state_1 of the next, and
optionally a camera that sees one colour only.  Uniform pixel noise (every other GPU test) has none of this: no near-constant
channel (whitening scales of 9 ... 1000 instead of 3.45, base_network.py:95-99), no large regions of EXACT pooling ties on positive
values (base_network.py:107), no repeated frames.  The frames come from synthetic_env.RasterCartpole, a small software rasteriser
with that structure, and enter the replay memory through add_episode as the reference's rollouts do (ddpg_cartpole.py:315-326).

Bars: north_star's 1e-5 on actions / Q / TD and 2e-5 per variable on pre-clip gradients wherever a float32 evaluation can meet them;
on inputs with NEARLY constant channels (scale ~ 1000 on values that do not cancel exactly) no float32 evaluation can -- there the
bar is relative to the float32 numpy evaluation of the same step (the rounding the reference's TF CPU kernels are entitled to):
err_device <= F32_FACTOR * err_f32_oracle + 1e-7."""
import numpy as np
import pytest

from oracle import ddpg_np as O
from tests.helpers import fused_step_against_f64_oracle, make_pair, fill_with_rendered_episodes, device_pool_codes

pytestmark = pytest.mark.gpu
CFG3 = (64, 64, 3, 2, 3)
F32_FACTOR = 1.5


@pytest.mark.parametrize("fill", ["render", "render-blind"])
def test_cfg3_B256_graph_replayed_fused_step_on_rendered_episodes(fill):
    """the release kernels (two f16 pieces / six bf16 products), the hipGraph replay, device-drawn rows: every bar of the noise-input
    test (tests/test_gpu_fused_fullsize.py) holds on renders, with and without a blind camera (three channels of zero variance)."""
    rep = fused_step_against_f64_oracle(CFG3, 256, rows=600, graph=True, seed=11, fill=fill, f32_twin=True)
    print("cfg3 B=256 fused graph step on %s inputs vs f64 oracle:" % fill, rep)
    if fill == "render-blind":
        assert rep["white_scale_max"] == 1000.0              # (the oracle's table: rsqrt(0 + 1e-6))
    else:
        assert rep["white_scale_max"] > 2.0 * rep["white_scale_min"]      # near-constant channels: a sky, a floor
    for k in ("actions", "dq_da", "q", "td", "pool1", "pool2", "pool3"):
        assert rep["err_" + k] <= F32_FACTOR * rep["f32_err_" + k] + 1e-6, (k, rep)      # no further from float64 than float32 numpy is


def test_cfg3_B256_fused_step_on_rendered_episodes_from_the_8_bit_store():
    rep = fused_step_against_f64_oracle(CFG3, 256, rows=600, graph=True, seed=12, fill="render", replay_store="u8", f32_twin=True)
    print("cfg3 B=256 (u8 store) on render inputs:", rep)


def test_cfg2_B256_fused_step_on_rendered_episodes():
    rep = fused_step_against_f64_oracle((64, 64, 3, 1, 3), 256, rows=600, graph=True, seed=13, fill="render", f32_twin=True)
    print("cfg2 B=256 on render inputs:", rep)


def test_reference_default_render_50x50x6_B128_on_rendered_episodes():
    rep = fused_step_against_f64_oracle((50, 50, 3, 1, 2), 128, rows=500, graph=True, seed=14, fill="render", f32_twin=True)
    print("50x50x6 B=128 on render inputs:", rep)


def test_cfg5_geometry_B512_fused_step_on_rendered_episodes_with_a_blind_camera():
    """BASELINE configs[4]'s geometry (128 x 128 x 30, B = 512): conv1 forward and dW run their FIVE-chunk instances there (a different
    code path: no interior loop, 254 registers), which every earlier test fed noise only.  Rendered episodes with a camera that sees
    one colour (three channels of zero variance: whitening table (0, 0)); every bar of the cfg3 render test."""
    rep = fused_step_against_f64_oracle((128, 128, 3, 2, 5), 512, rows=700, graph=True, seed=15, fill="render-blind", f32_twin=True)
    print("cfg5 geometry B=512 fused graph step on render-blind inputs vs f64 oracle:", rep)
    assert rep["white_scale_max"] == 1000.0
    for k in ("actions", "dq_da", "q", "td", "pool1", "pool2", "pool3"):
        assert rep["err_" + k] <= F32_FACTOR * rep["f32_err_" + k] + 1e-6, (k, rep)


@pytest.mark.parametrize("fill", ["render", "render-blind"])
def test_cfg4_B256_naf_step_on_rendered_episodes(fill):
    from tests.test_gpu_naf import naf_fused_step_against_f64_oracle
    naf_fused_step_against_f64_oracle(CFG3, 256, True, fill=fill)


NEAR_CONSTANT_FACTOR = 1.5


def test_nearly_constant_channels_run_on_the_f32_input_kernels_and_meet_the_float32_bars():
    """a blind camera with a rare single off-colour pixel: three channels with scale ~ 990 whose whitened values do NOT cancel
    exactly (a glint whitens to ~400, the rest to -4e-4).  Conv outputs reach ~100 and Q ~6; float32 numpy itself is 1e-4 away
    from float64 here, so the bars are relative to it.  The f16-pipe conv1 kernels multiply the RAW pixel and cancel inside the
    MFMA: 2.6e-4 from float64 on conv1 outputs where float32 numpy is 1.3e-4 and the f32-input kernel, which whitens each element
    first as base_network.py:95-99 does, 0.7e-4 (rounds 1-4 shipped that, behind a factor-3 bar and without looking at a gradient).
    Round 5: the step publishes the largest whitening scale it saw and a LATER step runs conv1 (forward and dW) on the f32-input
    kernels (cpp_ctx_set_route_threshold, default 100; round 6: the next one if the stream was synchronised in between, the one after it
    otherwise -- never a matter of timing) -- here the captured first step sees 990 (and synchronises), the measured second one is routed.
    Every bar of the ordinary render test then holds at the ordinary factor, gradients included."""
    from cartpoleplusplus_amd import _lib
    # (atol: the helper's absolute bars are north_star's 1e-5, which float32 numpy itself misses here by 2.4x on actions and 12x on TD; the
    # asserts that matter are the ones below, relative to the float32 evaluation -- and the helper's per-variable gradient bars, which with
    # f32_twin are 1.5 x the float32 evaluation's own distance from float64)
    rep = fused_step_against_f64_oracle(CFG3, 256, rows=600, graph=True, seed=11, fill="render-glint", f32_twin=True, flip_tol=1e-4, atol=3e-4,
                                        param_rel=1e-5)      # (parameters after the step: 2e-6 elsewhere; conv3/biases sits at 2.8e-6 here)
    print("cfg3 B=256 on render-glint inputs (routed to the f32-input conv1 kernels):", rep)
    assert 300.0 < rep["white_scale_max"] < 1000.0
    routed, seen = _lib.default_context().route()
    assert routed and 300.0 < seen < 1000.0, (routed, seen)
    for k in ("actions", "dq_da", "q", "td", "pool1", "pool2", "pool3"):
        assert rep["err_" + k] <= NEAR_CONSTANT_FACTOR * rep["f32_err_" + k] + 1e-7, (k, rep)


def test_the_route_follows_the_whitening_scale_and_can_be_switched_off():
    """the switch itself: noise inputs (scale 3.4) never move it; glint renders do from the second step on; a memory of noise brings
    conv1 back to the f16 pipes once the scale has fallen under half the threshold; threshold 0 pins the f16 pipes."""
    from cartpoleplusplus_amd import _lib
    ctx = _lib.default_context()
    agent, _ref, _ = make_pair(CFG3, 64, True, seed=3, replay_size=700)
    try:
        agent.replay_memory.fill_synthetic(200, seed=5)
        for _ in range(3):
            agent.train_step(64, 2)
        ctx.sync()
        routed, seen = ctx.route()
        assert not routed and 3.0 < seen < 4.0, (routed, seen)
    finally:
        agent.close()
    agent, _ref, _ = make_pair(CFG3, 64, True, seed=3, replay_size=700)
    try:
        fill_with_rendered_episodes(agent, CFG3, 300, seed=4, blind_camera=True, glint=0.02)
        agent.train_step(64, 2); ctx.sync()
        assert not ctx.route()[0] and ctx.route()[1] > 300.0          # seen, not yet acted on
        agent.train_step(64, 2); ctx.sync()
        assert ctx.route()[0]
        p_routed = agent.actor.get_params()
        assert np.isfinite(p_routed).all()
        ctx.set_route_threshold(0.0)                                   # off: back on the f16 pipes at once, whatever the scale
        assert not ctx.route()[0]
        agent.train_step(64, 2); ctx.sync()
        assert not ctx.route()[0]
        ctx.set_route_threshold(100.0)
        agent.train_step(64, 2); ctx.sync()                            # (sees the scale again ...)
        agent.train_step(64, 2); ctx.sync()
        assert ctx.route()[0]                                          # (... and is routed again)
    finally:
        agent.close()
    agent, _ref, _ = make_pair(CFG3, 64, True, seed=3, replay_size=700)
    try:
        agent.replay_memory.fill_synthetic(200, seed=5)
        assert ctx.route()[0]                                          # (the context remembers: same GPU, next agent)
        agent.train_step(64, 2); ctx.sync()
        agent.train_step(64, 2); ctx.sync()
        assert not ctx.route()[0] and ctx.route()[1] < 4.0
    finally:
        agent.close()


def test_the_route_is_a_function_of_the_calls_not_of_host_timing():
    """Round 5 read the published scale "without waiting for anything": which later step first ran on the f32-input kernels depended on
    when an earlier graph's closing kernel happened to land, so two runs of one seed could differ.  Round 6: call k decides from call
    k - 2's scale (k - 1's if the stream was synchronised in between), behind an event -- a function of the program's order alone.  The
    same glint episodes trained twice, once with the host racing ahead of the GPU and once with the host asleep between the calls (every
    step long finished before the next is entered): identical parameters, bit for bit, and both runs routed."""
    import time
    from cartpoleplusplus_amd import _lib
    ctx = _lib.default_context()

    def run(pause):
        ctx.sync(); ctx.set_route_threshold(0.0); ctx.set_route_threshold(100.0)
        agent, _ref, _ = make_pair(CFG3, 64, True, seed=3, replay_size=700)
        try:
            fill_with_rendered_episodes(agent, CFG3, 300, seed=4, blind_camera=True, glint=0.02)
            for _ in range(8):
                agent.train_step(64, 2)
                if pause:
                    time.sleep(pause)
            ctx.sync()
            return (np.concatenate([agent.actor.get_params(), agent.critic.get_params(), agent.target_actor.get_params()]), ctx.route())
        finally:
            agent.close()

    p_fast, r_fast = run(0.0)
    p_slow, r_slow = run(0.02)
    assert r_fast[0] and r_slow[0] and r_fast[1] > 300.0, (r_fast, r_slow)
    assert np.isfinite(p_fast).all()
    assert np.array_equal(p_fast, p_slow), np.abs(p_fast - p_slow).max()


def _flat_images(B, shape, rng):
    """B states whose every channel is flat inside each image but differs between images: conv outputs are translation
    invariant away from the border, so EVERY interior pooling window of every layer is an exact four-way tie."""
    H, W = shape[0], shape[1]
    C = int(np.prod(shape[2:]))
    codes = rng.integers(20, 236, (B, 1, 1, C))
    x = np.broadcast_to(codes, (B, H, W, C)).astype(np.float64) / 255.0
    return x.astype(np.float16).reshape((B,) + tuple(shape))


def test_exact_positive_pooling_ties_route_to_the_first_maximum():
    """base_network.py:107 (slim.max_pool2d) on windows whose four pre-activations are EQUAL and positive: TF's max-pool gradient goes
    to the first maximum in window order (y, x) -- np.argmax in the oracle, strict `>` in the kernels' pool epilogues.  Forward: the
    device's arg-max code is 0 in every such window of every layer.  Backward: the device's gradients equal the oracle's with ITS
    OWN routing (no override), and are measurably different from an oracle that routes ties to the LAST maximum -- so the comparison
    is sensitive to the rule."""
    shape, B = (32, 32, 3, 2, 3), 16
    agent, ref, (aspec, cspec) = make_pair(shape, B, True, seed=8)
    rng = np.random.default_rng(3)

    class HB(object):
        pass
    hb = HB()
    s1 = _flat_images(B, shape, rng)
    # half of each image gets a second flat level: ties everywhere except along the seam, and gradient routes that matter there
    s1[:, :, shape[1] // 2:] = _flat_images(B, shape, rng)[:, :, shape[1] // 2:]
    t = O.synthetic_batch(rng, B, shape, 2, True)
    hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = s1, t[1], t[2], t[3], s1[::-1].copy()
    try:
        agent.critic.train(hb)
        got = agent.critic.get_grads()
        codes = device_pool_codes(agent.critic, B)
    finally:
        agent.close()
    batch = (hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2)
    cg = ref.critic_gradients(batch)
    cache = cg["cache_critic"]
    n_ties = 0
    for name, _k, _co in O.CONV_DEFS:
        _x, pooled, amax, _h, _w = cache[name]
        tie = (cache[name + ":margin"] == 0.0) & (pooled > 0)          # exact ties on positive values, as the float64 oracle sees them
        # (a window the oracle sees as an exact tie is one whose four inputs are translates of each other: equal on the device too)
        if name != "conv3":      # (32 x 32 inputs: conv3 sees 8 x 8, every window within reach of a border or the seam)
            assert tie.sum() > 50, (name, int(tie.sum()))
        # np.argmax = the FIRST of the equal maxima: code 0 in a four-way tie (the flat interior), 2 where only the window's lower
        # row ties at the top (an image's first rows see the SAME padding), 1 / 0 likewise at the left edge
        dev = codes[name].reshape(amax.shape)
        assert (dev[tie] == amax[tie]).all(), "%s: %d exact positive ties not routed to the first maximum" % (
            name, int((dev[tie] != amax[tie]).sum()))
        assert (name == "conv3" or (amax[tie] == 0).sum() > 50) and (dev[tie] == 0).sum() == (amax[tie] == 0).sum()
        n_ties += int(tie.sum())
    from tests.helpers import per_var_report
    rel_first = max(r[2] for r in per_var_report(cspec, got, cg["grads"]))
    # the same oracle with ties routed to the LAST maximum
    last = {}
    for name, _k, _co in O.CONV_DEFS:
        _x, pooled, amax, _h, _w = cache[name]
        tie = cache[name + ":margin"] == 0.0
        last[name] = np.where(tie, 3, amax).astype(np.uint8)
    ref.critic.amax_override = last
    try:
        rel_last = max(r[2] for r in per_var_report(cspec, got, ref.critic_gradients(batch)["grads"]))
    finally:
        ref.critic.amax_override = None
    print("exact positive ties: %d windows; gradient rel err vs first-max oracle %.2e, vs last-max oracle %.2e" % (n_ties, rel_first, rel_last))
    assert rel_first < 2e-5 and rel_last > 50 * rel_first, (rel_first, rel_last)
