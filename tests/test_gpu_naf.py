"""GPU parity tests of the NAF path (naf_cartpole.py) against oracle/naf_np.py, through the package's
public surface (C ABI underneath).  Same tolerances as the DDPG tests."""
import json

import numpy as np
import pytest

from oracle import ddpg_np as O
from oracle import naf_np as N
from oracle.replay_np import OracleReplayMemory
from tests.helpers import FakeEnv, assert_flat_close

pytestmark = pytest.mark.gpu
ATOL = 1e-5


class HB(object):
    def __init__(self, t):
        self.state_1, self.action, self.reward, self.terminal_mask, self.state_2 = t


def make_naf(shape, B, share, optimiser="GradientDescent", optimiser_args=None, seed=0, replay_size=64, clip=5.0,
             use_batch_norm=False, use_dropout=False):
    from cartpoleplusplus_amd import naf_cartpole as F
    pixel = len(shape) == 5
    kw = dict(batch_size=B, replay_memory_size=replay_size, share_input_state_representation=share,
              optimiser=optimiser, optimiser_args=json.dumps(optimiser_args or {"learning_rate": 0.01}),
              gradient_clip=clip, use_batch_norm=use_batch_norm, use_dropout=use_dropout)
    if pixel:
        kw.update(use_raw_pixels=True, render_height=shape[0], render_width=shape[1], num_cameras=shape[3],
                  action_repeats=shape[4])
    else:
        kw.update(use_raw_pixels=False, action_repeats=shape[0])
    F.set_opts(F.default_opts(**kw))
    agent = F.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
    agent.initialise_variables(seed=seed)
    rng = np.random.default_rng(seed + 5)
    for net in (agent.value_net, agent.naf.mu_net, agent.naf.l_net):
        p = net.get_params()
        net.set_params(p + rng.normal(0, 0.05, p.shape).astype(np.float32))
    agent.post_var_init_setup()
    p = agent.target_value_net.get_params()
    agent.target_value_net.set_params(p + rng.normal(0, 0.01, p.shape).astype(np.float32))
    skw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:])), batch_norm=use_batch_norm) if pixel else \
        dict(pixel=False, state_elems=int(np.prod(shape)))
    skw["dropout"] = use_dropout
    vspec = N.HeadSpec(1, "linear", [100, 50], **skw)
    if share:
        mspec = N.HeadSpec(2, "tanh", [], False, state_elems=50, head_only=True)
        lspec = N.HeadSpec(3, "linear", [], False, state_elems=50, head_only=True)
    else:
        mspec, lspec = N.HeadSpec(2, "tanh", [100, 50], **skw), N.HeadSpec(3, "linear", [100, 50], **skw)
    ref = N.NAF(vspec, mspec, lspec, agent.value_net.get_params(), agent.naf.mu_net.get_params(),
                agent.naf.l_net.get_params(), share, 2, np.float64, gradient_clip=clip,
                optimiser=N.make_optimiser(optimiser, optimiser_args or {"learning_rate": 0.01}))
    ref.target_value = O.Net(vspec, agent.target_value_net.get_params(), np.float64)
    return agent, ref, (vspec, mspec, lspec)


def params_of(agent):
    return np.concatenate([agent.value_net.get_params(), agent.naf.mu_net.get_params(), agent.naf.l_net.get_params()])


class CatSpec(object):
    def __init__(self, specs):
        self.specs = specs

    def layout(self):
        out = []
        for tag, sp in zip(("value/", "mu/", "l/"), self.specs):
            out += [(tag + n, s) for n, s in sp.layout()]
        return out


CASES = [
    pytest.param((8, 8, 3, 1, 2), 4, True, id="8x8x6-share"),
    pytest.param((12, 10, 3, 1, 3), 3, False, id="12x10x9-own-trunks"),
    pytest.param((64, 64, 3, 2, 3), 2, True, id="64x64x18-share-cfg4"),
    pytest.param((2, 2, 7), 6, True, id="lowdim-share"),
    pytest.param((2, 2, 7), 6, False, id="lowdim-own"),
    pytest.param((2, 2, 7), 300, True, id="lowdim-share-B300-two-workgroups-of-the-heads-kernel"),
]


@pytest.mark.parametrize("shape,B,share", CASES)
def test_naf_forward_gradients_and_sgd_step(shape, B, share):
    pixel = len(shape) == 5
    agent, ref, specs = make_naf(shape, B, share)
    rng = np.random.default_rng(3)
    t = O.synthetic_batch(rng, B, shape, 2, pixel)
    try:
        out = ref.forward_backward(t)
        l_values, loss, v, a, vp = agent.naf.debug_values(HB(t))
        assert np.abs(l_values - out["l_values"]).max() < ATOL
        assert np.abs(v - out["value"][:, 0]).max() < ATOL
        adv = out["advantage"][:, 0]          # exp(l)^2-scaled: tolerance relative to its magnitude
        assert (np.abs(a - adv) <= ATOL * np.maximum(1.0, np.abs(adv))).all()
        assert np.abs(vp - out["target_value"][:, 0]).max() < ATOL
        assert abs(loss - out["loss"]) < ATOL * max(1.0, abs(out["loss"]))
        assert np.abs(agent.naf.forward(t[0]) - out["mu"]).max() < ATOL
        assert np.abs(agent.value_net.value_given(t[0]) - out["value"]).max() < ATOL
        one = agent.naf.action_given(t[0][0].astype(np.float32), add_noise=False)
        assert np.abs(one - ref.action_given(t[0][0])).max() < ATOL
        got_loss = agent.naf.train(HB(t))
        assert abs(got_loss - out["loss"]) < ATOL * max(1.0, abs(out["loss"]))
        cat = CatSpec(specs)
        assert_flat_close(cat, agent.naf.get_grads(), out["grads"], what="naf grads")
        ref.apply(out["grads"])
        assert_flat_close(cat, params_of(agent), ref.flat(), rel=1e-5, what="naf params")
    finally:
        agent.close()


@pytest.mark.parametrize("optimiser,args", [
    ("Momentum", {"learning_rate": 0.01, "momentum": 0.9}),
    ("Adam", {"learning_rate": 0.001}),
])
def test_naf_optimisers_over_several_steps(optimiser, args):
    shape, B = (8, 8, 3, 1, 2), 4
    agent, ref, specs = make_naf(shape, B, True, optimiser, args)
    rng = np.random.default_rng(8)
    try:
        for _ in range(3):
            t = O.synthetic_batch(rng, B, shape, 2, True)
            agent.naf.train(HB(t))
            ref.train(t)
        agent.target_value_net.update_weights()
        ref.update_targets()
        assert_flat_close(CatSpec(specs), params_of(agent), ref.flat(), rel=2e-5, what="%s params" % optimiser)
        assert_flat_close(specs[0], agent.target_value_net.get_params(), ref.target_value.flat(), rel=1e-6, what="target value")
    finally:
        agent.close()


@pytest.mark.parametrize("optimiser,oargs", [("Momentum", {"learning_rate": 0.01, "momentum": 0.9}), ("Adam", {"learning_rate": 0.001})])
def test_naf_fused_train_step_matches_oracle(optimiser, oargs):
    """the fused step (gradients' norm partials folded into their producers, the optimiser's step counter advanced by the heads
    kernel -- Adam's bias correction reads it) against the oracle's loop of train() calls"""
    shape, B = (16, 16, 3, 2, 1), 6
    agent, ref, specs = make_naf(shape, B, True, optimiser, oargs, replay_size=30)
    rng = np.random.default_rng(12)
    orm = OracleReplayMemory(30, shape, 2)
    try:
        for _ in range(10):
            n = int(rng.integers(2, 6))
            mk = lambda: rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)
            s0, seq = mk(), [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.integers(0, 3)), mk()) for _ in range(n)]
            agent.replay_memory.add_episode(s0, seq); orm.add_episode(s0, seq)
        nb = 3
        idxs = rng.integers(0, 30, nb * B)
        batches = []
        for i in range(nb):
            ob = orm.batch(idxs=idxs[i * B:(i + 1) * B])
            batches.append((ob.state_1, ob.action, ob.reward, ob.terminal_mask, ob.state_2))
        outs = ref.train_step(batches)
        agent.train_step(B, nb, idxs=idxs)
        assert_flat_close(CatSpec(specs), params_of(agent), ref.flat(), rel=2e-5, what="naf params")
        assert_flat_close(specs[0], agent.target_value_net.get_params(), ref.target_value.flat(), rel=1e-6, what="target value")
        st = agent.naf.last_stats()
        assert abs(st[0] - outs[-1]["loss"]) < ATOL * max(1.0, abs(outs[-1]["loss"])) and st[2] == 0
        # graph path: deterministic across two replays of a fresh agent is covered for DDPG; here just run it
        agent.replay_memory.fill_synthetic(25, seed=3)
        for _ in range(3):
            agent.train_step(B, 2)
        assert np.isfinite(params_of(agent)).all()
    finally:
        agent.close()


def test_check_numerics_raises_and_leaves_parameters_untouched():
    shape, B = (2, 2, 7), 4
    agent, ref, specs = make_naf(shape, B, True)
    rng = np.random.default_rng(1)
    t = O.synthetic_batch(rng, B, shape, 2, False)
    try:
        p = agent.naf.l_net.get_params()
        agent.naf.l_net.set_params(p * 0 + 1e4)         # exp(l) overflows -> L is inf (naf_cartpole.py:208)
        before = params_of(agent)
        with pytest.raises(FloatingPointError):
            agent.naf.train(HB(t))
        assert np.array_equal(before, params_of(agent))
    finally:
        agent.close()


def test_a_non_finite_minibatch_of_the_literal_loop_stops_every_later_update_and_is_not_counted():
    """`naf.train(batch)` on a replay draw returns at once (cpp_naf_train_rows_async); the check_numerics flag of a
    non-finite minibatch reaches the host up to two calls later.  Until then the device must behave as the reference does after
    tf.check_numerics raised (naf_cartpole.py:242-245,265): NO further optimiser step runs (the flag is sticky), the skipped update
    is not counted as an Adam step, and close() does not swallow an error nobody looked at."""
    import json
    from cartpoleplusplus_amd import naf_cartpole as F
    from tests.helpers import FakeEnv
    shape, B = (2, 2, 7), 4
    F.set_opts(F.default_opts(use_raw_pixels=False, action_repeats=2, batch_size=B, replay_memory_size=64,
                              share_input_state_representation=True, optimiser="Adam", optimiser_args=json.dumps({"learning_rate": 0.01})))
    agent = F.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
    agent.initialise_variables(seed=2)
    agent.post_var_init_setup()
    agent.replay_memory.fill_synthetic(40, seed=3)
    try:
        np.random.seed(5)
        for _ in range(3):                              # three good minibatches
            float(agent.naf.train(agent.replay_memory.batch(B)))
        step0 = agent.naf.get_optimiser_state()["step"]
        assert step0 == 3
        agent.naf.l_net.set_params(agent.naf.l_net.get_params() * 0 + 1e4)         # exp(l) overflows (naf_cartpole.py:208)
        before = params_of(agent)
        losses = [agent.naf.train(agent.replay_memory.batch(B)) for _ in range(2)]      # the bad one and one more, nobody looks
        assert np.array_equal(before, params_of(agent)), "an optimiser step ran behind a non-finite minibatch"
        assert agent.naf.get_optimiser_state()["step"] == step0, "a skipped update was counted as an optimiser step"
        with pytest.raises(FloatingPointError):
            float(losses[0])
        with pytest.raises(FloatingPointError):         # ... and the second loss is still unresolved: close() raises it
            agent.naf.close()
        agent.naf.handle = None
    finally:
        agent.value_net.close(); agent.target_value_net.close(); agent.replay_memory.close()


def test_checkpoint_resume_restores_the_adam_slots(tmp_path):
    """util.SaverUtil must checkpoint the optimiser's slot variables like tf.train.Saver does (util.py:88-90): a run resumed
    from a checkpoint continues bit for bit like the uninterrupted one (Adam: moments + beta powers)."""
    import json
    from cartpoleplusplus_amd import naf_cartpole as F, util
    from tests.helpers import FakeEnv
    shape, B = (16, 16, 3, 1, 2), 8

    def make():
        F.set_opts(F.default_opts(use_raw_pixels=True, render_height=16, render_width=16, num_cameras=1, action_repeats=2,
                                  batch_size=B, replay_memory_size=200, share_input_state_representation=True,
                                  optimiser="Adam", optimiser_args=json.dumps({"learning_rate": 0.001})))
        return F.NormalizedAdvantageFunctionAgent(FakeEnv(shape))
    idxs = np.random.default_rng(0).integers(0, 150, (6, 2 * B))
    a = make()
    try:
        a.initialise_variables(seed=4); a.post_var_init_setup()
        a.replay_memory.fill_synthetic(150, seed=9)
        for k in range(3):
            a.train_step(B, 2, idxs=idxs[k])
        saver = util.SaverUtil.__new__(util.SaverUtil)            # (no restore-or-init: save this agent as it is)
        saver.agent, saver.ckpt_dir, saver.save_freq = a, str(tmp_path), 3600
        saver.force_save()
        for k in range(3, 6):
            a.train_step(B, 2, idxs=idxs[k])
        want = [n.get_params() for n in a.networks()]
        st = a.naf.get_optimiser_state()
        assert int(st["step"]) == 12 and np.abs(st["m"]).max() > 0 and np.abs(st["v"]).max() > 0
    finally:
        a.close()
    b = make()
    try:
        util.SaverUtil(b, str(tmp_path), 3600)                    # restores the latest checkpoint
        # (no post_var_init_setup here: it would clobber the restored target with the value network, naf_cartpole.py:300-303)
        assert int(b.naf.get_optimiser_state()["step"]) == 6
        b.replay_memory.fill_synthetic(150, seed=9)
        for k in range(3, 6):
            b.train_step(B, 2, idxs=idxs[k])
        got = [n.get_params() for n in b.networks()]
    finally:
        b.close()
    for x, y in zip(want, got):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("shape,B,share", [((64, 64, 3, 2, 3), 256, True), ((50, 50, 3, 1, 2), 128, False)],
                         ids=["cfg4-64x64x18-B256-shared-trunk", "reference-defaults-50x50x6-B128-own-trunks"])
def test_cfg4_B256_graph_replayed_naf_step_against_f64_oracle(shape, B, share):
    naf_fused_step_against_f64_oracle(shape, B, share)


def naf_fused_step_against_f64_oracle(shape, B, share, fill="noise"):
    """cfg4 at the size the metric is quoted on (64x64x18, B = 256, shared trunk, Momentum as in exps/run_93.sh): the hipGraph REPLAY
    of the fused NAF step (naf_cartpole.py:365-373) on rows drawn by the device's sampler against oracle.NAF(float64) started from the
    same parameters and Momentum slots: loss at 1e-5, the pre-clip gradient list per variable at 2e-5 (the trunk's two
    discontinuities -- pool route, ReLU -- taken from the device and accepted only at rounding-level ties), the clipped Momentum
    update and the target update.  Second case: the reference's own defaults (50 x 50 x 6 render, batch 128,
    naf_cartpole.py's three networks on trunks of their own)."""
    import ctypes
    from cartpoleplusplus_amd import _lib
    from tests.helpers import (device_pool_codes, device_relu_active, pool_flips_are_near_ties, relu_flips_are_at_the_boundary)
    rows = 700
    oargs = {"learning_rate": 0.01, "momentum": 0.9}
    agent, _ref, specs = make_naf(shape, B, share, "Momentum", oargs, seed=4, replay_size=rows + 50)
    try:
        rm = agent.replay_memory
        if fill == "noise":
            rm.fill_synthetic(rows, seed=33)
        else:      # rendered episodes through add_episode (tests/test_gpu_render_inputs.py)
            from cartpoleplusplus_amd import naf_cartpole as F
            from tests.helpers import fill_with_rendered_episodes
            fill_with_rendered_episodes(agent, shape, rows, seed=33, blind_camera=(fill == "render-blind"), opts=F.opts)
        agent.train_step(B, 1)                                    # eager pass + capture (also fills the Momentum slots)
        nets = (agent.value_net, agent.naf.mu_net, agent.naf.l_net, agent.target_value_net)
        P = [n.get_params() for n in nets]
        opt = agent.naf.get_optimiser_state()
        agent.train_step(B, 1)                                    # hipGraph replay, device-drawn rows
        idxs = np.empty(B, np.int32)
        _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
        assert idxs.min() >= 0 and idxs.max() < rows and len(np.unique(idxs)) > B // 2
        grads, stats = agent.naf.get_grads(), agent.naf.last_stats()
        Pn = [n.get_params() for n in nets]
        trunks = [agent.value_net] if share else [agent.value_net, agent.naf.mu_net, agent.naf.l_net]
        codes, relu = [device_pool_codes(n, B) for n in trunks], [device_relu_active(n, B) for n in trunks]
        s1, s2 = rm.state[rm.state_1_idx[idxs]], rm.state[rm.state_2_idx[idxs]]
        hb = rm.batch(idxs=idxs)
        batch = (s1, hb.action, hb.reward, hb.terminal_mask, s2)
    finally:
        agent.close()
    vspec, mspec, lspec = specs
    ref = N.NAF(vspec, mspec, lspec, P[0], P[1], P[2], share, 2, np.float64, gradient_clip=5.0,
                optimiser=N.make_optimiser("Momentum", oargs))
    ref.target_value = O.Net(vspec, P[3], np.float64)
    ref.m = opt["m"].astype(np.float64)
    rnets = [ref.value] if share else [ref.value, ref.mu, ref.l]
    for net, cd, rl in zip(rnets, codes, relu):
        net.amax_override, net.relu_override = cd, rl
    out = ref.forward_backward(batch)
    flips = rflips = 0
    for net, cd, rl, what in zip(rnets, codes, relu, ("value", "mu", "l")):
        cache = net.forward(s1, white=ref._white(net, s1), training=True)
        flips += pool_flips_are_near_ties(cache, cd, what=what + " trunk")
        rflips += relu_flips_are_at_the_boundary(cache, rl, what=what + " trunk")
    assert abs(stats[0] - out["loss"]) < ATOL * max(1.0, abs(out["loss"])) and stats[2] == 0, (stats, out["loss"])
    cat = CatSpec(specs)
    assert_flat_close(cat, grads, out["grads"], rel=2e-5, what="NAF pre-clip grads vs f64 oracle (flips %d / %d)" % (flips, rflips))
    before = ref.flat()
    ref.apply(out["grads"])
    ref.update_targets()
    got = np.concatenate(Pn[:3])
    assert_flat_close(cat, got, ref.flat(), rel=2e-6, what="NAF params after the step")
    assert_flat_close(vspec, Pn[3], ref.target_value.flat(), rel=1e-6, what="target value net")
    d_got, d_want = got.astype(np.float64) - before, ref.flat() - before
    assert np.linalg.norm(d_got - d_want) < 2.0 ** -23 * np.linalg.norm(before) + 5e-5 * np.linalg.norm(d_want)


@pytest.mark.parametrize("switch", ["CPP_NAF_MLP", "CPP_NAF_HEADS"])
def test_the_other_naf_head_paths_stay_parity_green_when_selected(switch):
    """The shared-trunk NAF step runs everything between the first hidden layer and the one backward GEMM level that is left in
    naf_mlp_kernel (two hidden layers, the reference's 100, 50).  CPP_NAF_MLP=0 (ablation build) leaves the second hidden layer to GEMM
    levels and the heads to naf_heads_kernel (the path of one or three hidden layers); CPP_NAF_HEADS=0 keeps GEMM levels + naf_head_kernel
    (no hidden stack, action_dim > 4, wide representations).  The same parity cases must pass on both."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_naf.py"), "-q", "-x", "-m", "gpu",
                        "-k", "forward_gradients or fused_train_step or cfg4"], cwd=root,
                       env=dict(os.environ, CARTPOLEPP_ABLATION="1", **{switch: "0"}), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0 and " passed" in tail, tail


@pytest.mark.parametrize("shape,B,share,opt,oargs", [((64, 64, 3, 2, 3), 256, True, "Momentum", {"learning_rate": 0.01, "momentum": 0.9}),
                                                     ((64, 64, 3, 1, 3), 96, False, "Adam", {"learning_rate": 0.001})],
                         ids=["cfg4", "9ch-B96-own-trunks-adam"])
def test_the_fused_naf_step_is_bit_reproducible_from_run_to_run(shape, B, share, opt, oargs):
    """tests/test_gpu_distributed.py's three-run bit comparison for NAF (cfg4: the shared trunk at B = 256 -- conv1 forward walked as two
    bands of rows per image since round 6 -- under Momentum; three networks on trunks of their own under Adam): no kernel of the step may
    depend on timing.  (Round 6 met a store hazard of gfx950 that made a build differ from run to run: DESIGN 4.)"""
    runs = []
    for _ in range(3):
        agent, _ref, _ = make_naf(shape, B, share, opt, oargs, seed=9, replay_size=2 * B + 64)
        try:
            agent.replay_memory.fill_synthetic(2 * B, seed=12)
            for _ in range(6):
                agent.train_step(B, 3)
            agent.value_net.ctx.sync()
            runs.append(np.concatenate([params_of(agent), agent.target_value_net.get_params()]))
        finally:
            agent.close()
    for other in runs[1:]:
        assert np.array_equal(runs[0], other), float(np.abs(runs[0] - other).max())
