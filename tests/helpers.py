"""Shared builders for the GPU parity tests: a device agent (through the public Python surface, i.e.
through the C ABI) and an oracle agent holding the same parameters."""
import numpy as np

from oracle import ddpg_np as O


def make_opts(D, shape, B, pixel, **kw):
    if pixel:
        o = D.default_opts(use_raw_pixels=True, render_height=shape[0], render_width=shape[1],
                           num_cameras=shape[3], action_repeats=shape[4], batch_size=B, **kw)
    else:
        o = D.default_opts(use_raw_pixels=False, action_repeats=shape[0], batch_size=B, **kw)
    D.set_opts(o)
    return o


class FakeEnv(object):
    class _S(object):
        def __init__(self, shape):
            self.shape = tuple(shape)

    def __init__(self, shape):
        self.observation_space, self.action_space = self._S(shape), self._S((1, 2))


def make_pair(shape, B, pixel, seed=0, replay_size=64, perturb=True, dt=np.float64, **optkw):
    """returns (agent, oracle DDPG, specs).  Parameters are perturbed away from the near-zero actor
    head / zero biases so every path carries signal."""
    import os
    from cartpoleplusplus_amd import ddpg_cartpole as D
    if os.environ.get("TEST_EXACT_PRODUCTS") == "1":      # (test snippets run in subprocesses: the TEST's switch for --exact-products)
        optkw.setdefault("exact_products", True)
    make_opts(D, shape, B, pixel, replay_memory_size=replay_size, **optkw)
    agent = D.DeepDeterministicPolicyGradientAgent(FakeEnv(shape))
    agent.initialise_variables(seed=seed)
    rng = np.random.default_rng(seed + 100)
    if perturb:
        for net in (agent.actor, agent.critic):
            p = net.get_params()
            net.set_params(p + rng.normal(0, 0.05, p.shape).astype(np.float32))
    agent.post_var_init_setup()
    if perturb:
        for net in (agent.target_actor, agent.target_critic):
            p = net.get_params()
            net.set_params(p + rng.normal(0, 0.01, p.shape).astype(np.float32))
    if pixel:
        kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:])), batch_norm=bool(optkw.get("use_batch_norm", False)))
    else:
        kw = dict(pixel=False, state_elems=int(np.prod(shape)))
    aspec = O.NetSpec("actor", 2, [100, 100, 50], dropout=bool(optkw.get("use_dropout", False)), **kw)
    cspec = O.NetSpec("critic", 2, [100, 100, 50], **kw)
    ref = O.DDPG(aspec, cspec, agent.actor.get_params(), agent.critic.get_params(), dt)
    ref.set_targets(agent.target_actor.get_params(), agent.target_critic.get_params())
    return agent, ref, (aspec, cspec)


def per_var_report(spec, got, want):
    """[(name, max_abs_err, rel_l2_err)] per variable of a flat vector."""
    rows, off = [], 0
    for name, shp in spec.layout():
        n = int(np.prod(shp))
        g, w = got[off:off + n].astype(np.float64), np.asarray(want[off:off + n], np.float64)
        denom = np.linalg.norm(w)
        rows.append((name, float(np.abs(g - w).max()), float(np.linalg.norm(g - w) / denom) if denom > 0 else float(np.abs(g).max())))
        off += n
    return rows


def assert_flat_close(spec, got, want, rel=2e-5, what="", abs_floor=0.0, rel_of=None):
    """abs_floor: an absolute error one-element variables are not held below (the critic's q_value bias gradient is 2 mean(td), a
    sum with cancellation: it cannot be closer to the oracle than the Q values that make up td).  rel_of: {variable: tolerance}
    that replaces `rel` where it is larger (the float32 evaluation's own distance from float64, times a factor)."""
    rows = per_var_report(spec, got, want)
    single = set(name for name, shp in spec.layout() if int(np.prod(shp)) == 1)
    scale = float(np.linalg.norm(np.asarray(want, np.float64))) / np.sqrt(len(want)) + 1e-30
    tol = lambda name: max(rel, (rel_of or {}).get(name, 0.0))
    bad = [r for r in rows if r[2] > tol(r[0]) and r[1] > tol(r[0]) * scale and not (r[0] in single and r[1] <= abs_floor)]
    msg = "\n".join("%-28s max_abs=%.3e rel_l2=%.3e" % r for r in rows)
    assert not bad, "%s mismatch (rel tol %g):\n%s" % (what, rel, msg)


def device_pool_codes(net, B):
    """arg-max codes (0..3) of the 2x2 pooling windows of the device network's last forward, per conv layer."""
    from cartpoleplusplus_amd._lib import lib, check, ptr
    out = {}
    for i, (name, _k, _co) in enumerate(O.CONV_DEFS):
        shp = getattr(net, "pool%d" % (i + 1)).get_shape()
        codes = np.empty((B,) + tuple(shp[1:]), np.float32)
        check(lib.cpp_net_get_pool(net.handle, 11 + i, B, ptr(codes)))
        out[name] = codes.astype(np.uint8)
    return out


def assert_grads_close_modulo_pool_ties(spec, device_net, B, oracle_net, oracle_cache_fn, oracle_grads_fn, got,
                                        what="", rel=2e-5, margin_tol=1e-5):
    """Gradient parity with the max-pool's discontinuity taken into account.  The pool routes a window's gradient
    to its arg-max; where the two largest pre-activations of a window agree to rounding level, a different (but
    equally valid) f32 summation order picks the other element and moves that gradient to a neighbouring pixel.
    Such a flip is accepted only if the oracle itself sees a near tie there (margin <= margin_tol relative); the
    oracle's gradient is then recomputed with the device's choice at exactly those windows and must match."""
    want = oracle_grads_fn()
    try:
        assert_flat_close(spec, got, want, rel=rel, what=what)
        return 0
    except AssertionError as e:
        first = e
    cache = oracle_cache_fn()
    codes = device_pool_codes(device_net, B)
    flips = 0
    for name, _k, _co in O.CONV_DEFS:
        _x, pooled, amax, _h, _w = cache[name]
        margin = cache[name + ":margin"]
        diff = (codes[name] != amax) & (pooled > 0)
        bad = diff & (margin > margin_tol * np.maximum(1.0, np.abs(pooled)))
        assert not bad.any(), "%s: %s arg-max differs at %d window(s) that are not near ties\n%s" % (
            what, name, int(bad.sum()), first)
        flips += int(diff.sum())
    assert flips > 0, first
    oracle_net.amax_override = codes
    try:
        want = oracle_grads_fn()
    finally:
        oracle_net.amax_override = None
    assert_flat_close(spec, got, want, rel=rel,
                      what="%s (oracle re-run with the device's choice at %d near-tie pooling windows)" % (what, flips))
    return flips


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al.), the generator of the replay sampler and of the dropout masks."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF,
             ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def dropout_masks(namespace, hidden, B, step):
    """the keep masks the device draws for the `step`-th training-mode forward of network `namespace`
    (include/cartpolepp_abi.h, cpp_net_spec.use_dropout): {'h<i>': (B, units) of 0/1}."""
    import zlib
    seed = zlib.crc32(namespace.encode()) & 0xffffffff
    out = {}
    for layer, units in enumerate(hidden):
        m = np.empty((B, units), np.float64)
        for b in range(B):
            for j in range(units):
                m[b, j] = philox4x32_10([b * units + j, layer, step & 0xFFFFFFFF, step >> 32], [seed, 0])[0] & 1
        out["h%d" % layer] = m
    return out


def device_relu_active(net, B):
    """which pooled conv outputs of the device network's last forward are > 0 (the cells its backward lets gradient through)."""
    return {name: getattr(net, "pool%d" % (i + 1)).eval(B) > 0 for i, (name, _k, _co) in enumerate(O.CONV_DEFS)}


def relu_flips_are_at_the_boundary(cache, device_active, tol=1e-5, what=""):
    """a forward cache of the oracle vs the device's ReLU decisions on the pooled conv outputs: where they differ, the oracle's
    own pre-activation maximum must be zero to rounding (|z| <= tol) -- max(z, 0) is continuous there but its gradient is not.
    Returns the number of such cells."""
    flips = 0
    for name, _k, _co in O.CONV_DEFS:
        _x, pooled, _amax, _h, _w = cache[name]
        zmax = cache[name + ":zmax"]
        diff = np.asarray(device_active[name]).reshape(pooled.shape) != (pooled > 0)
        bad = diff & (np.abs(zmax) > tol)
        assert not bad.any(), "%s: %s ReLU decision differs at %d cell(s) that are not at the boundary (largest |z| %.3e)" % (
            what, name, int(bad.sum()), float(np.abs(zmax[bad]).max()))
        flips += int(diff.sum())
    return flips


def pool_flips_are_near_ties(cache, device_codes, margin_tol=1e-5, what=""):
    """a forward cache computed with `amax_override = device_codes`: wherever the device routed a pooling window to another
    element than the oracle's own arg-max (and the pooled value is positive, i.e. the route carries gradient), the oracle
    must itself see a near tie there -- two largest pre-activations within margin_tol (relative to max(1, |value|)).
    Returns the number of such windows."""
    flips = 0
    for name, _k, _co in O.CONV_DEFS:
        _x, pooled, _amax, _h, _w = cache[name]
        own, margin = cache[name + ":amax_own"], cache[name + ":margin"]
        diff = (np.asarray(device_codes[name]).reshape(own.shape) != own) & (pooled > 0)
        bad = diff & (margin > margin_tol * np.maximum(1.0, np.abs(pooled)))
        assert not bad.any(), "%s: %s arg-max differs at %d window(s) that are not near ties (largest margin %.3e)" % (
            what, name, int(bad.sum()), float(margin[bad].max()))
        flips += int(diff.sum())
    return flips


def fill_with_rendered_episodes(agent, shape, rows, seed=0, blind_camera=False, as_u8=False, glint=0.0, opts=None):
    """`rows` transitions of random-policy episodes of the software-rasterised cart and pole (synthetic_env.RasterCartpole: flat
    backgrounds, R nearly identical repeat frames, optionally a camera that sees one colour only) into the agent's replay memory,
    through ReplayMemory.add_episode as the reference's rollout loop does (ddpg_cartpole.py:315-326)."""
    from cartpoleplusplus_amd import ddpg_cartpole as D
    from cartpoleplusplus_amd.synthetic_env import RasterCartpole, play_episodes
    env = RasterCartpole(opts if opts is not None else D.opts, seed=seed + 1, blind_camera=blind_camera, glint=glint)
    for first, seq in play_episodes(env, rows, np.random.default_rng(seed + 2)):
        if as_u8:
            first = np.rint(first * 255).astype(np.uint8)
            seq = [(a, r, np.rint(s2 * 255).astype(np.uint8)) for a, r, s2 in seq]
        agent.replay_memory.add_episode(first, seq)
    assert agent.replay_memory.size() == rows


F32_GRAD_FACTOR = 1.5


def fused_step_against_f64_oracle(shape, B, rows, replay_store="f16", replay_size=None, seed=0, graph=True,
                                  atol=1e-5, grad_rel=2e-5, param_rel=2e-6, warm="philox", report_only=False,
                                  fill="noise", f32_twin=False, flip_tol=1e-5):
    """ONE minibatch of the fused inner step (cpp_ddpg_train_step, default kernels: f16-pipe conv1 reading the replay store
    through the sampled slots, bf16-pipe conv2, fused heads, paired launches) -- with graph=True the hipGraph REPLAY of it,
    on rows drawn by the device's Philox sampler -- against oracle.DDPG(float64) on the same rows and the same starting
    parameters: actions / Q / TD / dQ/da at `atol` (north_star: 1e-5), both pre-clip gradient lists per variable at
    `grad_rel` (pool routes: the device's, accepted only at near ties), the clipped SGD result and the target updates."""
    import ctypes
    from cartpoleplusplus_amd import _lib
    agent, _ref, (aspec, cspec) = make_pair(shape, B, True, seed=seed, replay_size=replay_size or rows + 50,
                                           replay_store=replay_store)
    report = {}
    try:
        rm = agent.replay_memory
        if fill == "noise":
            rm.fill_synthetic(rows, seed=21 + seed)
        else:
            assert fill in ("render", "render-blind", "render-glint"), fill
            fill_with_rendered_episodes(agent, shape, rows, seed=seed, blind_camera=(fill != "render"),
                                        glint=0.02 if fill == "render-glint" else 0.0)
        if graph or warm == "philox-eager":
            agent.train_step(B, 1)                        # eager pass + capture
        elif warm == "rows":
            agent.train_step(B, 1, idxs=np.random.default_rng(seed + 77).integers(0, rows, B).astype(np.int32))
        nets = (agent.actor, agent.critic, agent.target_actor, agent.target_critic)
        P = [n.get_params() for n in nets]
        if graph:
            agent.train_step(B, 1)                        # hipGraph replay, device-drawn rows
            idxs = np.empty(B, np.int32)
            _lib.check(_lib.lib.cpp_replay_last_indexes(rm.handle, B, idxs.ctypes.data_as(ctypes.c_void_p)))
        else:
            idxs = np.random.default_rng(seed + 5).integers(0, rows, B).astype(np.int32)
            agent.train_step(B, 1, idxs=idxs)             # same launch sequence, eager, caller's rows
        assert idxs.min() >= 0 and idxs.max() < rows and len(np.unique(idxs)) > B // 2
        actions, dq_da, q, td = agent.trainer.last_values(B)
        g_a, g_c = agent.actor.get_grads(), agent.critic.get_grads()
        stats = agent.trainer.last_stats()
        Pn = [n.get_params() for n in nets]
        codes_a, codes_c = device_pool_codes(agent.actor, B), device_pool_codes(agent.critic, B)
        relu_a, relu_c = device_relu_active(agent.actor, B), device_relu_active(agent.critic, B)
        pools_c = [getattr(agent.critic, "pool%d" % i).eval(B) for i in (1, 2, 3)]
        # the minibatch, read back through paths that do not involve the gather kernel's state copy
        s1, s2 = rm.state[rm.state_1_idx[idxs]], rm.state[rm.state_2_idx[idxs]]
        hb = rm.batch(idxs=idxs)
        a, r, m = hb.action, hb.reward, hb.terminal_mask
        assert np.array_equal(m[:, 0], rm.terminal_mask[idxs, 0]) and np.array_equal(r[:, 0], rm.reward[idxs, 0])
    finally:
        agent.close()
    ref = O.DDPG(aspec, cspec, P[0], P[1], np.float64)
    ref.set_targets(P[2], P[3])
    # the two discontinuities of the trunk's gradient -- which element of a 2x2 window carries it, and whether the ReLU lets it
    # through -- are taken from the device and must coincide with the oracle's own except at rounding-level ties
    ref.actor.amax_override, ref.critic.amax_override = codes_a, codes_c
    ref.actor.relu_override, ref.critic.relu_override = relu_a, relu_c
    t = (s1, a, r, m, s2)
    ag = ref.actor_gradients(s1)
    cg = ref.critic_gradients(t)
    for key, fn, args in (("flips_actor", pool_flips_are_near_ties, (ag["cache_actor"], codes_a)),
                          ("flips_critic", pool_flips_are_near_ties, (cg["cache_critic"], codes_c)),
                          ("relu_flips_actor", relu_flips_are_at_the_boundary, (ag["cache_actor"], relu_a)),
                          ("relu_flips_critic", relu_flips_are_at_the_boundary, (cg["cache_critic"], relu_c))):
        try:
            report[key] = fn(*args, flip_tol, what=key.split("_")[-1])
        except AssertionError as e:
            if not report_only:
                raise
            report[key] = "FAILED: %s" % e
    report["err_actions"] = float(np.abs(actions - ag["actions"]).max())
    report["err_dq_da"] = float(np.abs(dq_da - ag["dq_da"]).max())
    report["err_q"] = float(np.abs(q - cg["q"]).max())
    report["err_td"] = float(np.abs(td - cg["td"]).max())
    report["q_scale"] = float(np.abs(cg["q"]).max())
    for i, (name, _k, _co) in enumerate(O.CONV_DEFS):
        want = cg["cache_critic"][name][1]
        report["err_pool%d" % (i + 1)] = float(np.abs(pools_c[i].reshape(want.shape) - want).max())
        report["mag_pool%d" % (i + 1)] = float(np.abs(want).max())
    if f32_twin:
        # the same evaluation in float32 numpy (the rounding an f32 implementation such as the reference's TF CPU kernels is
        # entitled to): how far IT sits from the float64 values on these inputs
        ref32 = O.DDPG(aspec, cspec, P[0], P[1], np.float32)
        ref32.set_targets(P[2], P[3])
        ref32.actor.amax_override, ref32.critic.amax_override = codes_a, codes_c      # (the same routes: rounding is what is compared)
        ref32.actor.relu_override, ref32.critic.relu_override = relu_a, relu_c
        ag32, cg32 = ref32.actor_gradients(s1), ref32.critic_gradients(t)
        f32_rel_a = {n: F32_GRAD_FACTOR * r_ for n, _m, r_ in per_var_report(aspec, ag32["grads"], ag["grads"])}
        f32_rel_c = {n: F32_GRAD_FACTOR * r_ for n, _m, r_ in per_var_report(cspec, cg32["grads"], cg["grads"])}
        report["f32_rel_actor_grads"] = max(f32_rel_a.values()) / F32_GRAD_FACTOR
        report["f32_rel_critic_grads"] = max(f32_rel_c.values()) / F32_GRAD_FACTOR
        report["f32_err_actions"] = float(np.abs(ag32["actions"] - ag["actions"]).max())
        report["f32_err_dq_da"] = float(np.abs(ag32["dq_da"] - ag["dq_da"]).max())
        report["f32_err_q"] = float(np.abs(cg32["q"] - cg["q"]).max())
        report["f32_err_td"] = float(np.abs(cg32["td"] - cg["td"]).max())
        for i, (name, _k, _co) in enumerate(O.CONV_DEFS):
            report["f32_err_pool%d" % (i + 1)] = float(np.abs(cg32["cache_critic"][name][1] - cg["cache_critic"][name][1]).max())
        report["white_scale_max"] = float(np.max(cg["cache_critic"]["white"][0]))
        report["white_scale_min"] = float(np.min(cg["cache_critic"]["white"][0]))
    if report_only:
        report["events"] = {k: report[k] for k in report if "flips" in k}
        report["actor"] = [(n, "%.2e" % r_) for n, _m, r_ in per_var_report(aspec, g_a, ag["grads"])][:6]
        report["critic"] = [(n, "%.2e" % r_) for n, _m, r_ in per_var_report(cspec, g_c, cg["grads"])][:6]
        return report
    assert report["err_actions"] < atol and report["err_dq_da"] < atol, report
    assert report["err_q"] < atol and report["err_td"] < atol, report
    assert abs(stats[0] - cg["loss"]) < atol * max(1.0, abs(cg["loss"])), (stats, cg["loss"])
    # (f32_twin: a variable's gradient may be as far from float64 as F32_GRAD_FACTOR x the float32 numpy evaluation's -- the critic's
    # gradients are linear in TD, and on correlated minibatches (renders) sum_b td_b cancels: 5e-6 on TD is 5e-5 of the head's gradient)
    assert_flat_close(aspec, g_a, ag["grads"], rel=grad_rel, what="actor pre-clip grads vs f64 oracle", rel_of=f32_rel_a if f32_twin else None)
    try:
        assert_flat_close(cspec, g_c, cg["grads"], rel=grad_rel, what="critic pre-clip grads vs f64 oracle",
                          abs_floor=2.0 * report["err_td"], rel_of=f32_rel_c if f32_twin else None)
    except AssertionError:
        # the critic's gradients are LINEAR in TD: g = (2 / B) sum_b td_b dq_b/dtheta.  On a correlated minibatch (consecutive
        # renders) the td_b nearly cancel in that sum and the TD error admitted above (< atol) is a large fraction of what is left.
        # Second chance: the oracle's backward pass fed with the DEVICE's TD values -- the backward arithmetic alone, at grad_rel
        cg_dev = ref.critic_gradients(t, td_override=td)
        assert_flat_close(cspec, g_c, cg_dev["grads"], rel=grad_rel, what="critic pre-clip grads vs f64 oracle's backward pass of the device's TD")
        report["critic_grads_checked_at_device_td"] = True
        nc_dev = float(np.linalg.norm(cg_dev["grads"]))
    report["rel_actor_grads"] = max(r_[2] for r_ in per_var_report(aspec, g_a, ag["grads"]))
    report["rel_critic_grads"] = max(r_[2] for r_ in per_var_report(cspec, g_c, cg["grads"]))
    na, nc = float(np.linalg.norm(ag["grads"])), float(np.linalg.norm(cg["grads"]))
    # (the reported critic norm is the norm of the gradient checked above: where that check needed the device's TD values -- B = 1 with a
    # TD of 1e-2: the admitted 1e-5 on TD is 1e-3 of the gradient -- the norm is held to the same gradient; rs16_geometry_parity.py 601, draw 1)
    nc_ref = nc_dev if report.get("critic_grads_checked_at_device_td") else nc
    assert abs(stats[1] - na) < 1e-4 * max(1.0, na) and abs(stats[2] - nc_ref) < 1e-4 * max(1.0, nc_ref), (stats, na, nc, nc_ref)
    # clip + SGD (util.py:47-50, ddpg_cartpole.py:118-119,218) and the target updates (:336-337) on top of them
    hp = O.DEFAULT_HYPER
    ca, _ = O.clip_by_global_norm(ag["grads"], hp.gradient_clip, np.float64)
    cc, _ = O.clip_by_global_norm(cg["grads"], hp.gradient_clip, np.float64)
    want_a, want_c = P[0] - hp.actor_lr * ca, P[1] - hp.critic_lr * cc
    assert_flat_close(aspec, Pn[0], want_a, rel=param_rel, what="actor params after the step")
    assert_flat_close(cspec, Pn[1], want_c, rel=param_rel, what="critic params after the step")
    assert_flat_close(aspec, Pn[2], O.soft_update(P[2], want_a, hp.target_update_rate, np.float64), rel=1e-6, what="target actor")
    assert_flat_close(cspec, Pn[3], O.soft_update(P[3], want_c, hp.target_update_rate, np.float64), rel=1e-6, what="target critic")
    # the update itself (not hidden behind the much larger parameters): delta vs -lr * clipped gradient
    for name, new, old, want in (("actor", Pn[0], P[0], want_a), ("critic", Pn[1], P[1], want_c)):
        d_got, d_want = new.astype(np.float64) - old, want - old
        report["rel_delta_" + name] = float(np.linalg.norm(d_got - d_want) / np.linalg.norm(d_want))
        # f32 parameters: storing theta - lr*g rounds at |theta| * 2^-24 per element, on top of the gradient's own error
        # (... which, with f32_twin, may be as far from float64 as F32_GRAD_FACTOR x the float32 numpy evaluation's own gradients are: on
        # nearly constant channels that is 1e-3, not 5e-5)
        rel_d = max(5e-5, F32_GRAD_FACTOR * report["f32_rel_%s_grads" % name]) if f32_twin else 5e-5
        bound = 2.0 ** -23 * np.linalg.norm(old) + rel_d * np.linalg.norm(d_want)
        assert np.linalg.norm(d_got - d_want) < bound, (name, report, bound)
    return report


def philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32-10: uint64 numpy arrays (values < 2^32) in, four uint32 words out."""
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85, np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) for x in (c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def synthetic_state_codes(slot, elems, seed):
    """the 8-bit pixel codes cpp_replay_fill_synthetic writes into state slot `slot` (csrc/replay.hip: byte e of the Philox block
    of flat position // 16, key = seed), regenerated on the host -- an independent witness for gathers from anywhere in a store
    of any size (64-bit positions)."""
    first = int(slot) * int(elems)
    blocks = np.arange(first // 16, (first + elems + 15) // 16, dtype=np.uint64)
    w = philox4x32_10_np(blocks & np.uint64(0xFFFFFFFF), blocks >> np.uint64(32), np.full_like(blocks, 0x5eed), np.full_like(blocks, 1),
                         seed & 0xFFFFFFFF, seed >> 32)
    by = np.stack([(w[e >> 2] >> np.uint64(8 * (e & 3))) & np.uint64(0xFF) for e in range(16)], axis=1).astype(np.uint8).ravel()
    off = first - int(blocks[0]) * 16
    return by[off:off + elems]
