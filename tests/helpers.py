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
    from cartpoleplusplus_amd import ddpg_cartpole as D
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


def assert_flat_close(spec, got, want, rel=2e-5, what=""):
    rows = per_var_report(spec, got, want)
    scale = float(np.linalg.norm(np.asarray(want, np.float64))) / np.sqrt(len(want)) + 1e-30
    bad = [r for r in rows if r[2] > rel and r[1] > rel * scale]
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
