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
        kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:])))
    else:
        kw = dict(pixel=False, state_elems=int(np.prod(shape)))
    aspec = O.NetSpec("actor", 2, [100, 100, 50], **kw)
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
