"""numpy restatement of the reference's DDPG networks + train step.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for this file: the reference has no numerical test of any network
output and cannot run here (Python 2 / TensorFlow 0.x).  The arithmetic lives in
TensorFlow r0.9-r0.11 + tensorflow.contrib.slim (un-vendored, unpinned); this file
restates the published semantics of the ops the reference *calls*, anchored on the
call sites below, and is cross-checked against torch-CPU autograd in
tests/test_oracle_vs_torch.py.

Call sites followed (all paths relative to /root/reference):
  reshape + whitening ............. base_network.py:85-99   (tf.nn.moments one-pass
                                    var = E[x^2]-mu^2; tf.nn.batch_normalization
                                    x*inv + (-mu*inv), inv = rsqrt(var + 1e-6))
  conv1/pool1/conv2/pool2/conv3/pool3 base_network.py:103-127 (slim.conv2d: stride 1,
                                    SAME, bias, ReLU; slim.max_pool2d 2x2/2 VALID)
  flatten + hidden stack .......... base_network.py:58-71, 129-134 (slim.fully_connected, ReLU)
  actor head ...................... ddpg_cartpole.py:94-100 (tanh, W~U(+-1e-3))
  critic (pixel, *intent*) ........ ddpg_cartpole.py:166-171, 180-184 (flatten->200->50->
                                    concat action->50->q;  SURVEY appendix B2)
  critic (low-dim) ................ ddpg_cartpole.py:172-177 (concat(flat,action)->stack->q; B1)
  bellman target / TD loss ........ ddpg_cartpole.py:199-209
  dQ/da ........................... ddpg_cartpole.py:220-222
  actor gradients (sum over batch)  ddpg_cartpole.py:111-113
  clip_by_global_norm(5) .......... util.py:45-50
  SGD apply ....................... ddpg_cartpole.py:118-119, 213, 218
  target copy / soft update ....... base_network.py:20-49
  inner train step order .......... ddpg_cartpole.py:329-337

All functions take a numpy dtype `dt`: np.float64 is the ground truth, np.float32 the
"expected rounding" twin and the timed CPU baseline (BLAS-threaded matmuls).
"""
import collections

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

CONV_DEFS = (("conv1", 5, 10), ("conv2", 5, 10), ("conv3", 3, 10))  # base_network.py:103,111,119
WHITEN_EPS = 1e-6                                                   # base_network.py:99

Hyper = collections.namedtuple(
    "Hyper", "actor_lr critic_lr discount gradient_clip target_update_rate")
DEFAULT_HYPER = Hyper(1e-3, 1e-2, 0.99, 5.0, 1e-4)  # ddpg_cartpole.py:35,41-43; util.py:11


# --------------------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------------------
class NetSpec(object):
    """kind: 'actor' | 'critic'.  pixel nets take (H, W, C) with C = 3*cameras*repeats
    (base_network.py:85-90); low-dim nets take `state_elems` flattened inputs."""

    def __init__(self, kind, action_dim, hidden, pixel, H=0, W=0, C=0, state_elems=0, batch_norm=False, dropout=False):
        assert kind in ("actor", "critic")
        # --use-dropout (base_network.py:69-70): slim.dropout (keep 0.5) after the ReLU of every layer built by
        # hidden_layers_starting_at *with opts*: the 'h<i>' stacks of actor-like networks; the critics have none
        self.dropout = bool(dropout) and kind == "actor"
        self.kind, self.action_dim, self.pixel = kind, int(action_dim), bool(pixel)
        # --use-batch-norm (base_network.py:74-79): slim.batch_norm between every conv and its ReLU.  The conv then has
        # no bias; the slot "<conv>/biases" of the flat layout holds BatchNorm/beta instead (same size, same place).
        self.batch_norm = bool(batch_norm) and self.pixel
        self.hidden = [int(h) for h in hidden]
        self.H, self.W, self.C = int(H), int(W), int(C)
        if self.pixel:
            h, w = self.H, self.W
            self.conv_hw = []
            for _ in CONV_DEFS:
                self.conv_hw.append((h, w))
                h, w = h // 2, w // 2
            self.flat = h * w * CONV_DEFS[-1][2]
            self.state_elems = self.H * self.W * self.C
        else:
            self.flat = int(state_elems)
            self.state_elems = int(state_elems)
        self.fc = self._fc_layers()

    def _fc_layers(self):
        """[(name, n_in, n_out, act, concat_action)] in creation order."""
        A = self.action_dim
        out, n_in = [], self.flat
        if self.kind == "actor":
            for i, h in enumerate(self.hidden):
                out.append(("h%d" % i, n_in, h, "relu", False))
                n_in = h
            out.append(("output_action", n_in, A, "tanh", False))
        elif self.pixel:
            out.append(("hidden1", n_in, 200, "relu", False))
            out.append(("hidden2", 200, 50, "relu", False))
            out.append(("hidden3", 50 + A, 50, "relu", True))
            out.append(("q_value", 50, 1, "linear", False))
        else:
            n_in += A
            for i, h in enumerate(self.hidden):
                out.append(("h%d" % i, n_in, h, "relu", i == 0))
                n_in = h
            out.append(("q_value", n_in, 1, "linear", False))
        return out

    def layout(self):
        """[(name, shape)] in flat-buffer order (= TF variable creation order)."""
        out = []
        if self.pixel:
            cin = self.C
            for name, k, cout in CONV_DEFS:
                out.append((name + "/weights", (k, k, cin, cout)))
                out.append((name + "/biases", (cout,)))
                cin = cout
        for name, n_in, n_out, _a, _c in self.fc:
            out.append((name + "/weights", (n_in, n_out)))
            out.append((name + "/biases", (n_out,)))
        return out

    def num_params(self):
        return int(sum(int(np.prod(s)) for _n, s in self.layout()))


def init_params(spec, rng):
    """slim defaults: xavier-uniform weights, zero biases; actor head U(+-1e-3)
    (ddpg_cartpole.py:94).  Returns a flat float32 vector in layout() order."""
    parts = []
    for name, shape in spec.layout():
        if name.endswith("/biases"):
            parts.append(np.zeros(shape, np.float32))
        elif name.startswith("output_action"):
            parts.append(rng.uniform(-1e-3, 1e-3, shape).astype(np.float32))
        else:
            if len(shape) == 4:
                fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
            else:
                fan_in, fan_out = shape
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            parts.append(rng.uniform(-lim, lim, shape).astype(np.float32))
    return np.concatenate([p.ravel() for p in parts])


def unflatten(spec, flat, dt):
    out, off = collections.OrderedDict(), 0
    for name, shape in spec.layout():
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[off:off + n], dtype=dt).reshape(shape)
        off += n
    assert off == len(flat), (off, len(flat))
    return out


def flatten(spec, named, dt=np.float32):
    return np.concatenate([np.asarray(named[n], dtype=dt).ravel() for n, _s in spec.layout()])


# --------------------------------------------------------------------------------------
# ops
# --------------------------------------------------------------------------------------
def whiten_stats(x, dt):
    """base_network.py:95-96: per-channel moments over (batch, y, x), one-pass form var = E[x^2] - mu^2.
    The sums are always taken in float64 and only the resulting scale / shift are rounded to `dt`: a naive
    float32 one-pass variance over 10^6 pixels loses 3-4 digits to cancellation (measured 1e-4 relative on
    the scale), which would make the float32 twin a worse reference than the device path it checks."""
    x64 = np.asarray(x, dtype=np.float64)
    n = x64.shape[0] * x64.shape[1] * x64.shape[2]
    flat = x64.reshape(n, x64.shape[3])
    mean = flat.mean(axis=0)
    var = (flat * flat).mean(axis=0) - mean * mean
    inv = 1.0 / np.sqrt(var + WHITEN_EPS)
    return inv.astype(dt), (-mean * inv).astype(dt)     # y = x*scale + shift  (base_network.py:97-99)


BN_EPS = 1e-3          # slim.batch_norm default epsilon (decay 0.999, center=True, scale=False)


def bn_stats(z, dt, training):
    """slim.batch_norm statistics over (batch, y, x): batch moments (one-pass form, float64 sums as in whiten_stats)
    when training; otherwise the moving averages -- which the reference never updates (its train ops do not depend on
    UPDATE_OPS), so they are the initial mean 0 / variance 1.  Returns (mean, 1/sqrt(var + eps))."""
    C = z.shape[3]
    if not training:
        return np.zeros(C, dt), (np.ones(C) / np.sqrt(1.0 + BN_EPS)).astype(dt)
    z64 = np.asarray(z, dtype=np.float64).reshape(-1, C)
    mean = z64.mean(axis=0)
    var = (z64 * z64).mean(axis=0) - mean * mean
    return mean.astype(dt), (1.0 / np.sqrt(var + BN_EPS)).astype(dt)


def whiten(x, dt):
    scale, shift = whiten_stats(x, dt)
    return (x.astype(dt, copy=False) * scale + shift).astype(dt)


def _im2col(x, k):
    """x (B,H,W,C) -> (B,H,W, k*k*C) with column order (ky, kx, c); SAME zero padding."""
    p = k // 2
    xp = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))
    win = sliding_window_view(xp, (k, k), axis=(1, 2))      # (B,H,W,C,ky,kx)
    return np.ascontiguousarray(win.transpose(0, 1, 2, 4, 5, 3)).reshape(
        x.shape[0], x.shape[1], x.shape[2], k * k * x.shape[3])


def conv_fwd(x, W, b, chunk=8):
    """slim.conv2d pre-activation: stride 1, SAME, cross-correlation, + bias."""
    k, cout = W.shape[0], W.shape[3]
    Wm = W.reshape(-1, cout)
    z = np.empty(x.shape[:3] + (cout,), dtype=x.dtype)
    for s in range(0, x.shape[0], chunk):
        z[s:s + chunk] = _im2col(x[s:s + chunk], k) @ Wm + b
    return z


def conv_bwd(x, W, dz, need_dx, chunk=8):
    k, cin, cout = W.shape[0], W.shape[2], W.shape[3]
    p = k // 2
    Wm = W.reshape(-1, cout)
    dW = np.zeros_like(Wm)
    dx = np.zeros_like(x) if need_dx else None
    for s in range(0, x.shape[0], chunk):
        cols = _im2col(x[s:s + chunk], k)
        g = dz[s:s + chunk]
        dW += cols.reshape(-1, cols.shape[-1]).T @ g.reshape(-1, cout)
        if need_dx:
            dcols = (g @ Wm.T).reshape(g.shape[:3] + (k, k, cin))
            nb, H, Wd = g.shape[:3]
            dxp = np.zeros((nb, H + 2 * p, Wd + 2 * p, cin), dtype=x.dtype)
            for ky in range(k):
                for kx in range(k):
                    dxp[:, ky:ky + H, kx:kx + Wd, :] += dcols[:, :, :, ky, kx, :]
            dx[s:s + chunk] = dxp[:, p:p + H, p:p + Wd, :]
    return dW.reshape(W.shape), dz.sum(axis=(0, 1, 2)), dx


def relu_pool_fwd(z, with_zmax=False):
    """relu then 2x2/2 VALID max-pool (base_network.py:107): pool(relu(z)) == relu(max z).
    Returns pooled (B,H//2,W//2,C) and the window arg-max code dy*2+dx (with_zmax: also the pre-ReLU window maximum)."""
    B, H, W, C = z.shape
    hp, wp = H // 2, W // 2
    win = z[:, :2 * hp, :2 * wp, :].reshape(B, hp, 2, wp, 2, C).transpose(0, 1, 3, 5, 2, 4)
    win = win.reshape(B, hp, wp, C, 4)
    amax = win.argmax(axis=-1)
    zmax = np.take_along_axis(win, amax[..., None], axis=-1)[..., 0]
    if with_zmax:
        return np.maximum(zmax, 0), amax.astype(np.uint8), zmax
    return np.maximum(zmax, 0), amax.astype(np.uint8)


def pool_window_margin(z):
    """largest minus second-largest value of every 2x2 window: how close the arg-max decision of relu_pool_fwd is
    to a tie (the routing of the max-pool gradient is discontinuous there; tests use this to tell a rounding-level
    tie-break difference from an error)."""
    B, H, W, C = z.shape
    hp, wp = H // 2, W // 2
    win = z[:, :2 * hp, :2 * wp, :].reshape(B, hp, 2, wp, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, hp, wp, C, 4)
    part = np.partition(win, 2, axis=-1)
    return part[..., 3] - part[..., 2]


def relu_pool_bwd(dp, pooled, amax, H, W, active=None):
    """route dp to the window arg-max where the pooled (post-relu) value is > 0 (`active`: tests only -- another
    implementation's ReLU decision for windows whose maximum is zero to rounding, see Net.relu_override)."""
    B, hp, wp, C = dp.shape
    g = np.where(pooled > 0 if active is None else active, dp, 0)
    win = np.zeros((B, hp, wp, C, 4), dtype=dp.dtype)
    np.put_along_axis(win, amax[..., None].astype(np.int64), g[..., None], axis=-1)
    dz = np.zeros((B, H, W, C), dtype=dp.dtype)
    dz[:, :2 * hp, :2 * wp, :] = win.reshape(B, hp, wp, C, 2, 2).transpose(
        0, 1, 4, 2, 5, 3).reshape(B, 2 * hp, 2 * wp, C)
    return dz


def _act(z, kind):
    if kind == "relu":
        return np.maximum(z, 0)
    if kind == "tanh":
        return np.tanh(z)
    return z


def _act_bwd(dy, y, kind):
    if kind == "relu":
        return np.where(y > 0, dy, 0)
    if kind == "tanh":
        return dy * (1 - y * y)
    return dy


# --------------------------------------------------------------------------------------
# network forward / backward
# --------------------------------------------------------------------------------------
class Net(object):
    def __init__(self, spec, flat_params, dt):
        self.spec, self.dt = spec, dt
        self.p = unflatten(spec, flat_params, dt)
        self.amax_override = None     # tests only: {conv name: arg-max codes} to route the pool gradient with
        self.relu_override = None     # tests only: {conv name: bool (B,h,w,C)} which pooled outputs the ReLU lets gradient through
        self.drop_masks = None        # dropout keep masks {layer name: (B, units) of 0/1} for the next training forward

    def flat(self):
        return flatten(self.spec, self.p, self.dt)

    def forward(self, state, action=None, white=None, training=True):
        """state: (B, ...) any float dtype (f16 from replay upcasts exactly).  `white`:
        optional precomputed (scale, shift).  `training` is base_network.IS_TRAINING (only batch norm looks at it).
        Returns a cache dict; cache['out'] is (B, A) actions (actor) or (B, 1) q-values (critic)."""
        sp, dt = self.spec, self.dt
        B = state.shape[0]
        c = {"B": B}
        if sp.pixel:
            x = np.asarray(state).reshape(B, sp.H, sp.W, sp.C)
            if white is None:
                white = whiten_stats(x, dt)
            x = (x.astype(dt) * white[0] + white[1]).astype(dt)
            c["white"] = white
            for (name, _k, _co), (h, w) in zip(CONV_DEFS, sp.conv_hw):
                if sp.batch_norm:      # conv (no bias) -> (z - mean) * inv + beta -> relu -> pool
                    z = conv_fwd(x, self.p[name + "/weights"], np.zeros(self.p[name + "/biases"].shape, dt))
                    mean, inv = bn_stats(z, dt, training)
                    zhat = ((z - mean) * inv).astype(dt)
                    c[name + ":bn"] = (zhat, inv, bool(training))
                    z = (zhat + self.p[name + "/biases"]).astype(dt)
                else:
                    z = conv_fwd(x, self.p[name + "/weights"], self.p[name + "/biases"])
                pooled, amax, zmax = relu_pool_fwd(z, with_zmax=True)
                c[name + ":zmax"] = zmax              # pre-ReLU window maximum (tests: tell ReLU-boundary flips from errors)
                c[name + ":margin"] = pool_window_margin(z)
                c[name + ":amax_own"] = amax          # this forward's own arg-max (tests: tell near-tie flips from errors)
                if self.amax_override is not None and name in self.amax_override:
                    amax = np.asarray(self.amax_override[name]).astype(np.uint8).reshape(amax.shape)
                c[name] = (x, pooled, amax, h, w)
                x = pooled
            c["pool_shape"] = x.shape
            h = x.reshape(B, -1)
        else:
            h = np.asarray(state).reshape(B, -1).astype(dt)
        c["fc"] = []
        for name, _n_in, _n_out, act, cat in sp.fc:
            if cat:
                h = np.concatenate([h, np.asarray(action, dtype=dt).reshape(B, -1)], axis=1)
            y = _act(h @ self.p[name + "/weights"] + self.p[name + "/biases"], act)
            if sp.dropout and training and name.startswith("h") and name[1:].isdigit():
                # slim.dropout: x * keep / keep_prob.  The random bits are the caller's (no RNG is shared with TF)
                y = (y * np.asarray(self.drop_masks[name], dt) * dt(2.0)).astype(dt)
                c.setdefault("dropped", set()).add(name)
            c["fc"].append((h, y))
            h = y
        c["out"] = h
        return c

    def backward_trunk(self, c, dp):
        """conv trunk backward from the gradient w.r.t. pool3 (B, h, w, 10): {name: grad} of the conv variables."""
        sp, dt = self.spec, self.dt
        g = collections.OrderedDict()
        for idx in range(len(CONV_DEFS) - 1, -1, -1):
            name = CONV_DEFS[idx][0]
            x, pooled, amax, h, w = c[name]
            active = None
            if self.relu_override is not None and name in self.relu_override:
                active = np.asarray(self.relu_override[name], bool).reshape(pooled.shape)
            dz = relu_pool_bwd(dp, pooled, amax, h, w, active)
            if sp.batch_norm:
                zhat, inv, training = c[name + ":bn"]
                dbeta = dz.sum(axis=(0, 1, 2))
                if training:       # through the batch moments
                    m1 = dz.mean(axis=(0, 1, 2), dtype=np.float64).astype(dt)
                    m2 = (dz * zhat).mean(axis=(0, 1, 2), dtype=np.float64).astype(dt)
                    dz = (inv * (dz - m1 - zhat * m2)).astype(dt)
                else:
                    dz = (inv * dz).astype(dt)
                dW, _db, dp = conv_bwd(x, self.p[name + "/weights"], dz, need_dx=idx > 0)
                db = dbeta
            else:
                dW, db, dp = conv_bwd(x, self.p[name + "/weights"], dz, need_dx=idx > 0)
            g[name + "/weights"], g[name + "/biases"] = dW, db
        return g

    def backward(self, c, dout, params=True):
        """Returns (grads OrderedDict in layout order or None, d_action or None)."""
        sp, dt = self.spec, self.dt
        g = collections.OrderedDict()
        d_action = None
        dh = np.asarray(dout, dtype=dt)
        for (name, n_in, _n_out, act, cat), (h, y) in reversed(list(zip(sp.fc, c["fc"]))):
            dz = _act_bwd(dh, y, act)
            if name in c.get("dropped", ()):      # y > 0 <=> kept and active; the kept units carry the factor 1/keep_prob
                dz = (dz * dt(2.0)).astype(dt)
            if params:
                g[name + "/biases"] = dz.sum(axis=0)
                g[name + "/weights"] = h.T @ dz
            dh = dz @ self.p[name + "/weights"].T
            if cat:
                d_action = dh[:, n_in - sp.action_dim:]
                dh = dh[:, :n_in - sp.action_dim]
                if not params:
                    return None, d_action
        if not params:
            return None, d_action
        if sp.pixel:
            g.update(self.backward_trunk(c, dh.reshape(c["pool_shape"])))
        ordered = collections.OrderedDict((n, g[n]) for n, _s in sp.layout())
        return ordered, d_action


# --------------------------------------------------------------------------------------
# optimiser pieces
# --------------------------------------------------------------------------------------
def clip_by_global_norm(flat_grads, clip, dt):
    """util.py:47-50 / tf.clip_by_global_norm: g * clip * min(1/norm, 1/clip)."""
    g = np.asarray(flat_grads, dtype=dt)
    norm = np.sqrt((g * g).sum(dtype=dt))
    if clip is None:
        return g, norm
    scale = dt(clip) * min(dt(1.0) / norm if norm > 0 else dt(np.inf), dt(1.0) / dt(clip))
    return (g * dt(scale)).astype(dt), norm


def soft_update(target_flat, source_flat, coeff, dt):
    """base_network.py:31: target.assign_sub(coeff * (target - source))."""
    t = np.asarray(target_flat, dtype=dt)
    s = np.asarray(source_flat, dtype=dt)
    return (t - dt(coeff) * (t - s)).astype(dt)


# --------------------------------------------------------------------------------------
# DDPG
# --------------------------------------------------------------------------------------
class DDPG(object):
    """The four networks of ddpg_cartpole.py:270-273 + the train ops of :102-119, :186-218."""

    def __init__(self, actor_spec, critic_spec, actor_flat, critic_flat, dt=np.float64,
                 hyper=DEFAULT_HYPER):
        self.dt, self.hp = dt, hyper
        self.actor = Net(actor_spec, actor_flat, dt)
        self.critic = Net(critic_spec, critic_flat, dt)
        # set_as_target_network_for: t - 1.0*(t - s)  (base_network.py:39; appendix B5).
        # The targets' own initial values are whatever the initialiser gave them; with
        # coeff 1.0 the result is s up to one rounding, so start from zeros here.
        self.target_actor = Net(actor_spec, soft_update(
            np.zeros_like(actor_flat), actor_flat, 1.0, dt), dt)
        self.target_critic = Net(critic_spec, soft_update(
            np.zeros_like(critic_flat), critic_flat, 1.0, dt), dt)

    def set_targets(self, target_actor_flat, target_critic_flat):
        self.target_actor = Net(self.actor.spec, target_actor_flat, self.dt)
        self.target_critic = Net(self.critic.spec, target_critic_flat, self.dt)

    # ddpg_cartpole.py:121-125 (noise is added outside, :133-134)
    def action_given(self, state):
        return self.actor.forward(np.asarray(state)[None], training=False)["out"]       # IS_TRAINING: False (:125)

    def _white(self, net, s):
        if not net.spec.pixel:
            return None
        sp = net.spec
        return whiten_stats(np.asarray(s).reshape(-1, sp.H, sp.W, sp.C), self.dt)

    def actor_gradients(self, s1):
        """ddpg_cartpole.py:111-113 + :220-222.  Returns dict with actions, q, dq_da, grads."""
        w1 = self._white(self.actor, s1)
        ca = self.actor.forward(s1, white=w1)
        cc = self.critic.forward(s1, action=ca["out"], white=w1)
        ones = np.ones_like(cc["out"])
        _, dq_da = self.critic.backward(cc, ones, params=False)       # d(sum_b Q)/da
        grads, _ = self.actor.backward(ca, -dq_da)                     # tf.neg(...) as grad_ys
        return {"actions": ca["out"], "q": cc["out"], "dq_da": dq_da,
                "grads": flatten(self.actor.spec, grads, self.dt), "cache_actor": ca}

    def critic_gradients(self, batch, training=True, td_override=None):
        """ddpg_cartpole.py:199-214.  batch = (s1, a, r, mask, s2).  IS_TRAINING is one placeholder for the whole
        graph: in critic.train (:237) the target networks run in training mode as well.  td_override (tests only): back-propagate
        THESE temporal differences instead of this forward pass's own -- the gradients are linear in TD, and where sum_b td_b
        cancels (correlated minibatches) another implementation's 5e-6 on TD is 5e-5 of a gradient; with its TD fed in, what
        remains is the backward arithmetic alone."""
        s1, a, r, mask, s2 = batch
        dt = self.dt
        w2 = self._white(self.target_actor, s2)
        ta = self.target_actor.forward(s2, white=w2, training=training)
        tq = self.target_critic.forward(s2, action=ta["out"], white=w2, training=training)
        y = np.asarray(r, dt) + np.asarray(mask, dt) * dt(self.hp.discount) * tq["out"]
        cb = self.critic.forward(s1, action=np.asarray(a, dt), training=training)
        td = cb["out"] - y
        B = td.shape[0]
        loss = (td * td).mean(dtype=dt)
        td_back = td if td_override is None else np.asarray(td_override, dt).reshape(td.shape)
        grads, _ = self.critic.backward(cb, (dt(2.0) * td_back / dt(B)).astype(dt))
        return {"q": cb["out"], "td": td, "loss": loss, "target_q": tq["out"],
                "target_actions": ta["out"], "cache_critic": cb,
                "grads": flatten(self.critic.spec, grads, self.dt)}

    def check_loss(self, batch):      # ddpg_cartpole.py:239-248 (IS_TRAINING: False)
        out = self.critic_gradients(batch, training=False)
        return out["loss"], out["td"], out["q"]

    def train_minibatch(self, batch):
        """One pass of the loop body ddpg_cartpole.py:331-334: actor.train(s1) then
        critic.train(batch).  The critic step never reads the live actor and the actor step
        never writes the critic, so both gradient sets come from the same snapshot."""
        dt, hp = self.dt, self.hp
        ag = self.actor_gradients(batch[0])
        cg = self.critic_gradients(batch)
        a_clip, a_norm = clip_by_global_norm(ag["grads"], hp.gradient_clip, dt)
        c_clip, c_norm = clip_by_global_norm(cg["grads"], hp.gradient_clip, dt)
        new_a = (self.actor.flat() - dt(hp.actor_lr) * a_clip).astype(dt)
        new_c = (self.critic.flat() - dt(hp.critic_lr) * c_clip).astype(dt)
        self.actor = Net(self.actor.spec, new_a, dt)
        self.critic = Net(self.critic.spec, new_c, dt)
        return {"actions": ag["actions"], "q_actor": ag["q"], "dq_da": ag["dq_da"],
                "actor_grads": ag["grads"], "actor_norm": a_norm,
                "q": cg["q"], "td": cg["td"], "loss": cg["loss"], "target_q": cg["target_q"],
                "critic_grads": cg["grads"], "critic_norm": c_norm}

    def update_targets(self):          # ddpg_cartpole.py:336-337
        tau = self.hp.target_update_rate
        self.target_actor = Net(self.actor.spec, soft_update(
            self.target_actor.flat(), self.actor.flat(), tau, self.dt), self.dt)
        self.target_critic = Net(self.critic.spec, soft_update(
            self.target_critic.flat(), self.critic.flat(), tau, self.dt), self.dt)

    def train_step(self, batches):
        """ddpg_cartpole.py:331-337: `batches_per_step` minibatches, then both target updates."""
        outs = [self.train_minibatch(b) for b in batches]
        self.update_targets()
        return outs


# --------------------------------------------------------------------------------------
# exploration noise (util.py:134-156) -- host side, f64 like numpy
# --------------------------------------------------------------------------------------
class OUNoise(object):
    def __init__(self, dim, theta=0.01, sigma=0.2, max_magnitude=1.5, rng=None):
        self.theta, self.sigma, self.max_magnitude = theta, sigma, max_magnitude
        self.state = np.zeros(dim)
        self.rng = rng or np.random

    def sample(self):
        self.state += self.theta * -self.state
        self.state += self.sigma * self.rng.randn(len(self.state))
        # util.py:155 np.clip(max, -max, state) == minimum(max, state): upper bound only
        self.state = np.minimum(self.max_magnitude, self.state)
        return np.copy(self.state)


def synthetic_batch(rng, B, state_shape, action_dim, pixel):
    """Seeded minibatch shaped like replay output: f16 states k/255, a~U(-1,1),
    reward 1, terminal w.p. 1/50."""
    if pixel:
        mk = lambda: (rng.integers(0, 256, (B,) + tuple(state_shape)).astype(np.float16)
                      / np.float16(255))
    else:
        mk = lambda: rng.standard_normal((B,) + tuple(state_shape)).astype(np.float16)
    s1, s2 = mk(), mk()
    a = rng.uniform(-1, 1, (B, action_dim)).astype(np.float32)
    r = np.ones((B, 1), np.float32)
    mask = (rng.uniform(size=(B, 1)) >= 1.0 / 50).astype(np.float32)
    return s1, a, r, mask, s2
