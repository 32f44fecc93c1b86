"""numpy restatement of the reference's NAF agent (naf_cartpole.py).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (no numerical test in the reference; Python 2 + TensorFlow 0.x cannot run here); cross-
checked against torch-CPU autograd in tests/test_oracle_vs_torch.py.

Call sites followed (paths relative to /root/reference):
  ValueNetwork ........................ naf_cartpole.py:93-114  (input_state_network + linear 'fc' -> (B,1))
  mu head ('naf/output_action/fc') ... naf_cartpole.py:149-161 (tanh, W~U(+-1e-3); own trunk unless
                                        --share-input-state-representation, :151-154)
  l_values head ('naf/l_values/fc') ... naf_cartpole.py:174-184 (linear, A(A+1)/2 outputs)
  L rows = [lower, exp(diag), zeros] .. naf_cartpole.py:194-207
  P = L L^T, A = -1/2 d^T P d ......... naf_cartpole.py:210-218
  Q = V + A; y = r + mask*gamma*V'(s2)  naf_cartpole.py:221-227
  loss = mean((Q - y)^2) .............. naf_cartpole.py:230
  optimiser / clip / apply ............ naf_cartpole.py:231-239, util.py:45-50, :73-76
  check_numerics (l_values, L, loss) .. naf_cartpole.py:242-245
  inner step + target update .......... naf_cartpole.py:365-373 (target_value <- value only)
  optimisers (tf.train.*Optimizer) .... GradientDescent; Momentum: accum = m*accum + g, var -= lr*accum;
                                        Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), m,v moments,
                                        var -= lr_t*m/(sqrt(v)+eps)   (TF defaults b1=.9 b2=.999 eps=1e-8)
"""
import collections

import numpy as np

from . import ddpg_np as O


class HeadSpec(O.NetSpec):
    """trunk + hidden stack like the actor, then one 'fc' head (naf_cartpole.py:105-109,156-161,180-184).
    head_only: no trunk / hidden layers -- the input is another network's state representation."""

    def __init__(self, head_out, head_act, hidden, pixel, H=0, W=0, C=0, state_elems=0, head_only=False,
                 batch_norm=False, dropout=False):
        self.head_out, self.head_act, self.head_only = int(head_out), head_act, head_only
        O.NetSpec.__init__(self, "actor", head_out, [] if head_only else hidden, pixel and not head_only,
                           H, W, C, state_elems, batch_norm=batch_norm, dropout=dropout)

    def _fc_layers(self):
        out, n_in = [], self.flat
        for i, h in enumerate(self.hidden):
            out.append(("h%d" % i, n_in, h, "relu", False))
            n_in = h
        out.append(("fc", n_in, self.head_out, self.head_act, False))
        return out


def init_head_params(spec, rng, small_head=False):
    """xavier-uniform / zero biases; the mu head uses U(+-1e-3) (naf_cartpole.py:155)."""
    parts = []
    for name, shape in spec.layout():
        if name.endswith("/biases"):
            parts.append(np.zeros(shape, np.float32))
        elif small_head and name.startswith("fc/"):
            parts.append(rng.uniform(-1e-3, 1e-3, shape).astype(np.float32))
        else:
            fan_in = shape[0] * shape[1] * shape[2] if len(shape) == 4 else shape[0]
            fan_out = shape[0] * shape[1] * shape[3] if len(shape) == 4 else shape[1]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            parts.append(rng.uniform(-lim, lim, shape).astype(np.float32))
    return np.concatenate([p.ravel() for p in parts])


def num_l_values(action_dim):
    return (action_dim * (action_dim + 1)) // 2


def build_L(l_values, action_dim, dt):
    """naf_cartpole.py:194-207: row i = [l[off:off+i], exp(l[off+i]), 0...], off = i(i+1)/2."""
    B = l_values.shape[0]
    L = np.zeros((B, action_dim, action_dim), dtype=dt)
    for i in range(action_dim):
        off = (i * (i + 1)) // 2
        L[:, i, :i] = l_values[:, off:off + i]
        L[:, i, i] = np.exp(l_values[:, off + i])
    return L


Optimiser = collections.namedtuple("Optimiser", "kind learning_rate momentum beta1 beta2 epsilon")


def make_optimiser(name="GradientDescent", args=None):
    """util.construct_optimiser (util.py:73-76) for the three optimisers the exps use."""
    args = dict(args or {"learning_rate": 0.001})
    lr = float(args.get("learning_rate", 0.001))
    if name == "GradientDescent":
        return Optimiser("sgd", lr, 0.0, 0.0, 0.0, 0.0)
    if name == "Momentum":
        return Optimiser("momentum", lr, float(args.get("momentum", 0.9)), 0.0, 0.0, 0.0)
    if name == "Adam":
        return Optimiser("adam", lr, 0.0, float(args.get("beta1", 0.9)), float(args.get("beta2", 0.999)),
                         float(args.get("epsilon", 1e-8)))
    raise ValueError(name)


class NAF(object):
    def __init__(self, value_spec, mu_spec, l_spec, value_flat, mu_flat, l_flat, share, action_dim,
                 dt=np.float64, discount=0.99, gradient_clip=5.0, target_update_rate=1e-4,
                 optimiser=make_optimiser()):
        self.dt, self.share, self.A = dt, share, action_dim
        self.discount, self.clip, self.tau, self.opt = discount, gradient_clip, target_update_rate, optimiser
        self.value = O.Net(value_spec, value_flat, dt)
        self.mu = O.Net(mu_spec, mu_flat, dt)
        self.l = O.Net(l_spec, l_flat, dt)
        self.target_value = O.Net(value_spec, O.soft_update(np.zeros_like(value_flat), value_flat, 1.0, dt), dt)
        n = len(value_flat) + len(mu_flat) + len(l_flat)
        self.m = np.zeros(n, dt)
        self.v = np.zeros(n, dt)
        self.t = 0

    def flat(self):
        return np.concatenate([self.value.flat(), self.mu.flat(), self.l.flat()])

    def _white(self, net, s):
        sp = net.spec
        if not sp.pixel:
            return None
        return O.whiten_stats(np.asarray(s).reshape(-1, sp.H, sp.W, sp.C), self.dt)

    def _forward(self, s1, training=True):
        w1 = self._white(self.value, s1)
        cv = self.value.forward(s1, white=w1, training=training)
        if self.share:
            rep = cv["fc"][-1][0]          # input of value's 'fc' head = input_state_representation
            cm, cl = self.mu.forward(rep), self.l.forward(rep)
        else:
            cm, cl = self.mu.forward(s1, white=w1, training=training), self.l.forward(s1, white=w1, training=training)
        return cv, cm, cl

    def action_given(self, state):
        return self._forward(np.asarray(state)[None], training=False)[1]["out"]      # IS_TRAINING: False (:253)

    def forward_backward(self, batch, backward=True):
        s1, a, r, mask, s2 = batch
        dt, A = self.dt, self.A
        B = np.asarray(a).shape[0]
        # naf_cartpole.py:271 feeds IS_TRAINING True to the train op (whole graph, target network included); the debug
        # fetch of :282 feeds False
        cv, cm, cl = self._forward(s1, training=backward)
        V, mu, lv = cv["out"], cm["out"], cl["out"]
        L = build_L(lv, A, dt)
        d = np.asarray(a, dt) - mu                                   # (B, A)
        z = np.einsum("bij,bi->bj", L, d)                            # L^T d
        adv = (-0.5 * (z * z).sum(axis=1, keepdims=True)).astype(dt)  # -1/2 d^T L L^T d
        q = V + adv
        tv = self.target_value.forward(s2, training=backward)["out"]
        y = np.asarray(r, dt) + np.asarray(mask, dt) * dt(self.discount) * tv
        td = q - y
        loss = (td * td).mean(dtype=dt)
        out = {"l_values": lv, "loss": loss, "value": V, "advantage": adv, "target_value": tv, "q": q,
               "mu": mu, "td": td,
               "finite": bool(np.isfinite(lv).all() and np.isfinite(L).all() and np.isfinite(loss))}
        if not backward:
            return out
        dq = (dt(2.0) * td / dt(B)).astype(dt)
        dz = -z * dq                                                 # dA/dz = -z
        dL = np.einsum("bi,bj->bij", d, dz)                          # dL[i][j] = d_i dz_j
        dd = np.einsum("bij,bj->bi", L, dz)
        dl = np.zeros_like(lv)
        for i in range(A):
            off = (i * (i + 1)) // 2
            dl[:, off:off + i] = dL[:, i, :i]
            dl[:, off + i] = dL[:, i, i] * L[:, i, i]                # through exp
        gl, drep_l = self._backward_head(self.l, cl, dl)
        gm, drep_m = self._backward_head(self.mu, cm, -dd)
        if self.share:
            gv = self._backward_value(cv, dq, extra=drep_l + drep_m)
        else:
            gv = self._backward_value(cv, dq, extra=None)
        out["grads"] = np.concatenate([gv, gm, gl])
        return out

    def _backward_head(self, net, cache, dout):
        """full backward of a head network; for head-only nets also returns d(representation)."""
        if net.spec.head_only:
            (name, _i, _o, act, _c), (h, yv) = net.spec.fc[0], cache["fc"][0]
            dz = O._act_bwd(np.asarray(dout, self.dt), yv, act)
            g = collections.OrderedDict()
            g[name + "/weights"] = h.T @ dz
            g[name + "/biases"] = dz.sum(axis=0)
            return O.flatten(net.spec, g, self.dt), dz @ net.p[name + "/weights"].T
        grads, _ = net.backward(cache, dout)
        return O.flatten(net.spec, grads, self.dt), None

    def _backward_value(self, cv, dq, extra):
        net = self.value
        if extra is None:
            grads, _ = net.backward(cv, dq)
            return O.flatten(net.spec, grads, self.dt)
        # shared representation: the value head's own d(rep) plus the two NAF heads' (naf_cartpole.py:151-152)
        name = net.spec.fc[-1][0]
        h, _y = cv["fc"][-1]
        g_head = {name + "/weights": h.T @ dq, name + "/biases": dq.sum(axis=0)}
        drep = dq @ net.p[name + "/weights"].T + extra
        sub = dict(cv)
        sub["fc"] = cv["fc"][:-1]          # hidden stack only; drep is w.r.t. its post-activation output
        grads, _ = self._backward_trunc(net, sub, drep)
        grads.update(g_head)
        return O.flatten(net.spec, grads, self.dt)

    def _backward_trunc(self, net, c, drep):
        """backward through hidden stack + trunk given d(loss)/d(last hidden activation)."""
        sp, dt = net.spec, self.dt
        g = collections.OrderedDict()
        dh = np.asarray(drep, dt)
        for (name, _n_in, _n_out, act, _cat), (h, yv) in reversed(list(zip(sp.fc, c["fc"]))):
            dz = O._act_bwd(dh, yv, act)
            if name in c.get("dropped", ()):
                dz = (dz * dt(2.0)).astype(dt)
            g[name + "/biases"] = dz.sum(axis=0)
            g[name + "/weights"] = h.T @ dz
            dh = dz @ net.p[name + "/weights"].T
        if sp.pixel:
            g.update(net.backward_trunk(c, dh.reshape(c["pool_shape"])))
        return g, None

    def apply(self, grads):
        dt, o = self.dt, self.opt
        g, norm = O.clip_by_global_norm(grads, self.clip, dt)
        flat = self.flat()
        self.t += 1
        if o.kind == "sgd":
            flat = flat - dt(o.learning_rate) * g
        elif o.kind == "momentum":
            self.m = dt(o.momentum) * self.m + g
            flat = flat - dt(o.learning_rate) * self.m
        else:
            lr_t = o.learning_rate * np.sqrt(1.0 - o.beta2 ** self.t) / (1.0 - o.beta1 ** self.t)
            self.m = dt(o.beta1) * self.m + dt(1.0 - o.beta1) * g
            self.v = dt(o.beta2) * self.v + dt(1.0 - o.beta2) * g * g
            flat = flat - dt(lr_t) * self.m / (np.sqrt(self.v) + dt(o.epsilon))
        flat = flat.astype(dt)
        nv, nm = self.value.spec.num_params(), self.mu.spec.num_params()
        self.value = O.Net(self.value.spec, flat[:nv], dt)
        self.mu = O.Net(self.mu.spec, flat[nv:nv + nm], dt)
        self.l = O.Net(self.l.spec, flat[nv + nm:], dt)
        return norm

    def train(self, batch):                     # naf_cartpole.py:264-272
        out = self.forward_backward(batch)
        if not out["finite"]:
            raise FloatingPointError("check_numerics")
        out["norm"] = self.apply(out["grads"])
        return out

    def update_targets(self):                   # naf_cartpole.py:373
        self.target_value = O.Net(self.value.spec, O.soft_update(
            self.target_value.flat(), self.value.flat(), self.tau, self.dt), self.dt)

    def train_step(self, batches):              # naf_cartpole.py:367-373
        outs = [self.train(b) for b in batches]
        self.update_targets()
        return outs
