"""A second, independent witness for the (unpinned) network oracle: the WHOLE inner train step of
ddpg_cartpole.py:329-337 -- whiten -> conv trunk -> heads -> dQ/da -> actor update, TD target -> critic
update (each: tf.gradients, clip_by_global_norm 5, SGD), twice, then both target soft updates -- written
with torch-CPU float64 autograd in the reference's LITERAL order (actor.train(s1) is applied before
critic.train(batch) runs, five trunk forwards per minibatch, nothing deduplicated), and compared with the
committed golden vectors tests/golden/ddpg_step_*.npz (outputs of oracle/ddpg_np.py, whose train_minibatch
takes both gradient sets from one parameter snapshot).  Agreement shows that (1) the hand-derived backward
passes, the clip and the update rules of the oracle compose to the same step as autograd + the formulas of
util.py:47-50 / base_network.py:31, and (2) the oracle's "same snapshot" reordering is exact.

This is NOT the reference (Python 2 + TensorFlow 0.x cannot run here): parity stays "unpinned".
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ddpg_np as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {
    "pixel_8x8x6_B4": dict(shape=(8, 8, 3, 1, 2), pixel=True),
    "pixel_12x10x9_B3": dict(shape=(12, 10, 3, 1, 3), pixel=True),
    "lowdim_28_B5": dict(shape=(2, 2, 7), pixel=False),
}
ACTOR_LR, CRITIC_LR, DISCOUNT, CLIP, TAU = 1e-3, 1e-2, 0.99, 5.0, 1e-4   # ddpg_cartpole.py:35,41-43; util.py:11


def _layout(kind, pixel, shape, A=2):
    """variable list in TF creation order, written out here independently of oracle.NetSpec."""
    out = []
    if pixel:
        H, W, cin = shape[0], shape[1], int(np.prod(shape[2:]))
        for name, k in (("conv1", 5), ("conv2", 5), ("conv3", 3)):
            out += [(name + "/weights", (k, k, cin, 10)), (name + "/biases", (10,))]
            cin, H, W = 10, H // 2, W // 2
        n = H * W * 10
    else:
        n = int(np.prod(shape))
    if kind == "actor":
        fcs = [("h0", n, 100), ("h1", 100, 100), ("h2", 100, 50), ("output_action", 50, A)]
    elif pixel:
        fcs = [("hidden1", n, 200), ("hidden2", 200, 50), ("hidden3", 50 + A, 50), ("q_value", 50, 1)]
    else:
        fcs = [("h0", n + A, 100), ("h1", 100, 100), ("h2", 100, 50), ("q_value", 50, 1)]
    for name, i, o in fcs:
        out += [(name + "/weights", (i, o)), (name + "/biases", (o,))]
    return out


class TNet(object):
    def __init__(self, kind, pixel, shape, flat):
        self.kind, self.pixel, self.shape = kind, pixel, shape
        self.p, off = {}, 0
        for name, shp in _layout(kind, pixel, shape):
            n = int(np.prod(shp))
            self.p[name] = torch.tensor(np.asarray(flat[off:off + n], np.float64).reshape(shp), requires_grad=True)
            off += n
        assert off == len(flat)

    def params(self):
        return list(self.p.values())

    def flat(self):
        return np.concatenate([v.detach().numpy().ravel() for v in self.p.values()])

    def trunk(self, state):
        B = state.shape[0]
        if not self.pixel:
            return state.reshape(B, -1)
        H, W, C = self.shape[0], self.shape[1], int(np.prod(self.shape[2:]))
        x = state.reshape(B, H, W, C)                                     # base_network.py:88-90
        mean = x.mean(dim=(0, 1, 2))                                      # tf.nn.moments (:95-96)
        var = (x * x).mean(dim=(0, 1, 2)) - mean * mean
        inv = torch.rsqrt(var + 1e-6)
        x = (x * inv - mean * inv).permute(0, 3, 1, 2)                    # tf.nn.batch_normalization (:97-99)
        for name, k in (("conv1", 5), ("conv2", 5), ("conv3", 3)):        # slim.conv2d + max_pool2d (:103-123)
            x = F.max_pool2d(F.relu(F.conv2d(x, self.p[name + "/weights"].permute(3, 2, 0, 1),
                                             self.p[name + "/biases"], padding=k // 2)), 2)
        return x.permute(0, 2, 3, 1).reshape(B, -1)                       # slim.flatten, NHWC order (:133)

    def fc(self, name, h, act):
        y = h @ self.p[name + "/weights"] + self.p[name + "/biases"]
        return {"relu": F.relu, "tanh": torch.tanh, None: lambda t: t}[act](y)

    def actor(self, state):
        h = self.trunk(state)
        for n in ("h0", "h1", "h2"):
            h = self.fc(n, h, "relu")
        return self.fc("output_action", h, "tanh")                        # ddpg_cartpole.py:95-100

    def critic(self, state, action):
        h = self.trunk(state)
        if self.pixel:                                                    # :166-171 (intent)
            h = self.fc("hidden2", self.fc("hidden1", h, "relu"), "relu")
            h = self.fc("hidden3", torch.cat([h, action], dim=1), "relu")
        else:                                                             # :172-177 (B1)
            h = torch.cat([h, action], dim=1)
            for n in ("h0", "h1", "h2"):
                h = self.fc(n, h, "relu")
        return self.fc("q_value", h, None)                                # :180-184


def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm (util.py:47-50): g * clip / max(norm, clip)."""
    norm = torch.sqrt(sum((g * g).sum() for g in grads))
    return [g * clip / torch.maximum(norm, torch.tensor(clip, dtype=torch.float64)) for g in grads], norm


def sgd(net, grads, lr):
    with torch.no_grad():
        for v, g in zip(net.params(), grads):
            v -= lr * g


@pytest.mark.parametrize("name", sorted(CASES))
def test_torch_autograd_step_in_reference_order_reproduces_the_golden_step(name):
    torch.set_default_dtype(torch.float64)
    case, g = CASES[name], np.load(os.path.join(GOLDEN, "ddpg_step_%s.npz" % name))
    pixel, shape = case["pixel"], case["shape"]
    actor, critic = TNet("actor", pixel, shape, g["actor"]), TNet("critic", pixel, shape, g["critic"])
    tactor, tcritic = TNet("actor", pixel, shape, g["target_actor"]), TNet("critic", pixel, shape, g["target_critic"])
    T = lambda a: torch.tensor(np.asarray(a, np.float64))
    for i in range(2):                                                     # batches_per_step = 2 in the fixtures
        s1, a, r, mask, s2 = (T(g["b%d_%s" % (i, k)]) for k in ("s1", "a", "r", "mask", "s2"))
        # ---- actor.train(batch.state_1) (ddpg_cartpole.py:140-145; graph :102-119)
        act = actor.actor(s1)
        a_in = act.detach().clone().requires_grad_(True)                   # stop_gradient(actor.output_action) (:161-162)
        q_of_actor = critic.critic(s1, a_in)
        dq_da, = torch.autograd.grad(q_of_actor.sum(), a_in)               # tf.gradients(q_value, input_action) (:222)
        a_grads = torch.autograd.grad(act, actor.params(), grad_outputs=-dq_da)      # grad_ys = tf.neg(...) (:111-113)
        a_flat = np.concatenate([x.numpy().ravel() for x in a_grads])
        a_clipped, _ = clip_by_global_norm(a_grads, CLIP)
        np.testing.assert_allclose(act.detach().numpy(), g["o%d_actions" % i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(q_of_actor.detach().numpy(), g["o%d_q_actor" % i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(dq_da.numpy(), g["o%d_dq_da" % i], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(a_flat, g["o%d_actor_grads" % i], rtol=1e-7, atol=1e-11)
        sgd(actor, a_clipped, ACTOR_LR)                                    # applied BEFORE critic.train runs, as in the reference
        # ---- critic.train(batch) (:230-237; graph :186-218)
        with torch.no_grad():
            y = r + mask * DISCOUNT * tcritic.critic(s2, tactor.actor(s2))          # bellman target (:199-202)
        q = critic.critic(s1, a)
        td = q - y
        loss = (td ** 2).mean()                                            # :208-209
        c_grads = torch.autograd.grad(loss, critic.params())
        c_flat = np.concatenate([x.numpy().ravel() for x in c_grads])
        c_clipped, _ = clip_by_global_norm(c_grads, CLIP)
        np.testing.assert_allclose(q.detach().numpy(), g["o%d_q" % i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(td.detach().numpy(), g["o%d_td" % i], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(loss.item(), float(g["o%d_loss" % i]), rtol=1e-9)
        np.testing.assert_allclose(c_flat, g["o%d_critic_grads" % i], rtol=1e-7, atol=1e-11)
        sgd(critic, c_clipped, CRITIC_LR)
    # ---- target_actor.update_weights(); target_critic.update_weights() (:336-337; base_network.py:31)
    with torch.no_grad():
        for t, s in ((tactor, actor), (tcritic, critic)):
            for tv, sv in zip(t.params(), s.params()):
                tv -= TAU * (tv - sv)
    for net, key in ((actor, "new_actor"), (critic, "new_critic"), (tactor, "new_target_actor"), (tcritic, "new_target_critic")):
        np.testing.assert_allclose(net.flat(), g[key], rtol=1e-9, atol=1e-12, err_msg=key)


def test_twin_layout_matches_the_oracle_layout():
    """the twin's own variable list and the oracle's NetSpec.layout() are written independently; they must agree
    (order = TF creation order, the order of the flat buffers that cross the C ABI)."""
    for case in CASES.values():
        shape, pixel = case["shape"], case["pixel"]
        kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:]))) if pixel else \
            dict(pixel=False, state_elems=int(np.prod(shape)))
        for kind in ("actor", "critic"):
            spec = O.NetSpec(kind, 2, [100, 100, 50], **kw)
            assert [(n, tuple(s)) for n, s in spec.layout()] == [(n, tuple(s)) for n, s in _layout(kind, pixel, shape)]


def test_cpu_baseline_torch_restatement_matches_the_oracle():
    """oracle/ddpg_torch.py (bench.py's torch-CPU `cpu_baseline`) computes the same minibatch update as the numpy oracle."""
    from oracle.ddpg_torch import TorchDDPG
    rng = np.random.default_rng(5)
    shape, B = (16, 16, 3, 1, 2), 6
    kw = dict(pixel=True, H=16, W=16, C=6)
    aspec, cspec = O.NetSpec("actor", 2, [100, 100, 50], **kw), O.NetSpec("critic", 2, [], **kw)
    af = O.init_params(aspec, rng) + rng.normal(0, 0.05, aspec.num_params()).astype(np.float32)
    cf = O.init_params(cspec, rng) + rng.normal(0, 0.05, cspec.num_params()).astype(np.float32)
    batch = O.synthetic_batch(rng, B, shape, 2, True)
    ref = O.DDPG(aspec, cspec, af, cf, np.float64)
    ref.set_targets(af, cf)
    out = ref.train_minibatch(batch)
    twin = TorchDDPG(aspec, cspec, af, cf, dtype=torch.float64)
    got = twin.train_minibatch(batch)
    np.testing.assert_allclose(got["loss"], out["loss"], rtol=1e-9)
    np.testing.assert_allclose(got["q"], out["q"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(twin.flat("actor"), ref.actor.flat(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(twin.flat("critic"), ref.critic.flat(), rtol=1e-9, atol=1e-12)
