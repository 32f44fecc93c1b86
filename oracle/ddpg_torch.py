"""torch-CPU restatement of one DDPG minibatch update.  TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Used by bench.py's `cpu_baseline` leg (and nothing in the product): the stand-in for "the reference CPU path"
that SURVEY 8(d) / BASELINE.md define -- TensorFlow 0.x cannot be installed here, so the same step is timed on
torch's CPU kernels (oneDNN convolutions, MKL/OpenBLAS GEMMs, autograd) with all host cores.  It follows the
same call sites as oracle/ddpg_np.py (paths relative to /root/reference):
  whitening ........ base_network.py:95-99      conv/pool trunk ... base_network.py:103-127
  MLP heads ........ base_network.py:58-71, ddpg_cartpole.py:95-100, :166-171, :180-184
  actor train op ... ddpg_cartpole.py:102-119 (grad_ys = -dQ/da, sum over the batch)
  critic train op .. ddpg_cartpole.py:186-218   clip / SGD ........ util.py:45-50, ddpg_cartpole.py:118,213
and is generous to the CPU: float32, the critic's trunk on state_1 is evaluated ONCE for both of its uses
(the reference's two session.run calls evaluate it twice), NCHW / channels_last is left to torch.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ddpg_np as O


class TorchDDPG(object):
    def __init__(self, actor_spec, critic_spec, actor_flat, critic_flat, dtype=torch.float32, hyper=O.DEFAULT_HYPER):
        self.aspec, self.cspec, self.hp, self.dtype = actor_spec, critic_spec, hyper, dtype
        mk = lambda spec, flat: {n: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)
                                 for n, v in O.unflatten(spec, flat, np.float64).items()}
        self.actor, self.critic = mk(actor_spec, actor_flat), mk(critic_spec, critic_flat)
        self.target_actor, self.target_critic = mk(actor_spec, actor_flat), mk(critic_spec, critic_flat)

    def _trunk(self, spec, p, state):
        B = state.shape[0]
        if not spec.pixel:
            return state.reshape(B, -1)
        x = state.reshape(B, spec.H, spec.W, spec.C)
        mean = x.mean(dim=(0, 1, 2))
        var = (x * x).mean(dim=(0, 1, 2)) - mean * mean
        inv = torch.rsqrt(var + O.WHITEN_EPS)
        x = (x * inv - mean * inv).permute(0, 3, 1, 2)
        for name, k, _co in O.CONV_DEFS:
            x = F.max_pool2d(F.relu(F.conv2d(x, p[name + "/weights"].permute(3, 2, 0, 1), p[name + "/biases"],
                                             padding=k // 2)), 2)
        return x.permute(0, 2, 3, 1).reshape(B, -1)

    def _head(self, spec, p, h, action=None):
        for name, _i, _o, act, cat in spec.fc:
            if cat:
                h = torch.cat([h, action], dim=1)
            h = h @ p[name + "/weights"] + p[name + "/biases"]
            h = {"relu": F.relu, "tanh": torch.tanh, "linear": lambda t: t}[act](h)
        return h

    def _apply(self, params, grads, lr):
        norm = torch.sqrt(sum((g * g).sum() for g in grads))
        scale = self.hp.gradient_clip / torch.clamp(norm, min=self.hp.gradient_clip)
        with torch.no_grad():
            for v, g in zip(params.values(), grads):
                v -= lr * scale * g

    def train_minibatch(self, batch):
        s1, a, r, mask, s2 = (torch.as_tensor(np.asarray(x, np.float32)).to(self.dtype) for x in batch)
        act = self._head(self.aspec, self.actor, self._trunk(self.aspec, self.actor, s1))
        feat_c = self._trunk(self.cspec, self.critic, s1)               # shared by both evaluations of critic(s1, .)
        a_in = act.detach().clone().requires_grad_(True)
        q_mu = self._head(self.cspec, self.critic, feat_c, a_in)
        dq_da, = torch.autograd.grad(q_mu.sum(), a_in, retain_graph=True)
        a_grads = torch.autograd.grad(act, list(self.actor.values()), grad_outputs=-dq_da)
        with torch.no_grad():
            ta = self._head(self.aspec, self.target_actor, self._trunk(self.aspec, self.target_actor, s2))
            tq = self._head(self.cspec, self.target_critic, self._trunk(self.cspec, self.target_critic, s2), ta)
            y = r + mask * self.hp.discount * tq
        q = self._head(self.cspec, self.critic, feat_c, a)
        loss = ((q - y) ** 2).mean()
        c_grads = torch.autograd.grad(loss, list(self.critic.values()))
        self._apply(self.actor, a_grads, self.hp.actor_lr)
        self._apply(self.critic, c_grads, self.hp.critic_lr)
        return {"loss": float(loss), "q": q.detach().numpy(), "actions": act.detach().numpy()}

    def flat(self, which):
        return np.concatenate([v.detach().numpy().ravel() for v in getattr(self, which).values()])
