"""Cross-check of oracle/ddpg_np.py against torch-CPU autograd (an independent implementation;
NOT the reference -- the reference needs Python 2 + TensorFlow 0.x and cannot run here)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ddpg_np as O

torch.set_default_dtype(torch.float64)


def torch_forward(spec, p, state, action=None, training=True):
    B = state.shape[0]
    if spec.pixel:
        x = state.reshape(B, spec.H, spec.W, spec.C)
        mean = x.mean(dim=(0, 1, 2))
        var = (x * x).mean(dim=(0, 1, 2)) - mean * mean
        inv = torch.rsqrt(var + 1e-6)
        x = (x * inv - mean * inv).permute(0, 3, 1, 2)
        for name, k, _co in O.CONV_DEFS:
            w = p[name + "/weights"].permute(3, 2, 0, 1)
            if spec.batch_norm:     # slim.batch_norm: decay .999, center, no scale, eps 1e-3; moving stats stay (0, 1)
                z = F.conv2d(x, w, None, padding=k // 2)
                co = z.shape[1]
                x = F.relu(F.batch_norm(z, torch.zeros(co), torch.ones(co), weight=None, bias=p[name + "/biases"],
                                        training=training, momentum=0.0, eps=1e-3))
            else:
                x = F.relu(F.conv2d(x, w, p[name + "/biases"], padding=k // 2))
            x = F.max_pool2d(x, 2)
        h = x.permute(0, 2, 3, 1).reshape(B, -1)
    else:
        h = state.reshape(B, -1)
    for name, _i, _o, act, cat in spec.fc:
        if cat:
            h = torch.cat([h, action], dim=1)
        h = h @ p[name + "/weights"] + p[name + "/biases"]
        h = {"relu": F.relu, "tanh": torch.tanh, "linear": lambda t: t}[act](h)
    return h


def tparams(spec, flat):
    return {n: torch.tensor(v, requires_grad=True) for n, v in O.unflatten(spec, flat, np.float64).items()}


def tflat(spec, p, grads):
    return np.concatenate([g.detach().numpy().ravel() for g in grads])


CASES = [
    dict(B=4, shape=(8, 8, 3, 1, 2), pixel=True),      # 8x8x6
    dict(B=3, shape=(12, 10, 3, 1, 3), pixel=True),    # 12x10x9, odd pooling 10->5->2->1
    dict(B=5, shape=(2, 2, 7), pixel=False),           # cfg1 low-dim pose state
    dict(B=4, shape=(8, 8, 3, 1, 2), pixel=True, batch_norm=True),     # --use-batch-norm
    dict(B=3, shape=(16, 12, 3, 1, 3), pixel=True, batch_norm=True),
]


@pytest.mark.parametrize("case", CASES)
def test_ddpg_gradients_match_autograd(case):
    rng = np.random.default_rng(7)
    B, shape, pixel = case["B"], case["shape"], case["pixel"]
    if pixel:
        kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:])), batch_norm=case.get("batch_norm", False))
    else:
        kw = dict(pixel=False, state_elems=int(np.prod(shape)))
    aspec = O.NetSpec("actor", 2, [100, 100, 50], **kw)
    cspec = O.NetSpec("critic", 2, [100, 100, 50], **kw)
    af, cf = O.init_params(aspec, rng), O.init_params(cspec, rng)
    # make the actor head non-degenerate and biases non-zero so every path carries signal
    af = af + rng.normal(0, 0.05, af.shape).astype(np.float32)
    cf = cf + rng.normal(0, 0.05, cf.shape).astype(np.float32)
    taf = af + rng.normal(0, 0.01, af.shape).astype(np.float32)
    tcf = cf + rng.normal(0, 0.01, cf.shape).astype(np.float32)
    batch = O.synthetic_batch(rng, B, shape, 2, pixel)
    s1, a, r, mask, s2 = batch

    agent = O.DDPG(aspec, cspec, af, cf, np.float64)
    agent.set_targets(taf, tcf)
    ag = agent.actor_gradients(s1)
    cg = agent.critic_gradients(batch)

    # --- torch: actor direction (ddpg_cartpole.py:111-113, :222)
    pa, pc = tparams(aspec, af), tparams(cspec, cf)
    ts1 = torch.tensor(s1.astype(np.float64))
    ts2 = torch.tensor(s2.astype(np.float64))
    act = torch_forward(aspec, pa, ts1)
    a_in = act.detach().clone().requires_grad_(True)          # stop_gradient (:162)
    q = torch_forward(cspec, pc, ts1, a_in)
    dq_da, = torch.autograd.grad(q.sum(), a_in)
    ga = torch.autograd.grad(act, list(pa.values()), grad_outputs=-dq_da)
    np.testing.assert_allclose(ag["actions"], act.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ag["q"], q.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ag["dq_da"], dq_da.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ag["grads"], tflat(aspec, pa, ga), rtol=1e-8, atol=1e-11)

    # --- torch: critic TD loss (ddpg_cartpole.py:199-214)
    pta, ptc = tparams(aspec, taf), tparams(cspec, tcf)
    with torch.no_grad():
        tq = torch_forward(cspec, ptc, ts2, torch_forward(aspec, pta, ts2))
        y = torch.tensor(r.astype(np.float64)) + torch.tensor(mask.astype(np.float64)) * 0.99 * tq
    qb = torch_forward(cspec, pc, ts1, torch.tensor(a.astype(np.float64)))
    loss = ((qb - y) ** 2).mean()
    gc = torch.autograd.grad(loss, list(pc.values()))
    np.testing.assert_allclose(cg["loss"], loss.item(), rtol=1e-10)
    np.testing.assert_allclose(cg["td"], (qb - y).detach().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cg["grads"], tflat(cspec, pc, gc), rtol=1e-8, atol=1e-11)

    # --- inference mode (IS_TRAINING False: action_given / check_loss): moving statistics, never updated
    with torch.no_grad():
        a_inf = torch_forward(aspec, pa, ts1[:1], training=False)
        tq_inf = torch_forward(cspec, ptc, ts2, torch_forward(aspec, pta, ts2, training=False), training=False)
        y_inf = torch.tensor(r.astype(np.float64)) + torch.tensor(mask.astype(np.float64)) * 0.99 * tq_inf
        q_inf = torch_forward(cspec, pc, ts1, torch.tensor(a.astype(np.float64)), training=False)
    np.testing.assert_allclose(agent.action_given(s1[0]), a_inf.numpy(), rtol=1e-10, atol=1e-12)
    loss_i, td_i, q_i = agent.check_loss(batch)
    np.testing.assert_allclose(q_i, q_inf.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(td_i, (q_inf - y_inf).numpy(), rtol=1e-9, atol=1e-12)

    # --- clip / sgd / target against torch utilities
    gvec = torch.tensor(cg["grads"]).clone().requires_grad_(False)
    big = gvec * (20.0 / gvec.norm())
    clipped, norm = O.clip_by_global_norm(big.numpy(), 5.0, np.float64)
    assert abs(norm - 20.0) < 1e-9 and abs(np.linalg.norm(clipped) - 5.0) < 1e-9
    small, _ = O.clip_by_global_norm((gvec * (1.0 / gvec.norm())).numpy(), 5.0, np.float64)
    assert abs(np.linalg.norm(small) - 1.0) < 1e-12


def test_f32_twin_tracks_f64():
    rng = np.random.default_rng(3)
    shape = (16, 16, 3, 2, 1)
    kw = dict(pixel=True, H=16, W=16, C=6)
    aspec, cspec = O.NetSpec("actor", 2, [100, 100, 50], **kw), O.NetSpec("critic", 2, [100, 100, 50], **kw)
    af, cf = O.init_params(aspec, rng), O.init_params(cspec, rng)
    batch = O.synthetic_batch(rng, 8, shape, 2, True)
    o64 = O.DDPG(aspec, cspec, af, cf, np.float64).train_minibatch(batch)
    o32 = O.DDPG(aspec, cspec, af, cf, np.float32).train_minibatch(batch)
    assert np.abs(o64["q"] - o32["q"]).max() < 1e-5
    assert np.abs(o64["actions"] - o32["actions"]).max() < 1e-5
    rel = np.linalg.norm(o64["critic_grads"] - o32["critic_grads"]) / np.linalg.norm(o64["critic_grads"])
    assert rel < 1e-4


def test_target_copy_and_soft_update():
    rng = np.random.default_rng(0)
    s = rng.normal(size=1000).astype(np.float32)
    t = rng.normal(size=1000).astype(np.float32)
    hard = O.soft_update(t, s, 1.0, np.float32)          # base_network.py:39 "copy" via t - 1.0*(t - s)
    assert np.abs(hard - s).max() < 5e-7                 # not bit exact in f32
    soft = O.soft_update(t, s, 1e-4, np.float64)
    t64, s64 = t.astype(np.float64), s.astype(np.float64)
    np.testing.assert_allclose(soft, t64 * (1 - 1e-4) + s64 * 1e-4, rtol=1e-12)


def test_ou_noise_quirk():
    n = O.OUNoise(2, theta=0.5, sigma=10.0, rng=np.random.RandomState(0))
    xs = np.array([n.sample() for _ in range(200)])
    assert xs.max() <= 1.5 and xs.min() < -1.5           # util.py:155 enforces the upper bound only


# ---------------------------------------------------------------------------------------------
# NAF (naf_cartpole.py) restatement vs torch autograd
# ---------------------------------------------------------------------------------------------
from oracle import naf_np as N


def _torch_head_forward(spec, p, state):
    """HeadSpec network in torch (trunk as torch_forward, then hidden stack + 'fc')."""
    return torch_forward(spec, p, state)


NAF_CASES = [
    dict(shape=(8, 8, 3, 1, 2), B=4, pixel=True, share=True),
    dict(shape=(12, 10, 3, 1, 3), B=3, pixel=True, share=False),
    dict(shape=(2, 2, 7), B=6, pixel=False, share=True),
    dict(shape=(2, 2, 7), B=6, pixel=False, share=False),
    dict(shape=(8, 8, 3, 1, 2), B=4, pixel=True, share=True, batch_norm=True),
    dict(shape=(12, 10, 3, 1, 3), B=3, pixel=True, share=False, batch_norm=True),
]


def naf_specs(shape, pixel, share, A=2, hidden=(100, 50), batch_norm=False):
    kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:])), batch_norm=batch_norm) if pixel else \
        dict(pixel=False, state_elems=int(np.prod(shape)))
    vspec = N.HeadSpec(1, "linear", list(hidden), **kw)
    if share:
        mspec = N.HeadSpec(A, "tanh", [], False, state_elems=hidden[-1], head_only=True)
        lspec = N.HeadSpec(N.num_l_values(A), "linear", [], False, state_elems=hidden[-1], head_only=True)
    else:
        mspec = N.HeadSpec(A, "tanh", list(hidden), **kw)
        lspec = N.HeadSpec(N.num_l_values(A), "linear", list(hidden), **kw)
    return vspec, mspec, lspec


@pytest.mark.parametrize("case", NAF_CASES)
def test_naf_gradients_match_autograd(case):
    rng = np.random.default_rng(17)
    shape, B, pixel, share = case["shape"], case["B"], case["pixel"], case["share"]
    vspec, mspec, lspec = naf_specs(shape, pixel, share, batch_norm=case.get("batch_norm", False))
    vf = N.init_head_params(vspec, rng) + rng.normal(0, 0.05, vspec.num_params()).astype(np.float32)
    mf = N.init_head_params(mspec, rng, small_head=True) + rng.normal(0, 0.05, mspec.num_params()).astype(np.float32)
    lf = N.init_head_params(lspec, rng) + rng.normal(0, 0.05, lspec.num_params()).astype(np.float32)
    tvf = vf + rng.normal(0, 0.01, vf.shape).astype(np.float32)
    batch = O.synthetic_batch(rng, B, shape, 2, pixel)
    s1, a, r, mask, s2 = batch
    naf = N.NAF(vspec, mspec, lspec, vf, mf, lf, share, 2, np.float64)
    naf.target_value = O.Net(vspec, tvf, np.float64)
    out = naf.forward_backward(batch)

    pv, pm, pl, ptv = tparams(vspec, vf), tparams(mspec, mf), tparams(lspec, lf), tparams(vspec, tvf)
    ts1, ts2 = torch.tensor(s1.astype(np.float64)), torch.tensor(s2.astype(np.float64))

    def rep_and_value(p, x):
        Bn = x.shape[0]
        if vspec.pixel:
            full = torch_forward(vspec, p, x)      # value
            # representation = input of the last layer: recompute without the head
            sub = N.HeadSpec(1, "linear", vspec.hidden, True, vspec.H, vspec.W, vspec.C, batch_norm=vspec.batch_norm)
            sub.fc = vspec.fc[:-1]
            rep = torch_forward(sub, p, x)
        else:
            full = torch_forward(vspec, p, x)
            sub = N.HeadSpec(1, "linear", vspec.hidden, False, state_elems=vspec.state_elems)
            sub.fc = vspec.fc[:-1]
            rep = torch_forward(sub, p, x)
        return rep, full

    rep, V = rep_and_value(pv, ts1)
    if share:
        mu = torch.tanh(rep @ pm["fc/weights"] + pm["fc/biases"])
        lv = rep @ pl["fc/weights"] + pl["fc/biases"]
    else:
        mu, lv = torch_forward(mspec, pm, ts1), torch_forward(lspec, pl, ts1)
    L = torch.zeros(B, 2, 2)
    L[:, 0, 0] = torch.exp(lv[:, 0]); L[:, 1, 0] = lv[:, 1]; L[:, 1, 1] = torch.exp(lv[:, 2])
    P = L @ L.transpose(1, 2)
    d = (torch.tensor(a.astype(np.float64)) - mu).unsqueeze(-1)
    adv = (-0.5 * d.transpose(1, 2) @ (P @ d)).reshape(-1, 1)
    with torch.no_grad():
        tv = torch_forward(vspec, ptv, ts2)
        y = torch.tensor(r.astype(np.float64)) + torch.tensor(mask.astype(np.float64)) * 0.99 * tv
    q = V + adv
    loss = ((q - y) ** 2).mean()
    params = list(pv.values()) + list(pm.values()) + list(pl.values())
    grads = torch.autograd.grad(loss, params)
    np.testing.assert_allclose(out["loss"], loss.item(), rtol=1e-10)
    np.testing.assert_allclose(out["advantage"], adv.detach().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out["grads"], np.concatenate([g.numpy().ravel() for g in grads]), rtol=1e-8, atol=1e-11)


def test_naf_optimisers_match_torch():
    rng = np.random.default_rng(5)
    shape = (2, 2, 7)
    vspec, mspec, lspec = naf_specs(shape, False, True)
    vf, mf, lf = N.init_head_params(vspec, rng), N.init_head_params(mspec, rng, True), N.init_head_params(lspec, rng)
    for name, args, topt in [
        ("GradientDescent", {"learning_rate": 0.01}, lambda p: torch.optim.SGD(p, lr=0.01)),
        ("Momentum", {"learning_rate": 0.01, "momentum": 0.9}, lambda p: torch.optim.SGD(p, lr=0.01, momentum=0.9)),
        ("Adam", {"learning_rate": 0.001}, lambda p: torch.optim.Adam(p, lr=0.001, eps=1e-8)),
    ]:
        naf = N.NAF(vspec, mspec, lspec, vf, mf, lf, True, 2, np.float64, gradient_clip=1e9,
                    optimiser=N.make_optimiser(name, args))
        x = torch.tensor(naf.flat().copy(), requires_grad=True)
        opt = topt([x])
        for step in range(4):
            g = rng.normal(size=x.shape[0])
            naf.apply(g.copy())
            opt.zero_grad(); x.grad = torch.tensor(g); opt.step()
            # TF Adam folds the bias correction into lr_t and adds eps to sqrt(v) (not sqrt(v_hat)):
            tol = 2e-5 if name == "Adam" else 1e-12
            np.testing.assert_allclose(naf.flat(), x.detach().numpy(), rtol=0, atol=tol)
