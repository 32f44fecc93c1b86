"""--use-batch-norm (base_network.py:74-79): slim.batch_norm between every conv and its ReLU.  Oracle: oracle/ddpg_np.py
(cross-checked against torch autograd in tests/test_oracle_vs_torch.py)."""
import numpy as np
import pytest

from oracle import ddpg_np as O
from tests.helpers import make_pair, assert_flat_close, assert_grads_close_modulo_pool_ties

pytestmark = pytest.mark.gpu

CASES = [
    pytest.param((8, 8, 3, 1, 2), 4, id="8x8x6-B4"),
    pytest.param((12, 10, 3, 1, 3), 3, id="12x10x9-B3-oddpool"),
    pytest.param((64, 64, 3, 2, 3), 2, id="64x64x18-B2-cfg3-shape"),
    pytest.param((50, 50, 3, 2, 3), 2, id="50x50x18-B2-exps-run_98-shape"),     # 1800-byte f16 rows: 8-byte staging chunks
    pytest.param((32, 32, 3, 2, 3), 3, id="32x32x18-B3"),                      # 18 channels on the narrow (<= 32 columns) dW instance
    pytest.param((84, 84, 3, 1, 2), 2, id="84x84x6-B2"),                       # 6 channels on the wide (65 .. 128 columns) dW instance
]


class HB(object):
    pass


def host_batch(t):
    hb = HB()
    hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = t
    return hb


@pytest.mark.parametrize("shape,B", CASES)
def test_inference_mode_uses_the_initial_moving_statistics(shape, B):
    """action_given / check_loss feed IS_TRAINING False: (z - 0) / sqrt(1 + 1e-3) + beta, the moving averages the
    reference never updates."""
    agent, ref, _ = make_pair(shape, B, True, use_batch_norm=True)
    rng = np.random.default_rng(3)
    t = O.synthetic_batch(rng, B, shape, 2, True)
    try:
        names = [v.name for v in agent.actor.trainable_model_vars()]
        assert "actor/conv1/BatchNorm/beta:0" in names and not any("conv1/biases" in n for n in names)
        want = ref.actor.forward(t[0], training=False)
        got = agent.actor.forward(t[0])
        assert np.abs(got - want["out"]).max() < 1e-5
        for i, name in enumerate(("conv1", "conv2", "conv3")):
            pool = getattr(agent.actor, "pool%d" % (i + 1)).eval(B)
            assert np.abs(pool - want[name][1]).max() < 2e-5, name
        loss, td, q = agent.critic.check_loss(host_batch(t))
        wl, wtd, wq = ref.check_loss(t)
        assert np.abs(q - wq).max() < 1e-5 and np.abs(td - wtd).max() < 1e-5 and abs(loss - wl) < 1e-5 * max(1.0, abs(wl))
    finally:
        agent.close()


@pytest.mark.parametrize("shape,B", CASES)
def test_training_mode_gradients(shape, B):
    """the train ops feed IS_TRAINING True for the whole graph: batch statistics in every network (targets included),
    gradients through the batch moments, dbeta in the '<conv>/BatchNorm/beta' slot."""
    agent, ref, (aspec, cspec) = make_pair(shape, B, True, use_batch_norm=True)
    rng = np.random.default_rng(5)
    t = O.synthetic_batch(rng, B, shape, 2, True)
    hb = host_batch(t)
    try:
        pa = agent.actor.get_params()
        agent.actor.train(hb.state_1)
        assert_grads_close_modulo_pool_ties(
            aspec, agent.actor, B, ref.actor, lambda: ref.actor.forward(t[0]),
            lambda: ref.actor_gradients(t[0])["grads"], agent.actor.get_grads(), what="actor grads (batch norm)", rel=5e-5)
        agent.actor.set_params(pa)
        agent.critic.train(hb)
        assert_grads_close_modulo_pool_ties(
            cspec, agent.critic, B, ref.critic, lambda: ref.critic.forward(t[0], action=np.asarray(t[1])),
            lambda: ref.critic_gradients(t)["grads"], agent.critic.get_grads(), what="critic grads (batch norm)", rel=5e-5)
    finally:
        agent.close()


@pytest.mark.parametrize("shape,B", [CASES[0], CASES[2]])
def test_fused_train_step_with_batch_norm(shape, B):
    """cpp_ddpg_train_step on batch-norm networks == the oracle's minibatch loop (parameters after two minibatches and a
    target update)."""
    agent, ref, (aspec, cspec) = make_pair(shape, B, True, use_batch_norm=True, replay_size=32)
    rng = np.random.default_rng(9)
    try:
        n = 12
        frames = [(rng.integers(0, 256, shape).astype(np.float16) / np.float16(255)) for _ in range(n + 1)]
        seq = [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.uniform(0, 1)), frames[i + 1]) for i in range(n)]
        agent.replay_memory.add_episode(frames[0], seq)
        idxs = rng.integers(0, n, 2 * B)
        batches = []
        for k in range(2):
            ii = idxs[k * B:(k + 1) * B]
            batches.append((np.stack([frames[i] for i in ii]), np.stack([seq[i][0][0] for i in ii]),
                            np.array([[seq[i][1]] for i in ii], np.float32),
                            np.array([[0.0 if i == n - 1 else 1.0] for i in ii], np.float32),
                            np.stack([frames[i + 1] for i in ii])))
        ref.train_step(batches)
        agent.train_step(B, 2, idxs=idxs)
        assert_flat_close(aspec, agent.actor.get_params(), ref.actor.flat(), rel=2e-5, what="actor params")
        assert_flat_close(cspec, agent.critic.get_params(), ref.critic.flat(), rel=2e-5, what="critic params")
        assert_flat_close(aspec, agent.target_actor.get_params(), ref.target_actor.flat(), rel=2e-5, what="target actor")
    finally:
        agent.close()


@pytest.mark.parametrize("shape,B,share", [((8, 8, 3, 1, 2), 4, True), ((12, 10, 3, 1, 3), 3, False), ((64, 64, 3, 2, 3), 2, True)],
                         ids=["8x8x6-share", "12x10x9-own-trunks", "64x64x18-share-cfg4"])
def test_naf_with_batch_norm(shape, B, share):
    """NAF (naf_cartpole.py:93-284) on batch-norm trunks: the debug fetch (:282) and action_given (:253) run in
    inference mode, the train op (:271) in training mode."""
    from tests.test_gpu_naf import make_naf, HB as NHB, CatSpec, params_of, ATOL
    agent, ref, specs = make_naf(shape, B, share, use_batch_norm=True)
    rng = np.random.default_rng(4)
    t = O.synthetic_batch(rng, B, shape, 2, True)
    try:
        dbg = ref.forward_backward(t, backward=False)
        l_values, loss, v, a, vp = agent.naf.debug_values(NHB(t))
        assert np.abs(l_values - dbg["l_values"]).max() < ATOL and np.abs(v - dbg["value"][:, 0]).max() < ATOL
        assert np.abs(vp - dbg["target_value"][:, 0]).max() < ATOL
        assert abs(loss - dbg["loss"]) < ATOL * max(1.0, abs(dbg["loss"]))
        one = agent.naf.action_given(t[0][0].astype(np.float32), add_noise=False)
        assert np.abs(one - ref.action_given(t[0][0])).max() < ATOL
        out = ref.forward_backward(t)
        got_loss = agent.naf.train(NHB(t))
        assert abs(got_loss - out["loss"]) < ATOL * max(1.0, abs(out["loss"]))
        cat = CatSpec(specs)
        assert_flat_close(cat, agent.naf.get_grads(), out["grads"], rel=5e-5, what="naf grads (batch norm)")
        ref.apply(out["grads"])
        assert_flat_close(cat, params_of(agent), ref.flat(), rel=1e-5, what="naf params (batch norm)")
    finally:
        agent.close()


def test_cli_with_batch_norm(capsys):
    """ddpg_cartpole.main --use-batch-norm end to end on the stand-in env (rollouts in inference mode, fused training
    steps in training mode)."""
    import json
    from cartpoleplusplus_amd import ddpg_cartpole as D
    D.main(["--synthetic-env", "--use-raw-pixels", "--use-batch-norm", "--render-width", "16", "--render-height", "16",
            "--batch-size", "8", "--replay-memory-size", "120", "--replay-memory-burn-in", "20", "--max-episode-len", "12",
            "--max-num-actions", "60"])
    out = capsys.readouterr().out
    stats = [json.loads(l.split("\t", 1)[1]) for l in out.splitlines() if l.startswith("STATS")]
    assert len(stats) >= 4 and any(np.isfinite(s["mean_losses"]) for s in stats)


def test_a_refused_geometry_leaves_the_context_usable():
    """batch norm has no dense-dY dW instance for 3 channels: the step is refused with an error -- and must not leave the dW
    reductions it had already queued behind (they point into that agent's buffers; the next agent on the context used to flush them
    into freed memory: a GPU memory fault)."""
    shape, B = (32, 32, 3, 1, 1), 4
    agent, _ref, _ = make_pair(shape, B, True, replay_size=60, use_batch_norm=True)
    try:
        agent.replay_memory.fill_synthetic(40, seed=1)
        with pytest.raises(RuntimeError, match="no kernel for"):
            agent.train_step(B, 2)
    finally:
        agent.close()
    agent, _ref, _ = make_pair((32, 32, 3, 2, 3), B, True, replay_size=60)
    try:
        agent.replay_memory.fill_synthetic(40, seed=1)
        agent.train_step(B, 2); agent.train_step(B, 2)
        agent.actor.ctx.sync()
        assert np.isfinite(agent.critic.get_params()).all() and np.isfinite(agent.actor.get_params()).all()
    finally:
        agent.close()
