"""--use-dropout (base_network.py:69-70): slim.dropout (keep 0.5) after the ReLU of the actor's / NAF networks' hidden
layers.  TensorFlow's random bits cannot be reproduced; the device draws its keep bits from Philox4x32-10 keyed by
(network, layer, forward count), and the oracle is fed the same masks."""
import numpy as np
import pytest

from oracle import ddpg_np as O
from tests.helpers import make_pair, assert_flat_close, dropout_masks

pytestmark = pytest.mark.gpu


class HB(object):
    def __init__(self, t):
        self.state_1, self.action, self.reward, self.terminal_mask, self.state_2 = t


@pytest.mark.parametrize("shape,B", [((2, 2, 7), 5), ((8, 8, 3, 1, 2), 4)], ids=["lowdim-28-B5", "8x8x6-B4"])
def test_dropout_training_and_inference(shape, B):
    pixel = len(shape) == 5
    agent, ref, (aspec, cspec) = make_pair(shape, B, pixel, use_dropout=True)
    rng = np.random.default_rng(21)
    t = O.synthetic_batch(rng, B, shape, 2, pixel)
    hidden = [100, 100, 50]
    try:
        # inference (action_given / check_loss): no dropout
        want = ref.actor.forward(t[0], training=False)["out"]
        assert np.abs(agent.actor.forward(t[0]) - want).max() < 1e-5
        loss, td, q = agent.critic.check_loss(HB(t))
        wl, wtd, wq = ref.check_loss(t)
        assert np.abs(q - wq).max() < 1e-5 and np.abs(td - wtd).max() < 1e-5
        # the train ops: forward count 0 of the actor (actor.train) and of the target actor (critic.train)
        ref.actor.drop_masks = dropout_masks("actor", hidden, B, 0)
        ag = ref.actor_gradients(t[0])
        pa, pc = agent.actor.get_params(), agent.critic.get_params()
        agent.actor.train(HB(t).state_1)
        assert_flat_close(aspec, agent.actor.get_grads(), ag["grads"], what="actor grads (dropout)")
        agent.actor.set_params(pa)
        ref.target_actor.drop_masks = dropout_masks("target_actor", hidden, B, 0)
        cg = ref.critic_gradients(t)
        agent.critic.train(HB(t))
        assert_flat_close(cspec, agent.critic.get_grads(), cg["grads"], what="critic grads (dropout in the target actor)")
        agent.critic.set_params(pc)
        # a second actor update draws new masks (forward count 1)
        ref.actor.drop_masks = dropout_masks("actor", hidden, B, 1)
        ag1 = ref.actor_gradients(t[0])
        agent.actor.train(HB(t).state_1)
        assert_flat_close(aspec, agent.actor.get_grads(), ag1["grads"], what="actor grads (dropout, second forward)")
        assert np.abs(ag1["grads"] - ag["grads"]).max() > 1e-6
    finally:
        agent.close()


def test_fused_train_step_with_dropout():
    """cpp_ddpg_train_step: minibatch k uses forward count k of the actor and of the target actor."""
    shape, B = (2, 2, 7), 6
    agent, ref, (aspec, cspec) = make_pair(shape, B, False, use_dropout=True, replay_size=32)
    rng = np.random.default_rng(23)
    hidden = [100, 100, 50]
    try:
        n = 12
        frames = [rng.normal(0, 1, shape).astype(np.float32) for _ in range(n + 1)]
        seq = [(rng.uniform(-1, 1, (1, 2)).astype(np.float32), float(rng.uniform(0, 1)), frames[i + 1]) for i in range(n)]
        agent.replay_memory.add_episode(frames[0], seq)
        idxs = rng.integers(0, n, 2 * B)
        for k in range(2):
            ii = idxs[k * B:(k + 1) * B]
            batch = (np.stack([frames[i] for i in ii]).astype(np.float16), np.stack([seq[i][0][0] for i in ii]),
                     np.array([[seq[i][1]] for i in ii], np.float32),
                     np.array([[0.0 if i == n - 1 else 1.0] for i in ii], np.float32),
                     np.stack([frames[i + 1] for i in ii]).astype(np.float16))
            ref.actor.drop_masks = dropout_masks("actor", hidden, B, k)
            ref.target_actor.drop_masks = dropout_masks("target_actor", hidden, B, k)
            ref.train_minibatch(batch)
        ref.update_targets()
        agent.train_step(B, 2, idxs=idxs)
        assert_flat_close(aspec, agent.actor.get_params(), ref.actor.flat(), rel=2e-5, what="actor params")
        assert_flat_close(cspec, agent.critic.get_params(), ref.critic.flat(), rel=2e-5, what="critic params")
    finally:
        agent.close()


@pytest.mark.parametrize("share", [True, False], ids=["shared-representation", "own-trunks"])
def test_naf_with_dropout(share):
    """NAF: the value / target-value (and, with their own trunks, mu / l_values) hidden stacks drop out in the train op
    (naf_cartpole.py:271), not in the debug fetch (:282)."""
    from tests.test_gpu_naf import make_naf, HB as NHB, CatSpec, params_of, ATOL
    shape, B = (2, 2, 7), 6
    agent, ref, specs = make_naf(shape, B, share, use_dropout=True)
    rng = np.random.default_rng(8)
    t = O.synthetic_batch(rng, B, shape, 2, False)
    hidden = [100, 50]
    try:
        dbg = ref.forward_backward(t, backward=False)
        l_values, loss, v, a, vp = agent.naf.debug_values(NHB(t))
        assert np.abs(l_values - dbg["l_values"]).max() < ATOL and np.abs(v - dbg["value"][:, 0]).max() < ATOL
        ref.value.drop_masks = dropout_masks("value", hidden, B, 0)
        ref.target_value.drop_masks = dropout_masks("target_value", hidden, B, 0)
        if not share:
            ref.mu.drop_masks = dropout_masks("naf/output_action", hidden, B, 0)
            ref.l.drop_masks = dropout_masks("naf/l_values", hidden, B, 0)
        out = ref.forward_backward(t)
        got_loss = agent.naf.train(NHB(t))
        assert abs(got_loss - out["loss"]) < ATOL * max(1.0, abs(out["loss"]))
        assert_flat_close(CatSpec(specs), agent.naf.get_grads(), out["grads"], what="naf grads (dropout)")
    finally:
        agent.close()
