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
