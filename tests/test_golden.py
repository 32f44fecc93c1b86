"""Golden-vector tests.  CPU: the oracle reproduces the committed vectors (and the replay known answers
of the reference's own test).  GPU: the HIP path reproduces the same vectors."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import ddpg_np as O
from oracle.replay_np import OracleReplayMemory

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP_FILES = sorted(glob.glob(os.path.join(GOLD, "ddpg_step_*.npz")))


def _shape_of(name):
    from tests.golden.make_golden import CASES
    c = CASES[name]
    return c["shape"], c["B"], c["pixel"]


def _specs(shape, pixel):
    kw = dict(pixel=True, H=shape[0], W=shape[1], C=int(np.prod(shape[2:]))) if pixel else \
        dict(pixel=False, state_elems=int(np.prod(shape)))
    return O.NetSpec("actor", 2, [100, 100, 50], **kw), O.NetSpec("critic", 2, [100, 100, 50], **kw)


def _replay_known(make):
    k = json.load(open(os.path.join(GOLD, "replay_known_answers.json")))
    su = k["setup"]
    rm = make(su["buffer_size"], tuple(su["state_shape"]), su["action_dim"], su["load_factor"])
    c = k["adds_to_full"]
    rm.add_episode(c["initial_state"], [tuple(x) for x in c["action_reward_state"]])
    e = c["expect"]
    assert (rm.size(), rm.insert, rm.full) == (e["size"], e["insert"], e["full"])
    assert [int(rm.state[i][0][0]) for i in range(4)] == e["state_first_elements"]
    rm = make(su["buffer_size"], tuple(su["state_shape"]), su["action_dim"], su["load_factor"])
    s_for = lambda i: (np.arange(1, 7) + 10 * i).reshape(2, 3)
    for ep in k["adds_over_full"]["episodes"]:
        rm.add_episode(s_for(ep["first"]), [(10 * i + 7, 10 * i + 8, s_for(i)) for i in ep["steps"]])
    b = rm.batch(idxs=[0, 1, 2])
    e = k["adds_over_full"]["expect"]
    assert rm.size() == e["size"]
    assert np.array_equal(b.reward, e["reward"]) and np.array_equal(b.terminal_mask, e["terminal_mask"])


def test_oracle_replay_reproduces_reference_known_answers():
    _replay_known(lambda n, s, a, lf: OracleReplayMemory(n, s, a, lf))


@pytest.mark.parametrize("path", STEP_FILES, ids=[os.path.basename(p)[10:-4] for p in STEP_FILES])
def test_oracle_reproduces_golden_step(path):
    g = np.load(path)
    shape, B, pixel = _shape_of(os.path.basename(path)[10:-4])
    aspec, cspec = _specs(shape, pixel)
    agent = O.DDPG(aspec, cspec, g["actor"], g["critic"], np.float64)
    agent.set_targets(g["target_actor"], g["target_critic"])
    outs = agent.train_step([tuple(g["b%d_%s" % (i, k)] for k in ("s1", "a", "r", "mask", "s2")) for i in range(2)])
    for i, o in enumerate(outs):
        for k in ("actions", "q", "td", "dq_da", "actor_grads", "critic_grads"):
            np.testing.assert_allclose(o[k], g["o%d_%s" % (i, k)], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(agent.actor.flat(), g["new_actor"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(agent.target_critic.flat(), g["new_target_critic"], rtol=1e-10, atol=1e-13)


@pytest.mark.gpu
def test_device_replay_reproduces_reference_known_answers():
    from cartpoleplusplus_amd.replay_memory import ReplayMemory
    _replay_known(lambda n, s, a, lf: ReplayMemory(n, s, a, lf))


@pytest.mark.gpu
@pytest.mark.parametrize("path", STEP_FILES, ids=[os.path.basename(p)[10:-4] for p in STEP_FILES])
def test_hip_path_reproduces_golden_step(path):
    from cartpoleplusplus_amd import ddpg_cartpole as D
    from tests.helpers import FakeEnv, make_opts, assert_flat_close
    g = np.load(path)
    shape, B, pixel = _shape_of(os.path.basename(path)[10:-4])
    aspec, cspec = _specs(shape, pixel)
    make_opts(D, shape, B, pixel, replay_memory_size=16)
    agent = D.DeepDeterministicPolicyGradientAgent(FakeEnv(shape))
    try:
        agent.actor.set_params(g["actor"]); agent.critic.set_params(g["critic"])
        agent.target_actor.set_params(g["target_actor"]); agent.target_critic.set_params(g["target_critic"])
        agent.target_actor.update_weights_op = agent.target_actor._create_variables_copy_op(agent.actor, D.opts.target_update_rate)
        agent.target_critic.update_weights_op = agent.target_critic._create_variables_copy_op(agent.critic, D.opts.target_update_rate)

        class HB(object):
            pass
        for i in range(2):
            hb = HB()
            hb.state_1, hb.action, hb.reward, hb.terminal_mask, hb.state_2 = (
                g["b%d_%s" % (i, k)] for k in ("s1", "a", "r", "mask", "s2"))
            if i == 0:
                loss, td, q = agent.critic.check_loss(hb)
                assert np.abs(q - g["o0_q"]).max() < 1e-5 and np.abs(td - g["o0_td"]).max() < 1e-5
                assert np.abs(agent.actor.forward(hb.state_1) - g["o0_actions"]).max() < 1e-5
                assert np.abs(agent.critic.q_gradients_wrt_actions(hb) - g["o0_dq_da"]).max() < 1e-5
            agent.actor.train(hb.state_1)
            agent.critic.train(hb)
        agent.target_actor.update_weights(); agent.target_critic.update_weights()
        assert_flat_close(aspec, agent.actor.get_params(), g["new_actor"], rel=1e-5, what="actor")
        assert_flat_close(cspec, agent.critic.get_params(), g["new_critic"], rel=1e-5, what="critic")
        assert_flat_close(aspec, agent.target_actor.get_params(), g["new_target_actor"], rel=1e-6, what="target actor")
        assert_flat_close(cspec, agent.target_critic.get_params(), g["new_target_critic"], rel=1e-6, what="target critic")
    finally:
        agent.close()
