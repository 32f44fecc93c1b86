#!/usr/bin/env python
"""DDPG agent for cartpole++ with the interface of the reference's ddpg_cartpole.py, running on
hand-written HIP kernels (MI355X) behind include/cartpolepp_abi.h.

Reference surface kept (paths relative to /root/reference/ddpg_cartpole.py): the flag set :18-57
(`opts` is a module global read by the network classes, as in the reference), ActorNetwork :78-145
(`init_ops_for_training`, `action_given`, `train`, attrs `input_state`, `output_action`,
`exploration_noise`, `train_op`), CriticNetwork :148-248 (`init_ops_for_training`,
`q_gradients_wrt_actions`, `train`, `check_loss`, attrs `input_state`, `input_action`, `q_value`,
`reward`, `terminal_mask`, `input_state_2`, `temporal_difference`, `temporal_difference_loss`),
DeepDeterministicPolicyGradientAgent :251-409 (`post_var_init_setup`, `run_training`, `run_eval`) and
the STATS / EVAL stdout lines :352-361, :396-399.

Decisions on reference defects: B1 low-dim critic = flatten|action -> hidden stack;
B2 pixel critic flattens pool3 before hidden1; B4 the np.clip(1, -1, actions) quirk is reproduced
(upper bound only); B10 mean_losses is filled from the last minibatch's TD loss.
"""
import argparse
import collections
import ctypes as C
import datetime
import json
import os
import sys
import time

import numpy as np

from . import _lib, base_network, replay_memory, util
from ._lib import lib, check, ptr

np.set_printoptions(precision=5, threshold=10000, suppress=True, linewidth=10000)

VERBOSE_DEBUG = False


def toggle_verbose_debug(signal, frame):           # SIGUSR1 (the reference installs the same two handlers at import)
    global VERBOSE_DEBUG
    VERBOSE_DEBUG = not VERBOSE_DEBUG


DUMP_WEIGHTS = False


def set_dump_weights(signal, frame):               # SIGUSR2: run_training dumps the weights after the current episode
    global DUMP_WEIGHTS
    DUMP_WEIGHTS = True


def _install_signal_handlers():
    import signal
    try:
        signal.signal(signal.SIGUSR1, toggle_verbose_debug)
        signal.signal(signal.SIGUSR2, set_dump_weights)
    except ValueError:                             # not the main thread (an embedding application): the toggles stay callable
        pass


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    a = parser.add_argument
    a('--num-eval', type=int, default=0, help="if >0 just run this many episodes with no training")
    a('--max-num-actions', type=int, default=0,
      help="train for (at least) this number of actions (always finish current episode) ignore if <=0")
    a('--max-run-time', type=int, default=0,
      help="train for (at least) this number of seconds (always finish current episode) ignore if <=0")
    a('--ckpt-dir', type=str, default=None, help="if set save ckpts to this dir")
    a('--ckpt-freq', type=int, default=3600, help="freq (sec) to save ckpts")
    a('--batch-size', type=int, default=128, help="training batch size")
    a('--batches-per-step', type=int, default=5, help="number of batches to train per step")
    a('--dont-do-rollouts', action="store_true", help="train from the replay memory only")
    a('--target-update-rate', type=float, default=0.0001,
      help="affine combo for updating target networks each time we run a training step")
    a('--use-batch-norm', action='store_true', help="whether to use batch norm on conv layers")
    a('--actor-hidden-layers', type=str, default="100,100,50", help="actor hidden layer sizes")
    a('--critic-hidden-layers', type=str, default="100,100,50", help="critic hidden layer sizes")
    a('--actor-learning-rate', type=float, default=0.001, help="learning rate for actor")
    a('--critic-learning-rate', type=float, default=0.01, help="learning rate for critic")
    a('--discount', type=float, default=0.99, help="discount for RHS of critic bellman equation update")
    a('--event-log-in', type=str, default=None, help="prepopulate replay memory from this event log")
    a('--replay-memory-size', type=int, default=22000, help="max size of replay memory")
    a('--replay-memory-burn-in', type=int, default=1000,
      help="dont train from replay memory until it reaches this size")
    a('--eval-action-noise', action='store_true', help="whether to use noise during eval")
    a('--action-noise-theta', type=float, default=0.01, help="OrnsteinUhlenbeckNoise theta")
    a('--action-noise-sigma', type=float, default=0.05, help="OrnsteinUhlenbeckNoise sigma")
    util.add_opts(parser)
    # the subset of bullet_cartpole.add_opts (bullet_cartpole.py:13-38) that shapes the observations
    a('--action-repeats', type=int, default=2, help="number of action repeats")
    a('--num-cameras', type=int, default=1, help="how many camera points to render; 1 or 2")
    a('--max-episode-len', type=int, default=200, help="maximum episode len for cartpole")
    a('--use-raw-pixels', action='store_true', help="use raw pixels as state instead of poses")
    a('--render-width', type=int, default=50, help="if --use-raw-pixels render with this width")
    a('--render-height', type=int, default=50, help="if --use-raw-pixels render with this height")
    # the rest of bullet_cartpole.add_opts (bullet_cartpole.py:13-38): read by the reference's pybullet environment when it is on
    # sys.path (make_env); --event-log-out is also honoured by the stand-in environment
    a('--gui', action='store_true', help="pybullet GUI")
    a('--delay', type=float, default=0.0, help="seconds to sleep per simulation step")
    a('--action-force', type=float, default=50.0, help="magnitude of action force applied per step")
    a('--initial-force', type=float, default=55.0, help="magnitude of initial push, in random direction")
    a('--no-random-theta', action='store_true', help="initial push always in the same direction")
    a('--steps-per-repeat', type=int, default=5, help="number of sim steps per repeat")
    a('--event-log-out', type=str, default=None, help="path to record event log.")
    a('--reward-calc', type=str, default='fixed',
      help="'fixed': 1 per step. 'angle': 2*max_angle - ox - oy. 'action': 1.5 - |action|. 'angle_action': both")
    # additions of this build
    a('--host-rng-sampling', action='store_true',
      help="draw minibatch rows with numpy's RNG on the host like the reference (default: Philox on the GPU)")
    a('--sample-seed', type=int, default=0, help="seed of the device-side minibatch sampler")
    a('--replay-store', type=str, default="f16", choices=["f16", "u8"],
      help="element type of the replay memory's state store: f16 as the reference, or u8 pixel codes "
           "(identical batches for rendered frames, half the memory)")
    a('--data-parallel', action='store_true',
      help="one actor-learner per GPU (launch with torch.distributed.run): own environment and replay shard per process, the "
           "gradients of every minibatch all-reduced over RCCL (cartpoleplusplus_amd/distributed.py)")
    a('--sync-every', type=int, default=1,
      help="--data-parallel: 1 = gradient all-reduce per minibatch; k > 1 = k local minibatch updates, then parameter averaging")
    a('--overlap-allreduce', action='store_true',
      help="--data-parallel: reduce the fully connected layers' gradients beside the conv backward")
    a('--synthetic-env', action='store_true', help="random-frame stand-in env (pybullet stays optional)")
    return parser


def default_opts(**overrides):
    o = build_parser().parse_args([])
    for k, v in overrides.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


opts = default_opts()        # module global, as in the reference (ddpg_cartpole.py:56)


def set_opts(o):
    global opts
    opts = o


def _hidden(spec):
    return [int(s) for s in str(spec).split(",")]


class _OpHandle(object):
    """names a fetchable 'tensor' of a network (output_action, q_value, train_op ...)."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<op %s>" % self.name


class ActorNetwork(base_network.Network):
    """ the actor represents the learnt policy mapping states to actions"""

    def __init__(self, namespace, input_state, action_dim):
        super(ActorNetwork, self).__init__(namespace)
        self.input_state = input_state
        self.action_dim = int(action_dim)
        self.exploration_noise = util.OrnsteinUhlenbeckNoise(action_dim, opts.action_noise_theta,
                                                             opts.action_noise_sigma)
        opts.hidden_layers = opts.actor_hidden_layers                 # ddpg_cartpole.py:91
        self.input_state_network(self.input_state, opts)
        self._build_native(_lib.CPP_ACTOR, action_dim, max(int(opts.batch_size), 1))
        self.output_action = _OpHandle(namespace + "/output_action")
        self.train_op = None
        self.critic = None

    def init_ops_for_training(self, critic):
        # gradients of output_action w.r.t. the actor's variables with grad_ys = -dQ/da from the critic
        # (sum over the batch), clipped by global norm, applied with plain SGD (ddpg_cartpole.py:102-119)
        self.critic = critic
        critic._register_actor_training(self)
        self.train_op = _OpHandle(self.namespace + "/optimiser/train_op")

    def forward(self, states):
        """output_action for a host batch of states (B, ...) -> (B, action_dim)."""
        s, dt = _lib.as_state_array(states)
        B = s.shape[0]
        out = np.empty((B, self.action_dim), np.float32)
        check(lib.cpp_net_forward(self.handle, ptr(s), dt, B, None, ptr(out)))
        return out

    def actions_given(self, states, add_noise=False):
        """`action_given` for many env workers at once: states (B, ...) -> actions (B, action_dim),
        row i identical to action_given(states[i]) -- every image is whitened with its own statistics.  With
        add_noise each row gets its own Ornstein-Uhlenbeck process (one per worker, created on first use)."""
        s, dt = _lib.as_state_array(states)
        B = s.shape[0]
        actions = np.empty((B, self.action_dim), np.float32)
        check(lib.cpp_net_forward_each(self.handle, ptr(s), dt, B, None, ptr(actions)))
        if add_noise:
            procs = self.__dict__.setdefault("_worker_noise", [])
            while len(procs) < B:
                procs.append(util.OrnsteinUhlenbeckNoise(self.action_dim, opts.action_noise_theta, opts.action_noise_sigma))
            for i in range(B):
                actions[i] += procs[i].sample()
            actions = np.minimum(1, actions)     # the reference's clip quirk, per row (:134)
        return actions

    def action_given(self, state, add_noise=False):
        # feed explicitly provided state (batch of one; whitening uses this image's own statistics)
        actions = self.forward(np.asarray(state)[None])
        # NOTE: noise is added outside the device graph, as in the reference (:127-134)
        if add_noise:
            if VERBOSE_DEBUG:
                pre_noise = str(actions)
            actions[0] += self.exploration_noise.sample()
            actions = np.minimum(1, actions)     # np.clip(1, -1, actions): upper bound only (:134)
            if VERBOSE_DEBUG:
                print("TRAIN action_given pre_noise %s post_noise %s" % (pre_noise, actions))
        return actions

    def train(self, state):
        # training actor only requires state since we are trying to maximise the q_value according
        # to the critic (ddpg_cartpole.py:140-145).  `state` may be a host array or a device Batch.
        if self.critic is None:
            raise Exception("init_ops_for_training not called")
        trainer = self.critic._trainer()
        dev = trainer.device_batch_for(state, state_only=True)
        check(lib.cpp_ddpg_train_actor(trainer.handle, dev.handle))
        if opts.print_gradients:
            print("gradient %s l2_norm %s" % (self.namespace, trainer.last_stats()[1]))


class _Trainer(object):
    """owns the cpp_ddpg handle that binds (actor, critic, target_actor, target_critic) and the
    hyper-parameters -- the 'optimiser' variable scope of the reference."""

    def __init__(self, actor, critic, target_actor, target_critic):
        hp = _lib.DdpgHyper(float(opts.actor_learning_rate), float(opts.critic_learning_rate),
                            float(opts.discount), util.gradient_clip_value(opts),
                            float(opts.target_update_rate))
        h = C.c_void_p()
        check(lib.cpp_ddpg_create(actor.ctx.handle, actor.handle, critic.handle, target_actor.handle,
                                  target_critic.handle, C.byref(hp), C.byref(h)))
        self.handle, self.ctx = h, actor.ctx
        self.nets = (actor, critic, target_actor, target_critic)
        self.state_elems = int(actor._state_elems)
        self.action_dim = int(actor.action_dim)
        self._upload = {}

    def device_batch_for(self, batch, state_only=False):
        if isinstance(batch, replay_memory.Batch) and batch.device is not None:
            return batch.device
        if state_only:
            s1 = np.asarray(batch.state_1 if hasattr(batch, "state_1") else batch)
            B = s1.shape[0]
            args = (s1, None, None, None, None)
        else:
            s1 = np.asarray(batch.state_1)
            B = s1.shape[0]
            args = (s1, batch.action, batch.reward, batch.terminal_mask, batch.state_2)
        if B not in self._upload:
            self._upload[B] = replay_memory.DeviceBatch(B, self.state_elems, self.action_dim, self.ctx)
        return self._upload[B].upload(*args)

    def last_stats(self):
        out = np.zeros(3, np.float32)
        check(lib.cpp_ddpg_last_stats(self.handle, ptr(out)))
        return out

    def last_values(self, B):
        """(actions, dq_da, q, td) of the last minibatch's gradient pass, as the device left them -- the values the
        reference dumps under VERBOSE_DEBUG (ddpg_cartpole.py:339-349); also after a fused / graph-replayed train_step."""
        B, A = int(B), self.action_dim
        actions, dq_da = np.empty((B, A), np.float32), np.empty((B, A), np.float32)
        q, td = np.empty((B, 1), np.float32), np.empty((B, 1), np.float32)
        check(lib.cpp_ddpg_last_values(self.handle, B, ptr(actions), ptr(dq_da), ptr(q), ptr(td)))
        return actions, dq_da, q, td

    def grad_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        check(lib.cpp_ddpg_grad_buffer(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def close(self):
        for b in self._upload.values():
            b.close()
        if self.handle:
            lib.cpp_ddpg_destroy(self.handle)
            self.handle = None


class CriticNetwork(base_network.Network):
    """ the critic represents a mapping from state & actors action to a quality score."""

    def __init__(self, namespace, actor):
        super(CriticNetwork, self).__init__(namespace)
        # input state to the critic is the _same_ state given to the actor; input action is the
        # (gradient-stopped) output action of the actor unless one is fed (ddpg_cartpole.py:161-162)
        self.actor = actor
        self.input_state = actor.input_state
        self.input_action = _OpHandle(namespace + "/input_action")
        self.action_dim = actor.action_dim
        self._state_elems = int(np.prod([int(d) for d in self.input_state.get_shape()[1:]]))
        if opts.use_raw_pixels:
            # conv trunk -> (flatten) -> 200 -> 50 -> concat action -> 50  (:166-171, intent; B2)
            self.simple_conv_net_on(self.input_state, opts)
            self._hidden = []
        else:
            # flatten(state) | action -> hidden stack (:172-177; B1: opts=None means no dropout)
            self.hidden_layers_starting_at(self.input_state, opts.critic_hidden_layers)
        self._build_native(_lib.CPP_CRITIC, self.action_dim, max(int(opts.batch_size), 1))
        self.q_value = _OpHandle(namespace + "/q_value")
        self.target_critic = None
        self._ddpg = None
        self._actor_for_training = None
        self.train_op = None

    def _register_actor_training(self, actor):
        self._actor_for_training = actor

    def init_ops_for_training(self, target_critic):
        # bellman: Q(s1, a) = reward + terminal_mask * discount * Q'(s2, A'(s2)); squared TD loss;
        # clip by global norm; SGD (ddpg_cartpole.py:186-218)
        self.target_critic = target_critic
        self.reward = base_network.Placeholder([None, 1], name="critic_reward")
        self.terminal_mask = base_network.Placeholder([None, 1], name="critic_terminal_mask")
        self.input_state_2 = target_critic.input_state
        self.temporal_difference = _OpHandle(self.namespace + "/temporal_difference")
        self.temporal_difference_loss = _OpHandle(self.namespace + "/temporal_difference_loss")
        self.train_op = _OpHandle(self.namespace + "/optimiser/train_op")

    def _trainer(self):
        if self._ddpg is None:
            actor = self._actor_for_training or self.actor
            if self.target_critic is not None:
                self._ddpg = _Trainer(actor, self, self.target_critic.actor, self.target_critic)
            else:   # actor-only training does not touch the targets; bind the live nets as stand-ins
                self._ddpg = _Trainer(actor, self, actor, self)
        return self._ddpg

    def forward(self, states, actions):
        s, dt = _lib.as_state_array(states)
        B = s.shape[0]
        a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(B, self.action_dim))
        out = np.empty((B, 1), np.float32)
        check(lib.cpp_net_forward(self.handle, ptr(s), dt, B, ptr(a), ptr(out)))
        return out

    def q_gradients_wrt_actions(self, batch=None):
        """ gradients for the q.value w.r.t just input_action; used for actor training.  With no
        argument returns the op handle (graph-building use, :111); with a batch / state array returns
        dQ/da evaluated at a = actor(state_1), shape (B, action_dim)."""
        if batch is None:
            return _OpHandle(self.namespace + "/q_gradients_wrt_actions")
        trainer = self._trainer()
        dev = trainer.device_batch_for(batch, state_only=True)
        out = np.empty((dev.size, self.action_dim), np.float32)
        check(lib.cpp_ddpg_q_gradients_wrt_actions(trainer.handle, dev.handle, ptr(out), None, None))
        return out

    def train(self, batch):
        if self.target_critic is None:
            raise Exception("init_ops_for_training not called")
        trainer = self._trainer()
        dev = trainer.device_batch_for(batch)
        check(lib.cpp_ddpg_train_critic(trainer.handle, dev.handle))
        if opts.print_gradients:
            print("gradient %s l2_norm %s" % (self.namespace, trainer.last_stats()[2]))

    def check_loss(self, batch):
        if self.target_critic is None:
            raise Exception("init_ops_for_training not called")
        trainer = self._trainer()
        dev = trainer.device_batch_for(batch)
        B = dev.size
        loss = np.zeros(1, np.float32)
        td = np.empty((B, 1), np.float32)
        q = np.empty((B, 1), np.float32)
        check(lib.cpp_ddpg_check_loss(trainer.handle, dev.handle, ptr(loss), ptr(td), ptr(q)))
        return [loss[0], td, q]


class DeepDeterministicPolicyGradientAgent(object):
    def __init__(self, env):
        self.env = env
        state_shape = self.env.observation_space.shape
        action_dim = self.env.action_space.shape[1]
        # replay memory: f16 state store resident in HBM (ddpg_cartpole.py:257-261)
        self.replay_memory = replay_memory.ReplayMemory(opts.replay_memory_size, state_shape, action_dim,
                                                       store_dtype=opts.replay_store)
        # s1 and s2 placeholders
        batched_state_shape = [None] + list(state_shape)
        s1 = base_network.Placeholder(batched_state_shape)
        s2 = base_network.Placeholder(batched_state_shape)
        # base models for actor / critic and their corresponding target networks
        self.actor = ActorNetwork("actor", s1, action_dim)
        self.critic = CriticNetwork("critic", self.actor)
        self.target_actor = ActorNetwork("target_actor", s2, action_dim)
        self.target_critic = CriticNetwork("target_critic", self.target_actor)
        # training ops
        self.actor.init_ops_for_training(self.critic)
        self.critic.init_ops_for_training(self.target_critic)
        self.train_steps = 0

    def initialise_variables(self, seed=None):
        """tf.initialize_all_variables() (ddpg_cartpole.py:424)."""
        rng = np.random.RandomState(seed) if seed is not None else np.random
        for net in (self.actor, self.critic, self.target_actor, self.target_critic):
            net.initialise_variables(rng)

    def networks(self):
        return [self.actor, self.critic, self.target_actor, self.target_critic]

    def post_var_init_setup(self):
        if opts.event_log_in:
            self.replay_memory.reset_from_event_log(opts.event_log_in)
        # hook networks up to their targets ( one off clobber of all vars in target network )
        self.target_actor.set_as_target_network_for(self.actor, opts.target_update_rate)
        self.target_critic.set_as_target_network_for(self.critic, opts.target_update_rate)

    @property
    def trainer(self):
        return self.critic._trainer()

    def train_step(self, batch_size, batches_per_step, idxs=None):
        """the inner train step ddpg_cartpole.py:331-337 as ONE device-side sequence (hipGraph after
        the first call): batches_per_step x {sample+gather, actor update, critic update}, then both
        target soft updates.  idxs: optional (batches_per_step*batch_size) rows instead of Philox."""
        t = self.trainer
        if idxs is None and getattr(opts, "data_parallel", False):      # one learner of N: the collective step (distributed.py)
            self._dp_learner(batch_size).train_step(batches_per_step)
            self.train_steps += 1
            return
        rows = None
        if idxs is not None:
            rows = np.ascontiguousarray(np.asarray(idxs).reshape(-1), dtype=np.int32)
            assert len(rows) == batch_size * batches_per_step
        self.replay_memory.stats[">batch"] += batches_per_step
        check(lib.cpp_ddpg_train_step(t.handle, self.replay_memory.handle, int(batch_size),
                                      int(batches_per_step), ptr(rows), int(opts.sample_seed)))
        self.train_steps += 1

    def _dp_learner(self, batch_size):
        cur = getattr(self, "_learner", None)
        if cur is None or cur.B != int(batch_size):
            from . import distributed
            if cur is not None:
                cur.close()
            self._learner = distributed.learner_for_agent(self, opts, batch_size)
        return self._learner

    def run_training(self, max_num_actions, max_run_time, batch_size, batches_per_step, saver_util):
        start_time = time.time()
        num_actions_taken = 0
        n = 0
        while True:
            rewards = []
            losses = []
            if not opts.dont_do_rollouts:
                # run an episode (physics + rendering stay on the host)
                state_1 = self.env.reset()
                initial_state = np.copy(state_1)
                action_reward_state_sequence = []
                done = False
                while not done:
                    action = self.actor.action_given(state_1, add_noise=True)
                    state_2, reward, done, _ = self.env.step(action)
                    rewards.append(reward)
                    action_reward_state_sequence.append((action, reward, np.copy(state_2)))
                    state_1 = state_2
                self.replay_memory.add_episode(initial_state, action_reward_state_sequence)

            # do a training step (after waiting for buffer to fill a bit...)
            if self.replay_memory.size() > opts.replay_memory_burn_in:
                if opts.host_rng_sampling:
                    for _ in range(batches_per_step):
                        batch = self.replay_memory.batch(batch_size)
                        self.actor.train(batch)
                        self.critic.train(batch)
                    self.target_actor.update_weights()
                    self.target_critic.update_weights()
                else:
                    self.train_step(batch_size, batches_per_step)
                losses.append(float(self.trainer.last_stats()[0]))
                if VERBOSE_DEBUG:
                    batch = self.replay_memory.batch(batch_size)
                    td_loss, td, q_value = self.critic.check_loss(batch)
                    print("-----")
                    print("temporal_difference_loss", td_loss)
                    print("temporal_difference", td.T)
                    print("q_value", q_value.T)

            stats = collections.OrderedDict()
            stats["time"] = time.time()
            stats["n"] = n
            stats["mean_losses"] = float(np.mean(losses)) if losses else float("nan")
            stats["total_reward"] = float(np.sum(rewards))
            stats["episode_len"] = len(rewards)
            stats["replay_memory_stats"] = self.replay_memory.current_stats()
            print("STATS %s\t%s" % (datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S'), json.dumps(stats)))
            sys.stdout.flush()
            n += 1

            if saver_util is not None:
                saver_util.save_if_required()
            if VERBOSE_DEBUG or n % 10 == 0:
                self.run_eval(1)
            global DUMP_WEIGHTS
            if DUMP_WEIGHTS:
                self.debug_dump_network_weights()
                DUMP_WEIGHTS = False

            num_actions_taken += len(rewards)
            if max_num_actions > 0 and num_actions_taken > max_num_actions:
                break
            if max_run_time > 0 and time.time() > start_time + max_run_time:
                break
            if opts.dont_do_rollouts and max_num_actions <= 0 and max_run_time <= 0:
                break

    def debug_dump_network_weights(self):
        fn = "/tmp/weights.%s" % time.time()
        with open(fn, "w") as f:
            f.write("DUMP time %s\n" % time.time())
            for net in self.networks():
                for var in net.trainable_model_vars():
                    f.write("VAR %s %s\n" % (var.name, tuple(var.get_shape())))
                    f.write("%s\n" % var.eval())
        print("weights written to", fn)
        return fn

    def run_eval(self, num_episodes, add_noise=False):
        """ run num_episodes of eval and output episode length and rewards """
        for i in range(num_episodes):
            state = self.env.reset()
            total_reward = 0
            steps = 0
            done = False
            while not done:
                action = self.actor.action_given(state, add_noise)
                state, reward, done, _ = self.env.step(action)
                print("EVALSTEP r%s %s %s %s %s" % (i, steps, np.squeeze(action), np.linalg.norm(action), reward))
                total_reward += reward
                steps += 1
            print("EVAL", i, steps, total_reward)
        sys.stdout.flush()

    def close(self):
        if getattr(self, "_learner", None) is not None:
            self._learner.close()
            self._learner = None
        if self.critic._ddpg is not None:
            self.critic._ddpg.close()
        for net in (self.actor, self.critic, self.target_actor, self.target_critic):
            net.close()
        self.replay_memory.close()


def make_env(o):
    if o.synthetic_env:
        from .synthetic_env import SyntheticCartpole
        return SyntheticCartpole(o)
    try:
        import bullet_cartpole      # the reference's pybullet env, if the user has it on sys.path
    except ImportError as e:
        raise ImportError("bullet_cartpole / pybullet not importable (%s); physics stays on the host "
                          "CPU and is not part of this package -- use --synthetic-env for a stand-in" % e)
    return bullet_cartpole.BulletCartpole(opts=o, discrete_actions=False)


def main(argv=None):
    _install_signal_handlers()
    set_opts(build_parser().parse_args(argv))
    sys.stderr.write("%s\n" % opts)
    env = make_env(opts)
    agent = DeepDeterministicPolicyGradientAgent(env=env)
    # either load the latest ckpt or init variables (ddpg_cartpole.py:419-424)
    saver_util = None
    if opts.ckpt_dir is not None and not (opts.data_parallel and int(os.environ.get("RANK", "0")) != 0):      # rank 0 keeps the checkpoints
        saver_util = util.SaverUtil(agent, opts.ckpt_dir, opts.ckpt_freq)
    else:
        agent.initialise_variables()
    for net in (agent.actor, agent.critic, agent.target_actor, agent.target_critic):
        for v in net.trainable_model_vars():
            sys.stderr.write("%s %s\n" % (v.name, util.shape_and_product_of(v.shape)))
    agent.post_var_init_setup()
    if opts.num_eval > 0:
        agent.run_eval(opts.num_eval, opts.eval_action_noise)
    else:
        agent.run_training(opts.max_num_actions, opts.max_run_time, opts.batch_size,
                           opts.batches_per_step, saver_util)
        if saver_util is not None:
            saver_util.force_save()
    env.reset()
    agent.close()
    if hasattr(env, "close"):
        env.close()


if __name__ == "__main__":
    main()
