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
import ctypes as C
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
    a('--exact-products', action='store_true',
      help="conv1 / conv2 on the f16 / bf16 matrix pipes with EVERY operand bit (three f16 pieces, nine bf16 products) instead of "
           "operands to within one f32 ulp (two / six): ~0.87 x the speed (cpp_ctx_set_precision)")
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
    a('--async-rollouts', action='store_true',
      help="play the episodes on a rollout thread while the learner trains back to back (training_loop.py): the learner -- and, with "
           "--data-parallel, every other rank -- never waits for this process's environment once it is past burn-in")
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
        # the reference's loop calls `actor.train(batch.state_1)` and then `critic.train(batch)` (:333-334).  When `state` is the
        # state_1 COLUMN of a replay Batch, the update is deferred: if the critic's call on the same Batch follows, both run as the
        # fused device sequence of train_step on those rows (legal: the critic's update never reads the live actor and the actor's
        # never writes the critic); anything else that touches the actor or the trainer first runs it on its own (_Trainer.flush).
        if (isinstance(state, replay_memory.StateColumn) and state.field == "state_1" and state.batch.in_replay()
                and self.critic.target_critic is not None):
            trainer.defer_actor(self, state.batch)
            return
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
        self._h, self.ctx = h, actor.ctx
        self.nets = (actor, critic, target_actor, target_critic)
        self.state_elems = int(actor._state_elems)
        self.action_dim = int(actor.action_dim)
        self._upload = {}
        self._pending = None         # (actor network, Batch) of a deferred actor.train(batch.state_1)
        self.fused_pairs = 0         # actor.train + critic.train pairs that ran as one fused sequence (tests, profiles)

    @property
    def handle(self):
        """the cpp_ddpg; a deferred actor update lands first (every trainer op but the fused pair goes through here)."""
        self.flush()
        return self._h

    def defer_actor(self, actor, batch):
        self.flush()
        self._pending = (actor, batch)
        actor._before_use = self.flush

    def flush(self):
        """run a deferred actor.train now, on its own (ddpg_cartpole.py:140-145)."""
        if self._pending is None:
            return
        (actor, batch), self._pending = self._pending, None
        actor._before_use = None
        check(lib.cpp_ddpg_train_actor(self._h, batch.device.handle))
        if opts.print_gradients:
            print("gradient %s l2_norm %s" % (actor.namespace, self.last_stats()[1]))

    def train_pair(self, batch):
        """critic.train(batch) arriving right behind the deferred actor.train(batch.state_1) of the SAME draw: both updates as the
        fused minibatch of cpp_ddpg_train_step on the draw's rows (cpp_ddpg_train_rows); False if that is not the situation."""
        if self._pending is None or self._pending[1] is not batch or not batch.in_replay():
            return False
        (actor, _), self._pending = self._pending, None
        actor._before_use = None
        rm = batch._memory
        check(lib.cpp_ddpg_train_rows(self._h, rm.handle, len(batch.idxs), ptr(batch.idxs)))
        self.fused_pairs += 1
        if opts.print_gradients:
            st = self.last_stats()
            print("gradient %s l2_norm %s" % (actor.namespace, st[1]))
            print("gradient %s l2_norm %s" % (self.nets[1].namespace, st[2]))
        return True

    def device_batch_for(self, batch, state_only=False):
        self.flush()          # (a deferred actor update gathers ITS draw into the shared minibatch buffer: before this one's, not after)
        if isinstance(batch, replay_memory.StateColumn):          # a state column of a replay Batch: its rows, gathered on the device
            batch = batch.batch
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
        if self._h:
            self.flush()
        for b in self._upload.values():
            b.close()
        if self._h:
            lib.cpp_ddpg_destroy(self._h)
            self._h = None


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
        if isinstance(batch, replay_memory.Batch) and trainer.train_pair(batch):
            return                  # ran together with the actor's deferred update (ActorNetwork.train)
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
        # (--exact-products asks for the exact arithmetic contract; without the flag the context keeps whatever mode its owner chose --
        # an explicit Context.set_precision("exact") is not undone, and a second agent on the shared context does not fight the first)
        if getattr(opts, "exact_products", False) and _lib.default_context().precision != "exact":
            _lib.default_context().set_precision("exact")
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
        from . import distributed
        return distributed.setup_data_parallel(self, opts, batch_size)

    def _action(self, state, add_noise):
        """action_given under the context lock of --async-rollouts (the rollout thread and the learner thread share one stream)."""
        lock = getattr(self, "device_lock", None)
        if lock is None:
            return self.actor.action_given(state, add_noise)
        with lock:
            return self.actor.action_given(state, add_noise)

    def _train_once(self, batch_size, batches_per_step):
        """the inner step ddpg_cartpole.py:331-337; returns the losses it logs (B10: the last minibatch's TD loss)."""
        if opts.host_rng_sampling:
            for _ in range(batches_per_step):
                batch = self.replay_memory.batch(batch_size)
                self.actor.train(batch.state_1)
                self.critic.train(batch)
            self.target_actor.update_weights()
            self.target_critic.update_weights()
        else:
            self.train_step(batch_size, batches_per_step)
        return [float(self.trainer.last_stats()[0])]

    def _verbose_after_train(self, batch_size):
        if VERBOSE_DEBUG:                          # ddpg_cartpole.py:339-349
            batch = self.replay_memory.batch(batch_size)
            td_loss, td, q_value = self.critic.check_loss(batch)
            print("-----")
            print("temporal_difference_loss", td_loss)
            print("temporal_difference", td.T)
            print("q_value", q_value.T)

    def run_training(self, max_num_actions, max_run_time, batch_size, batches_per_step, saver_util):
        """ddpg_cartpole.py:291-383.  The loop itself -- episode, add_episode, train after burn-in, STATS, checkpoint, eval every 10th,
        exit tests -- is training_loop.TrainingLoop, shared with the NAF agent; under --data-parallel it takes the train / stop
        decisions collectively, with --async-rollouts the episodes come from a rollout thread."""
        from . import training_loop
        agreement = None
        if getattr(opts, "data_parallel", False):
            agreement = self._dp_learner(batch_size).agreement()
        if getattr(opts, "async_rollouts", False) and getattr(self, "device_lock", None) is None:
            self.device_lock = training_loop.FairLock()

        def dump_requested():
            global DUMP_WEIGHTS
            if DUMP_WEIGHTS:
                DUMP_WEIGHTS = False
                return True
            return False
        loop = training_loop.TrainingLoop(self, opts, act=lambda s: self._action(s, True), train=self._train_once,
                                          agreement=agreement, verbose=lambda: VERBOSE_DEBUG,
                                          after_train=self._verbose_after_train, dump_weights_requested=dump_requested)
        loop.run(max_num_actions, max_run_time, batch_size, batches_per_step, saver_util)
        return loop

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
                action = self._action(state, add_noise)
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
    if opts.data_parallel and opts.host_rng_sampling:
        # the host-RNG path is the reference's literal loop: local actor.train / critic.train calls with no all-reduce -- N learners
        # would agree on when to train (LoopAgreement) and silently train N different networks
        raise SystemExit("--data-parallel draws minibatches with the device sampler inside the collective step: it cannot be combined with --host-rng-sampling")
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
    if opts.data_parallel and opts.num_eval <= 0:
        # process group, rank 0's parameters to every rank, RCCL communicator: collective, so at the same point on every rank
        from . import distributed
        distributed.setup_data_parallel(agent, opts, opts.batch_size)
    if opts.num_eval > 0:
        agent.run_eval(opts.num_eval, opts.eval_action_noise)
    else:
        agent.run_training(opts.max_num_actions, opts.max_run_time, opts.batch_size,
                           opts.batches_per_step, saver_util)
        if saver_util is not None:
            saver_util.force_save()
    env.reset()
    agent.close()
    if opts.data_parallel:
        from . import distributed
        distributed.shutdown_data_parallel()
    if hasattr(env, "close"):
        env.close()


if __name__ == "__main__":
    main()
