#!/usr/bin/env python
"""NAF agent for cartpole++ with the interface of the reference's naf_cartpole.py, on the same HIP
kernels as the DDPG path (conv trunk, MFMA MLP heads, fused clip + optimiser) plus a small fused NAF head
kernel (L matrix, advantage, TD loss and their gradients).

Reference surface kept (paths relative to /root/reference/naf_cartpole.py): flags :18-67, ValueNetwork
:93-114 (`value`, `input_state_representation`, `value_given`), NafNetwork :117-284 (`action_given`,
`train(batch) -> loss`, `debug_values`, attrs `input_state`, `input_action`, `output_action`, `q_value`,
`advantage`, `loss`, `train_op`, `exploration_noise`), NormalizedAdvantageFunctionAgent :287-456.
tf.check_numerics (:242-245) becomes a device flag: `train` raises FloatingPointError when l_values, L or
the loss is not finite.
"""
import argparse
import collections
import os
import ctypes as C
import sys
import time

import numpy as np

from . import _lib, base_network, replay_memory, util
from ._lib import lib, check, ptr

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
    a('--max-num-actions', type=int, default=0, help="train for (at least) this number of actions")
    a('--max-run-time', type=int, default=0, help="train for (at least) this number of seconds")
    a('--ckpt-dir', type=str, default=None, help="if set save ckpts to this dir")
    a('--ckpt-freq', type=int, default=3600, help="freq (sec) to save ckpts")
    a('--batch-size', type=int, default=128, help="training batch size")
    a('--batches-per-step', type=int, default=5, help="number of batches to train per step")
    a('--dont-do-rollouts', action="store_true", help="train from the replay memory only")
    a('--target-update-rate', type=float, default=0.0001, help="affine combo for updating target networks")
    a('--share-input-state-representation', action='store_true',
      help="one network for processing input state shared between value, l_value and output_action networks")
    a('--hidden-layers', type=str, default="100,50", help="hidden layer sizes")
    a('--use-batch-norm', action='store_true', help="whether to use batch norm on conv layers")
    a('--discount', type=float, default=0.99, help="discount for RHS of bellman equation update")
    a('--event-log-in', type=str, default=None, help="prepopulate replay memory from this event log")
    a('--replay-memory-size', type=int, default=22000, help="max size of replay memory")
    a('--replay-memory-burn-in', type=int, default=1000, help="dont train until replay memory reaches this size")
    a('--eval-action-noise', action='store_true', help="whether to use noise during eval")
    a('--action-noise-theta', type=float, default=0.01, help="OrnsteinUhlenbeckNoise theta")
    a('--action-noise-sigma', type=float, default=0.05, help="OrnsteinUhlenbeckNoise sigma")
    util.add_opts(parser)
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
    a('--gpu-mem-fraction', type=float, default=None,
      help="accepted for command-line compatibility and ignored (a TensorFlow session option in the reference: "
           "the replay memory and the networks size their own HBM allocations here)")
    a('--host-rng-sampling', action='store_true', help="draw minibatch rows with numpy's RNG like the reference")
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
      help="play the episodes on a rollout thread while the learner trains back to back (training_loop.py)")
    a('--synthetic-env', action='store_true', help="random-frame stand-in env")
    return parser


def default_opts(**overrides):
    o = build_parser().parse_args([])
    for k, v in overrides.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


opts = default_opts()


def set_opts(o):
    global opts
    opts = o


class _Op(object):
    def __init__(self, name):
        self.name = name


def _forward_host(net, states, n_out):
    s, dt = _lib.as_state_array(states)
    B = s.shape[0]
    out = np.empty((B, n_out), np.float32)
    check(lib.cpp_net_forward(net.handle, ptr(s), dt, B, None, ptr(out)))
    return out


class ValueNetwork(base_network.Network):
    """ Value network component of a NAF network. Created as seperate net because it has a target network."""

    def __init__(self, namespace, input_state, hidden_layer_config):
        super(ValueNetwork, self).__init__(namespace)
        self.input_state = input_state
        opts.hidden_layers = hidden_layer_config
        # exposed since it is the network "shared" by the l_value & output_action heads when running
        # --share-input-state-representation
        self.input_state_representation = self.input_state_network(input_state, opts)
        self._build_native(_lib.CPP_HEAD, 1, max(int(opts.batch_size), 1), head_out=1, head_act=0)
        self.value = _Op(namespace + "/fc")

    def value_given(self, state):
        return _forward_host(self, state, 1)


class _HeadNetwork(base_network.Network):
    """'naf/output_action' or 'naf/l_values': its own state network + 'fc' head, or (shared mode) just the
    'fc' head on top of the value network's representation (naf_cartpole.py:149-161, :174-184)."""

    def __init__(self, namespace, input_state, value_net, head_out, head_act):
        super(_HeadNetwork, self).__init__(namespace)
        if opts.share_input_state_representation:
            self._state_elems = int(value_net.input_state_representation.get_shape()[1])
            self._hidden, self._conv_input = [], None
        else:
            self.input_state_network(input_state, opts)
        self._build_native(_lib.CPP_HEAD, head_out, max(int(opts.batch_size), 1), head_out=head_out, head_act=head_act)


class DeferredLoss(object):
    """the loss `naf.train(batch)` returns for a replay draw: a number that is fetched from the device when somebody looks at it
    (float(), arithmetic, comparison, printing).  The reference's loop only appends the losses to a list and logs their mean
    (naf_cartpole.py:369-371, :377): waiting for every minibatch's loss before launching the next would leave the GPU idle for a
    host round trip per minibatch.  A non-finite minibatch (tf.check_numerics, :242-245) raises FloatingPointError when its loss is
    looked at, and at the latest two train calls later; its optimiser step never ran (the device checks the flag itself)."""
    __slots__ = ("_net", "_ticket", "_value", "_error", "__weakref__")

    def __init__(self, net, ticket):
        self._net, self._ticket, self._value, self._error = net, int(ticket), None, None

    def resolve(self):
        if self._value is None and self._error is None:
            loss = C.c_float()
            rc = lib.cpp_naf_loss_wait(self._net.handle, self._ticket, C.byref(loss))
            self._value = float(loss.value)
            if rc == 4:
                self._error = FloatingPointError(lib.cpp_last_error().decode())
            elif rc:
                self._error = RuntimeError(lib.cpp_last_error().decode())
        if self._error is not None:
            raise self._error
        return self._value

    def __float__(self):
        return self.resolve()

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.resolve(), dtype=dtype or np.float32)

    def __repr__(self):
        return repr(self.resolve())

    def __format__(self, spec):
        return format(self.resolve(), spec)


def _deferred_op(name):
    def op(self, *args):
        return getattr(self.resolve(), name)(*[float(a) if isinstance(a, DeferredLoss) else a for a in args])
    op.__name__ = name
    return op


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__neg__", "__abs__",
           "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__", "__pow__", "__rpow__", "__bool__", "__int__", "__round__"):
    setattr(DeferredLoss, _n, _deferred_op(_n))
DeferredLoss.__hash__ = None


class NafNetwork(base_network.Network):
    def __init__(self, namespace, input_state, input_state_2, value_net, target_value_net, action_dim):
        super(NafNetwork, self).__init__(namespace)
        self.exploration_noise = util.OrnsteinUhlenbeckNoise(action_dim, opts.action_noise_theta,
                                                             opts.action_noise_sigma)
        self.value_net, self.target_value_net = value_net, target_value_net
        self.input_state, self.input_state_2 = input_state, input_state_2
        self.action_dim = int(action_dim)
        self.input_action = base_network.Placeholder([None, action_dim], name="input_action")
        self.reward = base_network.Placeholder([None, 1], name="reward")
        self.terminal_mask = base_network.Placeholder([None, 1], name="terminal_mask")
        self.ctx = value_net.ctx
        num_l_values = (action_dim * (action_dim + 1)) // 2
        self.mu_net = _HeadNetwork(namespace + "/output_action", input_state, value_net, action_dim, 2)
        self.l_net = _HeadNetwork(namespace + "/l_values", input_state, value_net, num_l_values, 0)
        self.output_action = _Op(namespace + "/output_action/fc")
        self._l_values = _Op(namespace + "/l_values/fc")
        self.advantage, self.q_value = _Op(namespace + "/advantage"), _Op(namespace + "/q_value")
        self.loss, self.train_op = _Op(namespace + "/loss"), _Op(namespace + "/optimiser/train_op")
        kind, lr, mom, b1, b2, eps = util.construct_optimiser(opts)              # :233
        hp = _lib.NafHyper(float(opts.discount), util.gradient_clip_value(opts), float(opts.target_update_rate),
                           kind, lr, mom, b1, b2, eps)
        h = C.c_void_p()
        check(lib.cpp_naf_create(self.ctx.handle, value_net.handle, target_value_net.handle, self.mu_net.handle,
                                 self.l_net.handle, 1 if opts.share_input_state_representation else 0,
                                 C.byref(hp), C.byref(h)))
        self.handle = h
        self._state_elems = int(np.prod([int(d) for d in input_state.get_shape()[1:]]))
        self._upload = {}

    def trainable_model_vars(self):
        return self.mu_net.trainable_model_vars() + self.l_net.trainable_model_vars()

    def initialise_variables(self, rng=None):
        self.mu_net.initialise_variables(rng)
        self.l_net.initialise_variables(rng)

    def _device_batch(self, batch):
        if isinstance(batch, replay_memory.Batch) and batch.device is not None:
            return batch.device
        if isinstance(batch, replay_memory.Batch):
            raise RuntimeError("empty batch")
        B = np.asarray(batch.state_1).shape[0]
        if B not in self._upload:
            self._upload[B] = replay_memory.DeviceBatch(B, self._state_elems, self.action_dim, self.ctx)
        return self._upload[B].upload(batch.state_1, batch.action, batch.reward, batch.terminal_mask, batch.state_2)

    def forward(self, states):
        s, dt = _lib.as_state_array(states)
        out = np.empty((s.shape[0], self.action_dim), np.float32)
        check(lib.cpp_naf_action(self.handle, ptr(s), dt, s.shape[0], ptr(out)))
        return out

    def action_given(self, state, add_noise):
        actions = self.forward(np.asarray(state)[None])
        if add_noise:
            actions[0] += self.exploration_noise.sample()
            actions = np.minimum(1, actions)     # np.clip(1, -1, actions): upper bound only (:259)
        return actions

    def train(self, batch):
        loss = C.c_float()
        if isinstance(batch, replay_memory.Batch) and batch.in_replay():
            # a draw of the replay memory (naf_cartpole.py:367-371): the device samples those rows where they lie -- the B row
            # indexes are all that crosses PCIe -- and nobody waits for the minibatch: the loss comes back when it is looked at
            ticket = C.c_uint64()
            check(lib.cpp_naf_train_rows_async(self.handle, batch._memory.handle, len(batch.idxs), ptr(batch.idxs), C.byref(ticket)))
            out = DeferredLoss(self, ticket.value)
            pending = self.__dict__.setdefault("_losses", collections.deque())
            pending.append(out)
            while len(pending) > 2:              # (a ticket stays readable for 8 calls; a numeric error surfaces two calls late at most)
                pending.popleft().resolve()
            return out
        else:
            dev = self._device_batch(batch)
            rc = lib.cpp_naf_train(self.handle, dev.handle, C.byref(loss))
        if rc == 4:          # CPP_ERR_NUMERIC: the reference raises InvalidArgumentError from tf.check_numerics
            raise FloatingPointError(lib.cpp_last_error().decode())
        check(rc)
        return loss.value

    def debug_values(self, batch):
        dev = self._device_batch(batch)
        B = dev.size
        nl = (self.action_dim * (self.action_dim + 1)) // 2
        l_values, loss = np.empty((B, nl), np.float32), np.zeros(1, np.float32)
        v, a, vp = np.empty(B, np.float32), np.empty(B, np.float32), np.empty(B, np.float32)
        check(lib.cpp_naf_debug_values(self.handle, dev.handle, ptr(l_values), ptr(loss), ptr(v), ptr(a), ptr(vp)))
        return [np.squeeze(l_values), loss[0], v, a, vp]

    def get_grads(self):
        p, n = C.c_void_p(), C.c_int64()
        check(lib.cpp_naf_grad_buffer(self.handle, C.byref(p), C.byref(n)))
        return np.concatenate([self.value_net.get_grads(), self.mu_net.get_grads(), self.l_net.get_grads()])

    def last_stats(self):
        out = np.zeros(3, np.float32)
        check(lib.cpp_naf_last_stats(self.handle, ptr(out)))
        return out

    def get_optimiser_state(self):
        """the optimiser's slot variables {m, v, step} (what tf.train.Saver checkpoints besides the weights, util.py:88-90)."""
        n = int(lib.cpp_naf_opt_state_size(self.handle))
        m, v, step = np.empty(n, np.float32), np.empty(n, np.float32), C.c_uint64()
        check(lib.cpp_naf_get_opt_state(self.handle, ptr(m), ptr(v), n, C.byref(step)))
        return {"m": m, "v": v, "step": np.uint64(step.value)}

    def set_optimiser_state(self, state):
        m = np.ascontiguousarray(state["m"], np.float32)
        v = np.ascontiguousarray(state["v"], np.float32)
        check(lib.cpp_naf_set_opt_state(self.handle, ptr(m), ptr(v), len(m), int(state["step"])))

    def close(self):
        # a loss nobody has looked at may be a non-finite minibatch (check_numerics, naf_cartpole.py:242-245): closing does not hide
        # it -- the device is released first, then the first such error is raised
        unresolved = None
        for d in self.__dict__.get("_losses", ()):
            try:
                d.resolve()
            except FloatingPointError as e:
                unresolved = unresolved or e
            except Exception:       # noqa: BLE001 -- a ticket that has expired, a handle already gone: nothing the caller can act on
                pass
        for b in self._upload.values():
            b.close()
        if self.handle:
            lib.cpp_naf_destroy(self.handle)
            self.handle = None
        self.mu_net.close()
        self.l_net.close()
        if unresolved is not None:
            raise unresolved


class NormalizedAdvantageFunctionAgent(object):
    def __init__(self, env):
        self.env = env
        state_shape = self.env.observation_space.shape
        action_dim = self.env.action_space.shape[1]
        # (--exact-products asks for the exact arithmetic contract; without the flag the context keeps whatever mode its owner chose --
        # an explicit Context.set_precision("exact") is not undone, and a second agent on the shared context does not fight the first)
        if getattr(opts, "exact_products", False) and _lib.default_context().precision != "exact":
            _lib.default_context().set_precision("exact")
        self.replay_memory = replay_memory.ReplayMemory(opts.replay_memory_size, state_shape, action_dim,
                                                       store_dtype=opts.replay_store)
        batched_state_shape = [None] + list(state_shape)
        s1 = base_network.Placeholder(batched_state_shape)
        s2 = base_network.Placeholder(batched_state_shape)
        self.value_net = ValueNetwork("value", s1, opts.hidden_layers)
        self.target_value_net = ValueNetwork("target_value", s2, opts.hidden_layers)
        self.naf = NafNetwork("naf", s1, s2, self.value_net, self.target_value_net, action_dim)

    def initialise_variables(self, seed=None):
        rng = np.random.RandomState(seed) if seed is not None else np.random
        self.value_net.initialise_variables(rng)
        self.target_value_net.initialise_variables(rng)
        self.naf.initialise_variables(rng)

    def networks(self):
        return [self.value_net, self.target_value_net, self.naf.mu_net, self.naf.l_net]

    def post_var_init_setup(self):
        if opts.event_log_in:
            self.replay_memory.reset_from_event_log(opts.event_log_in)
        self.target_value_net.set_as_target_network_for(self.value_net, opts.target_update_rate)

    def train_step(self, batch_size, batches_per_step, idxs=None):
        """the inner step naf_cartpole.py:367-373 as one device-side sequence (hipGraph after the first call)."""
        if idxs is None and getattr(opts, "data_parallel", False):      # one learner of N: the collective step (distributed.py)
            self._dp_learner(batch_size).train_step(batches_per_step)
            return
        rows = None
        if idxs is not None:
            rows = np.ascontiguousarray(np.asarray(idxs).reshape(-1), dtype=np.int32)
        self.replay_memory.stats[">batch"] += batches_per_step
        check(lib.cpp_naf_train_step(self.naf.handle, self.replay_memory.handle, int(batch_size),
                                     int(batches_per_step), ptr(rows), int(opts.sample_seed)))

    def _dp_learner(self, batch_size):
        from . import distributed
        return distributed.setup_data_parallel(self, opts, batch_size)

    def _action(self, state, add_noise):
        """action_given under the context lock of --async-rollouts (the rollout thread and the learner thread share one stream)."""
        lock = getattr(self, "device_lock", None)
        if lock is None:
            return self.naf.action_given(state, add_noise)
        with lock:
            return self.naf.action_given(state, add_noise)

    def _train_once(self, batch_size, batches_per_step):
        """the inner step naf_cartpole.py:365-373; returns the losses it logs."""
        losses = []
        if opts.host_rng_sampling:
            for _ in range(batches_per_step):
                batch_start = time.time()
                batch = self.replay_memory.batch(batch_size)
                losses.append(self.naf.train(batch))
                print("batch_took", time.time() - batch_start)
            self.target_value_net.update_weights()
        else:
            batch_start = time.time()
            self.train_step(batch_size, batches_per_step)
            st = self.naf.last_stats()
            if st[2] != 0:
                raise FloatingPointError("check_numerics: non-finite l_values / L / loss")
            losses.append(float(st[0]))
            print("batch_took", (time.time() - batch_start) / batches_per_step)
        return losses

    def run_training(self, max_num_actions, max_run_time, batch_size, batches_per_step, saver_util):
        """naf_cartpole.py:323-389; the loop is training_loop.TrainingLoop, shared with the DDPG agent (rank agreement under
        --data-parallel, rollout thread with --async-rollouts)."""
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
                                          agreement=agreement, verbose=lambda: VERBOSE_DEBUG, timing_prints=True,
                                          dump_weights_requested=dump_requested)
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
        for i in range(num_episodes):
            state = self.env.reset()
            total_reward, steps, done = 0, 0, False
            while not done:
                action = self._action(state, add_noise)
                state, reward, done, _ = self.env.step(action)
                print("EVALSTEP e%d s%d action=%s (l2=%s) => reward %s" % (i, steps, action, np.linalg.norm(action), reward))
                total_reward += reward
                steps += 1
            print("EVAL", i, steps, total_reward)
        sys.stdout.flush()

    def close(self):
        if getattr(self, "_learner", None) is not None:
            self._learner.close()
            self._learner = None
        self.naf.close()
        self.value_net.close()
        self.target_value_net.close()
        self.replay_memory.close()


def main(argv=None):
    _install_signal_handlers()
    set_opts(build_parser().parse_args(argv))
    if opts.data_parallel and opts.host_rng_sampling:
        # the host-RNG path is the reference's literal loop: local actor.train / critic.train calls with no all-reduce -- N learners
        # would agree on when to train (LoopAgreement) and silently train N different networks
        raise SystemExit("--data-parallel draws minibatches with the device sampler inside the collective step: it cannot be combined with --host-rng-sampling")
    sys.stderr.write("%s\n" % opts)
    from .ddpg_cartpole import make_env
    env = make_env(opts)
    agent = NormalizedAdvantageFunctionAgent(env=env)
    saver_util = None
    if opts.ckpt_dir is not None and not (opts.data_parallel and int(os.environ.get("RANK", "0")) != 0):      # rank 0 keeps the checkpoints
        saver_util = util.SaverUtil(agent, opts.ckpt_dir, opts.ckpt_freq)
    else:
        agent.initialise_variables()
    agent.post_var_init_setup()
    if opts.data_parallel and opts.num_eval <= 0:
        from . import distributed        # collective set-up at the same point on every rank (see ddpg_cartpole.main)
        distributed.setup_data_parallel(agent, opts, opts.batch_size)
    if opts.num_eval > 0:
        agent.run_eval(opts.num_eval, opts.eval_action_noise)
    else:
        agent.run_training(opts.max_num_actions, opts.max_run_time, opts.batch_size, opts.batches_per_step, saver_util)
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
