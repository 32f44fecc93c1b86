"""Network construction and target-network plumbing with the interface of the reference's
base_network.py, backed by cpp_net_* (HIP kernels).

Reference surface kept (paths relative to /root/reference): module-global IS_TRAINING
(base_network.py:11); Network(namespace) with _create_variables_copy_op :20-33,
set_as_target_network_for :35-43, update_weights :45-49, trainable_model_vars :51-56,
hidden_layers_starting_at :58-71, simple_conv_net_on :73-127, input_state_network :129-134 and the
pool1/pool2/pool3 attributes :108,116,124.

The reference builds a TensorFlow graph; here the same calls record a network description that
`_build_native` hands to cpp_net_create.  A "placeholder" is a tiny shape-carrying object; running a
layer means calling the C ABI.
"""
import ctypes as C
import sys
import zlib

import numpy as np

from . import _lib, util
from ._lib import lib, check, ptr


class Placeholder(object):
    """stand-in for tf.placeholder: carries a shape [None, ...] and a name."""

    def __init__(self, shape, name=None, dtype=np.float32):
        self.shape, self.name, self.dtype = list(shape), name, dtype

    def get_shape(self):
        return self.shape


IS_TRAINING = Placeholder([], name="is_training", dtype=bool)      # base_network.py:11


class Layer(object):
    """symbolic result of a graph-building call (shape only)."""

    def __init__(self, shape, net=None, tag=None):
        self.shape, self.net, self.tag = list(shape), net, tag

    def get_shape(self):
        return self.shape

    def eval(self, batch_size):
        """pool activations of the owning network's last forward (render_* helpers, :136-154)."""
        which = {"pool1": 1, "pool2": 2, "pool3": 3}[self.tag]
        out = np.empty([batch_size] + self.shape[1:], np.float32)
        check(lib.cpp_net_get_pool(self.net.handle, which, batch_size, ptr(out)))
        return out


class Variable(object):
    """one entry of trainable_model_vars(): name '<namespace>/<scope>/{weights,biases}:0', shape, and
    a live view onto the flat device buffer via eval()/load()."""

    def __init__(self, net, name, shape, offset):
        self.net, self.name, self.shape, self.offset = net, name, tuple(shape), int(offset)

    def get_shape(self):
        return self.shape

    def eval(self):
        n = int(np.prod(self.shape))
        return self.net.get_params()[self.offset:self.offset + n].reshape(self.shape)

    def load(self, value):
        flat = self.net.get_params()
        n = int(np.prod(self.shape))
        flat[self.offset:self.offset + n] = np.asarray(value, np.float32).reshape(-1)
        self.net.set_params(flat)


def _forward_for_debug(net, state):
    fwd = getattr(net, "forward", None)
    if fwd is None:
        raise NotImplementedError("%s has no host-fed forward(states)" % type(net).__name__)
    fwd(np.asarray(state)[None])


class Network(object):
    """Common class for handling ops for making / updating target networks."""

    def __init__(self, namespace):
        self.namespace = namespace
        self.target_update_op = None
        self.update_weights_op = None        # the reference never initialises this
        self._handle = None
        self._before_use = None              # a deferred update of THIS network that must land before anyone touches it (see handle)
        self.ctx = None
        self._conv_input = None              # (H, W, C) once simple_conv_net_on ran
        self._hidden = []
        self._state_elems = 0

    @property
    def handle(self):
        """the cpp_net.  Every use of the network from Python goes through this attribute, so an update that was deferred (the
        actor's half of the reference's `actor.train(batch.state_1); critic.train(batch)` pair, ddpg_cartpole.py:333-334, waits
        for the critic's call to run both as one fused device sequence) is flushed before anything can observe the network."""
        hook = self._before_use
        if hook is not None:
            hook()
        return self._handle

    @handle.setter
    def handle(self, h):
        self._handle = h

    # ------------------------------------------------------------------ graph-building surface
    def hidden_layers_starting_at(self, layer, layer_sizes, opts=None):
        if not isinstance(layer_sizes, list):
            layer_sizes = [int(s) for s in str(layer_sizes).split(",")]
        assert len(layer_sizes) > 0
        # --use-dropout (base_network.py:69-70): slim.dropout after every ReLU of this stack -- only when the caller
        # passes opts (the low-dim critic does not, ddpg_cartpole.py:176-177)
        self._use_dropout = bool(opts is not None and getattr(opts, "use_dropout", False))
        self._hidden = [int(s) for s in layer_sizes]
        return Layer([None, self._hidden[-1]], self, "hidden")

    def simple_conv_net_on(self, input_layer, opts):
        # --use-batch-norm (base_network.py:74-79): slim.batch_norm after every conv (which then has no bias)
        self._use_batch_norm = bool(getattr(opts, "use_batch_norm", False))
        # state is (batch, height, width, rgb, camera_idx, repeat); rgb/camera/repeat roll up into
        # channels (base_network.py:85-90)
        shape = input_layer.get_shape()
        height, width = int(shape[1]), int(shape[2])
        num_channels = int(np.prod([int(d) for d in shape[3:]]))
        self._conv_input = (height, width, num_channels)
        sys.stderr.write("input_layer %s\n" % util.shape_and_product_of([None, height, width, num_channels]))
        h, w = height, width
        for i in (1, 2, 3):
            h, w = h // 2, w // 2
            pool = Layer([None, h, w, 10], self, "pool%d" % i)
            setattr(self, "pool%d" % i, pool)
            sys.stderr.write("pool%d %s\n" % (i, util.shape_and_product_of(pool.shape)))
        return self.pool3

    def input_state_network(self, input_state, opts):
        self._state_elems = int(np.prod([int(d) for d in input_state.get_shape()[1:]]))
        if opts.use_raw_pixels:
            input_state = self.simple_conv_net_on(input_state, opts)
        return self.hidden_layers_starting_at(input_state, opts.hidden_layers, opts)

    # ------------------------------------------------------------------ native instantiation
    def _build_native(self, kind, action_dim, max_batch, ctx=None, head_out=0, head_act=0):
        self.ctx = ctx or _lib.default_context()
        spec = _lib.NetSpec()
        spec.kind, spec.action_dim = kind, int(action_dim)
        spec.head_out, spec.head_act = int(head_out), int(head_act)
        spec.use_batch_norm = int(bool(getattr(self, "_use_batch_norm", False)))
        spec.use_dropout = int(bool(getattr(self, "_use_dropout", False)))
        spec.dropout_seed = zlib.crc32(self.namespace.encode()) & 0xffffffff     # one mask stream per network
        if self._conv_input is not None:
            spec.pixel, (spec.H, spec.W, spec.C) = 1, self._conv_input
        else:
            spec.pixel, spec.state_elems = 0, int(self._state_elems)
        spec.n_hidden = len(self._hidden)
        for i, h in enumerate(self._hidden):
            spec.hidden[i] = h
        handle = C.c_void_p()
        check(lib.cpp_net_create(self.ctx.handle, C.byref(spec), int(max_batch), C.byref(handle)))
        self.handle, self.max_batch, self.spec = handle, int(max_batch), spec
        self.num_params = int(lib.cpp_net_num_params(handle))

    def get_params(self):
        out = np.empty(self.num_params, np.float32)
        check(lib.cpp_net_get_params(self.handle, ptr(out), self.num_params))
        return out

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        check(lib.cpp_net_set_params(self.handle, ptr(flat), len(flat)))

    def get_grads(self):
        out = np.empty(self.num_params, np.float32)
        check(lib.cpp_net_get_grads(self.handle, ptr(out), self.num_params))
        return out

    def initialise_variables(self, rng=None):
        """tf.initialize_all_variables() for this namespace (ddpg_cartpole.py:424): slim defaults --
        xavier-uniform weights, zero biases; the actor head uses U(-1e-3, 1e-3) (ddpg_cartpole.py:94)."""
        rng = rng or np.random
        flat = np.zeros(self.num_params, np.float32)
        for v in self.trainable_model_vars():
            n = int(np.prod(v.shape))
            if v.name.endswith("biases:0") or v.name.endswith("BatchNorm/beta:0"):      # zeros_initializer both
                continue
            # the tanh action heads use U(-1e-3, 1e-3): 'actor/output_action/weights' (ddpg_cartpole.py:94) and
            # 'naf/output_action/fc/weights' (naf_cartpole.py:155) -- not the state networks beneath them
            if "/output_action/" in v.name and v.name.split("/output_action/")[1].split("/")[0] in ("weights:0", "fc"):
                vals = rng.uniform(-0.001, 0.001, n)
            else:
                if len(v.shape) == 4:
                    fan_in, fan_out = v.shape[0] * v.shape[1] * v.shape[2], v.shape[0] * v.shape[1] * v.shape[3]
                else:
                    fan_in, fan_out = v.shape
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                vals = rng.uniform(-lim, lim, n)
            flat[v.offset:v.offset + n] = vals
        self.set_params(flat)

    # ------------------------------------------------------------------ target-network plumbing
    def _create_variables_copy_op(self, source_network, affine_combo_coeff):
        """an op that moves every variable of this namespace towards its namesake in the source
        namespace: target.assign_sub(coeff * (target - source))  (base_network.py:31)"""
        assert affine_combo_coeff >= 0.0 and affine_combo_coeff <= 1.0
        mine = [(v.name.split("/", 1)[1], v.shape) for v in self.trainable_model_vars()]
        theirs = [(v.name.split("/", 1)[1], v.shape) for v in source_network.trainable_model_vars()]
        assert mine == theirs, "target and source networks differ"           # base_network.py:30
        coeff = float(affine_combo_coeff)
        return lambda: check(lib.cpp_net_soft_update(self.handle, source_network.handle, coeff))

    def set_as_target_network_for(self, source_network, target_update_rate):
        """Create an op that will update this networks weights based on a source_network"""
        # one off: initial target network is a "copy" (coeff 1.0) of the source network ...
        self._create_variables_copy_op(source_network, 1.0)()
        # ... then the op run during training
        self.update_weights_op = self._create_variables_copy_op(source_network, target_update_rate)

    def update_weights(self):
        """called during training to update target network."""
        if self.update_weights_op is None:
            raise Exception("not a target network? or set_source_network not yet called")
        return self.update_weights_op()

    def trainable_model_vars(self):
        out = []
        name = C.create_string_buffer(128)
        for i in range(lib.cpp_net_num_vars(self.handle)):
            rank, shape, off = C.c_int(), (C.c_int * 4)(), C.c_int64()
            check(lib.cpp_net_var_info(self.handle, i, name, 128, C.byref(rank), shape, C.byref(off)))
            out.append(Variable(self, "%s/%s:0" % (self.namespace, name.value.decode()),
                                [shape[k] for k in range(rank.value)], off.value))
        return out

    # ---- debug renderings (base_network.py:136-154) ----------------------------------------------------------------------------
    def render_convnet_activations(self, activations, filename_base):
        from . import util
        activations = np.array(activations, np.float32)
        _batch, height, width, num_filters = activations.shape
        for f_idx in range(num_filters):
            single_channel = activations[0, :, :, f_idx]
            peak = np.max(single_channel)
            single_channel = single_channel / peak if peak > 0 else single_channel
            util.write_img_to_png_file(np.repeat(single_channel[:, :, None], 3, axis=2), "%s_f%02d.png" % (filename_base, f_idx))

    def render_all_convnet_activations(self, step, input_state_placeholder, state):
        """the three pooled activations of ONE state (inference mode), one grey PNG per filter under /tmp"""
        _forward_for_debug(self, state)
        filename_base = "/tmp/activation_s%03d" % step
        for k, pool in enumerate((self.pool1, self.pool2, self.pool3)):
            self.render_convnet_activations(pool.eval(1), filename_base + "_p%d" % k)

    def close(self):
        if self._handle:
            h = self.handle               # (flushes a deferred update first)
            lib.cpp_net_destroy(h)
            self._handle = None
