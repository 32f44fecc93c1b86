"""Host-side helpers with the names of the reference's util.py (flags, stopwatch, exploration noise).

Reference: /root/reference/util.py -- add_opts :10-20, StopWatch :22-26, clip_and_debug_gradients
:45-58 (here: the clip value is handed to the fused clip+SGD kernel, see ddpg_cartpole.py),
collapsed_successive_ranges :60-71, OrnsteinUhlenbeckNoise :134-156.  Checkpoint / png helpers are out
of scope.
"""
import datetime
import os
import sys
import time

import numpy as np


def add_opts(parser):
    parser.add_argument('--gradient-clip', type=float, default=5,
                        help="do global clipping to this norm")
    parser.add_argument('--print-gradients', action='store_true',
                        help="print the (pre-clip) global l2 norm of each gradient list per train op")
    parser.add_argument('--optimiser', type=str, default="GradientDescent",
                        help="optimiser name (DDPG ignores it, as in the reference)")
    parser.add_argument('--optimiser-args', type=str, default="{\"learning_rate\": 0.001}",
                        help="json serialised args for optimiser constructor")
    parser.add_argument('--use-dropout', action='store_true',
                        help="include a dropout layers after each fully connected layer")


class StopWatch(object):
    def reset(self):
        self.start = time.time()

    def time(self):
        return time.time() - self.start


def l2_norm(tensor):
    """util.py:33-35 on a host array (the training step's own norms are computed on the device)"""
    t = np.asarray(tensor, np.float64)
    return float(np.sqrt((t * t).sum()))


def standardise(tensor):
    """util.py:37-43 on a host array: (x - mean) / sqrt(mean((x - mean)^2))"""
    t = np.asarray(tensor, np.float64)
    mean = t.mean()
    return (t - mean) / np.sqrt(((t - mean) ** 2).mean())


def clip_and_debug_gradients(gradients, opts):
    """util.py:45-58 for host arrays: `gradients` is a list of (gradient, variable) pairs; the gradients are scaled by
    clip / max(global norm, clip) (tf.clip_by_global_norm; None gradients are skipped) when opts.gradient_clip is set, and their
    norms printed under opts.print_gradients.  The device step does this inside its optimiser kernel; this is the same rule for
    scripts that handle gradients on the host (net.get_grads())."""
    gradients = list(gradients)
    clip = getattr(opts, "gradient_clip", None)
    if clip is not None:
        norm = np.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g, _ in gradients if g is not None))
        scale = float(clip) / max(norm, float(clip))
        gradients = [(None if g is None else (np.asarray(g) * np.asarray(g).dtype.type(scale)), v) for g, v in gradients]
    if getattr(opts, "print_gradients", False):
        for g, v in gradients:
            if g is not None:
                print("gradient %s l2_norm [%s]" % (getattr(v, "name", v), l2_norm(g)))
    return gradients


def gradient_clip_value(opts):
    """util.py:45-50: clip_by_global_norm(grads, opts.gradient_clip) unless None.  The kernel takes
    <= 0 as 'no clipping'."""
    clip = getattr(opts, "gradient_clip", None)
    return 0.0 if clip is None else float(clip)


def construct_optimiser(opts):
    """util.py:73-76: `tf.train.<opts.optimiser>Optimizer(**json.loads(opts.optimiser_args))`.  Returns the
    (kind, learning_rate, momentum, beta1, beta2, epsilon) tuple the fused clip+apply kernel takes; the three
    optimisers the reference's experiments use are built (GradientDescent, Momentum, Adam; TF defaults)."""
    import json
    args = json.loads(opts.optimiser_args)
    lr = float(args.get("learning_rate", 0.001))
    if opts.optimiser == "GradientDescent":
        return (0, lr, 0.0, 0.0, 0.0, 0.0)
    if opts.optimiser == "Momentum":
        return (1, lr, float(args["momentum"]), 0.0, 0.0, 0.0)
    if opts.optimiser == "Adam":
        return (2, lr, 0.0, float(args.get("beta1", 0.9)), float(args.get("beta2", 0.999)),
                float(args.get("epsilon", 1e-8)))
    raise ValueError("optimiser %r not built (GradientDescent, Momentum, Adam)" % opts.optimiser)


def collapsed_successive_ranges(values):
    """[2,3,4,5,13,14,15] -> '2-5, 13-15'"""
    spans, lo, prev = [], None, None
    for v in values:
        if lo is None:
            lo = v
        elif v != prev + 1:
            spans.append((lo, prev))
            lo = v
        prev = v
    if lo is not None:
        spans.append((lo, prev))
    return ", ".join("%d-%d" % s for s in spans)


def shape_and_product_of(shape):
    dims = [d for d in shape if d is not None]
    return "%s #%s" % (tuple(shape), int(np.prod(dims)) if dims else 1)


class SaverUtil(object):
    """checkpoint save / restore with the behaviour of the reference's util.SaverUtil (util.py:88-131): on start,
    restore the latest checkpoint named in `<dir>/checkpoint` or initialise the variables and save at once;
    `save_if_required()` saves every `save_freq` seconds; `force_save()` at exit.  The replay memory is not
    checkpointed (util.py:91).  The reference hands a tf.Session to tf.train.Saver; here the first argument is
    the agent (anything with `.networks()` -> [Network] and `.initialise_variables()`), and a checkpoint is one
    `.npz` with the flat f32 buffer of every namespace (variable names and shapes stored alongside and checked
    on restore) plus, for agents whose optimiser has slot variables (NAF with Momentum / Adam: `agent.naf`), those slots
    and the update count -- tf.train.Saver saves them too (util.py:88-90), and a resumed run must not restart Adam's bias
    correction.  The `checkpoint` index file keeps TF's `model_checkpoint_path: "<name>"` line.  Both files are written to
    a temporary name and renamed, so a crash never leaves the index pointing at a partial checkpoint.  The format is this
    package's own: it is not a TensorFlow checkpoint."""

    def __init__(self, agent, ckpt_dir="/tmp", save_freq=60):
        self.agent, self.ckpt_dir = agent, ckpt_dir
        if not os.path.exists(self.ckpt_dir):
            os.makedirs(self.ckpt_dir)
        assert save_freq > 0
        self.save_freq = save_freq
        self.load_latest_ckpt_or_init_if_none()

    def _index(self):
        return "%s/checkpoint" % self.ckpt_dir

    def load_latest_ckpt_or_init_if_none(self):
        if os.path.isfile(self._index()):
            line = [l for l in open(self._index()) if l.startswith("model_checkpoint_path")][0]
            name = line.split(":", 1)[1].strip().strip('"')
            most_recent_ckpt = "%s/%s" % (self.ckpt_dir, name)
            sys.stderr.write("loading ckpt %s\n" % most_recent_ckpt)
            data = np.load(most_recent_ckpt + ".npz", allow_pickle=False)
            for net in self.agent.networks():
                layout = "|".join("%s%s" % (v.name, tuple(v.shape)) for v in net.trainable_model_vars())
                assert str(data[net.namespace + "::layout"]) == layout, "checkpoint does not match %s" % net.namespace
                net.set_params(data[net.namespace])
            opt = getattr(self.agent, "naf", None)
            if opt is not None and "optimiser::m" in data:
                opt.set_optimiser_state({"m": data["optimiser::m"], "v": data["optimiser::v"], "step": data["optimiser::step"]})
            elif opt is not None:
                sys.stderr.write("checkpoint %s holds no optimiser slots: Momentum / Adam restart from zero\n" % most_recent_ckpt)
            self.next_scheduled_save_time = time.time() + self.save_freq
        else:
            sys.stderr.write("no latest ckpt in %s, just initing vars...\n" % self.ckpt_dir)
            self.agent.initialise_variables()
            self.force_save()

    def force_save(self):
        dts = datetime.datetime.now().strftime('%Y%m%d_%H%M%S_%f')
        name = "ckpt.%s" % dts
        sys.stderr.write("saving ckpt %s/%s\n" % (self.ckpt_dir, name))
        start_time = time.time()
        blob = {}
        for net in self.agent.networks():
            blob[net.namespace] = net.get_params()
            blob[net.namespace + "::layout"] = np.array(
                "|".join("%s%s" % (v.name, tuple(v.shape)) for v in net.trainable_model_vars()))
        opt = getattr(self.agent, "naf", None)
        if opt is not None:
            for k, v in opt.get_optimiser_state().items():
                blob["optimiser::" + k] = v
        final = "%s/%s.npz" % (self.ckpt_dir, name)
        with open(final + ".tmp", "wb") as f:          # (a file object: np.savez would append ".npz" to a temporary NAME)
            np.savez(f, **blob)
        os.replace(final + ".tmp", final)
        with open(self._index() + ".tmp", "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % name)
        os.replace(self._index() + ".tmp", self._index())
        print("save_took", time.time() - start_time)
        self.next_scheduled_save_time = time.time() + self.save_freq

    def save_if_required(self):
        if time.time() >= self.next_scheduled_save_time:
            self.force_save()


class OrnsteinUhlenbeckNoise(object):
    """time correlated exploration noise (util.py:134-156), numpy f64 on the host.  Reproduces the
    reference's clip-argument-order quirk: np.clip(max, -max, state) == minimum(max, state), i.e. only
    the upper bound is enforced."""

    def __init__(self, dim, theta=0.01, sigma=0.2, max_magnitude=1.5):
        self.dim, self.theta, self.sigma, self.max_magnitude = dim, theta, sigma, max_magnitude
        self.state = np.zeros(self.dim)

    def sample(self):
        self.state += self.theta * -self.state
        self.state += self.sigma * np.random.randn(self.dim)
        self.state = np.minimum(self.max_magnitude, self.state)
        return np.copy(self.state)


# ---- debug renderings (util.py:159-194; the reference calls them from a disabled branch of naf_cartpole.run_eval) -----------------
def write_img_to_png_file(img, filename):
    """img: (height, width, 3) floats in [0, 1] (what plt.imsave takes in the reference) or uint8"""
    from .event_log import rgb_to_png
    a = np.asarray(img)
    a = a.astype(np.float64) / 255.0 if a.dtype == np.uint8 else np.nan_to_num(a.astype(np.float64))
    print("writing", filename)
    with open(filename, "wb") as f:
        f.write(rgb_to_png(a))


def render_state_to_png(step, state, split_channels=False):
    state = np.asarray(state, np.float32)
    height, width, num_channels, num_cameras, num_repeats = state.shape
    for c_idx in range(num_cameras):
        for r_idx in range(num_repeats):
            if split_channels:
                for channel in range(num_channels):
                    img = np.zeros((height, width, 3))
                    img[:, :, channel] = state[:, :, channel, c_idx, r_idx]
                    write_img_to_png_file(img, "/tmp/state_s%03d_ch%s_c%s_r%s.png" % (step, channel, c_idx, r_idx))
            else:
                write_img_to_png_file(state[:, :, 0:3, c_idx, r_idx], "/tmp/state_s%03d_c%s_r%s.png" % (step, c_idx, r_idx))


def render_action_to_png(step, action):
    """a 50 x 50 grey tile with a black line from the centre to 25 (1 + action) -- drawn without PIL"""
    img = np.full((50, 50, 3), 50, np.uint8)
    lx, ly = int(25 + (action[0][0] * 25)), int(25 + (action[0][1] * 25))
    n = max(abs(lx - 25), abs(ly - 25), 1)
    for i in range(n + 1):
        x, y = int(round(25 + (lx - 25) * i / n)), int(round(25 + (ly - 25) * i / n))
        if 0 <= x < 50 and 0 <= y < 50:
            img[y, x] = 0
    write_img_to_png_file(img, "/tmp/action_%03d.png" % step)
