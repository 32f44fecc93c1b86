"""Host-side helpers with the names of the reference's util.py (flags, stopwatch, exploration noise).

Reference: /root/reference/util.py -- add_opts :10-20, StopWatch :22-26, clip_and_debug_gradients
:45-58 (here: the clip value is handed to the fused clip+SGD kernel, see ddpg_cartpole.py),
collapsed_successive_ranges :60-71, OrnsteinUhlenbeckNoise :134-156.  Checkpoint / png helpers are out
of scope.
"""
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
