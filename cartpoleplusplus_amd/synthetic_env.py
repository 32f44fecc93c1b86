"""Random-frame stand-in for the reference's pybullet environment (bullet_cartpole.py), for smoke
tests and examples only.  Physics / rendering are NOT part of the accelerated path (they stay on the
host CPU behind the reference's gym interface); this class only reproduces the observation / action
shapes and dtypes that interface hands to the agent:

  pixels : (H, W, 3, num_cameras, action_repeats) float32 holding f16(k/255) values
           (bullet_cartpole.py:117-122, :239-243)
  low-dim: (action_repeats, 2, 7) float32 poses (bullet_cartpole.py:123-128)
  action : (1, 2) in [-1, 1] (bullet_cartpole.py:87)
"""
import numpy as np


class _Space(object):
    def __init__(self, shape):
        self.shape = tuple(shape)


class SyntheticCartpole(object):
    def __init__(self, opts, seed=0):
        self.rng = np.random.RandomState(seed)
        self.max_episode_len = int(opts.max_episode_len)
        self.use_raw_pixels = bool(opts.use_raw_pixels)
        if self.use_raw_pixels:
            shape = (opts.render_height, opts.render_width, 3, opts.num_cameras, opts.action_repeats)
        else:
            shape = (opts.action_repeats, 2, 7)
        self.observation_space = _Space(shape)
        self.action_space = _Space((1, 2))
        self.steps = 0

    def _obs(self):
        shape = self.observation_space.shape
        if self.use_raw_pixels:
            k = self.rng.randint(0, 256, size=shape).astype(np.float16)
            k /= 255
            return k.astype(np.float32)
        return self.rng.standard_normal(shape).astype(np.float32)

    def reset(self):
        self.steps = 0
        self.episode_len = int(self.rng.randint(5, self.max_episode_len + 1))
        return self._obs()

    def step(self, action):
        assert np.asarray(action).shape == (1, 2)
        self.steps += 1
        done = self.steps >= self.episode_len
        return self._obs(), 1.0, done, {}
