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
        self.event_log = None
        if getattr(opts, "event_log_out", None):        # bullet_cartpole.py:90-94: every episode goes to the log as it is played
            from .event_log import EventLog
            self.event_log = EventLog(opts.event_log_out, self.use_raw_pixels)

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
        state = self._obs()
        if self.event_log:                              # :283-285
            self.event_log.reset()
            self.event_log.add_just_state(state)
        return state

    def step(self, action):
        assert np.asarray(action).shape == (1, 2)
        self.steps += 1
        done = self.steps >= self.episode_len
        state = self._obs()
        if self.event_log:                              # :221-222
            self.event_log.add(state, action, 1.0)
        return state, 1.0, done, {}

    def close(self):
        if self.event_log:
            ep = self.event_log.episode_entry
            if ep is not None and len(ep.event) <= 1:   # only an initial state (main()'s closing env.reset()): nothing to keep
                self.event_log.episode_entry = None
            self.event_log.close()                      # writes the episode in progress
            self.event_log = None
