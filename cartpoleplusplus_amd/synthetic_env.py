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


class RasterCartpole(SyntheticCartpole):
    """A cart and a pole drawn by a small software rasteriser: frames with the STRUCTURE of the reference's pybullet renders
    (bullet_cartpole.py:227-257) -- which i.i.d. pixel noise has none of:

      * flat sky / ground levels (large regions of identical pixels: exact pooling ties, near-constant channels),
      * a cart rectangle and a pole line that move a few pixels between the R action-repeat frames of one state
        (bullet_cartpole.py:185-206: one render per repeat), so the R frames of a camera are nearly identical,
      * camera 1 looks along the track (:228: the 90-degree side camera) and sees the cart change size, not position; with
        `blind_camera=True` it sees nothing but one background colour: 3 R channels of ZERO variance, the whitening scale of
        base_network.py:95-99 at its maximum rsqrt(1e-6) = 1000 (`glint` > 0: a rare single off-colour pixel in that view, i.e.
        channels that are ALMOST constant -- scales of several hundred on values that do not cancel exactly),
      * state_2 of a transition is state_1 of the next one (the env returns np.copy(self.state) each step).

    Still a stand-in (a point-mass cart with a damped pole, no contact physics) for tests and smoke runs; the accelerated path never looks at how a frame was made.  Pixel values are f16(k/255) of 8-bit colours,
    exactly what render_rgb produces (:239-243)."""
    SKY, GROUND, CART, POLE, BLIND = (135, 206, 235), (96, 96, 96), (200, 30, 30), (240, 200, 40), (222, 222, 222)

    def __init__(self, opts, seed=0, blind_camera=False, substeps=2, glint=0.0):
        SyntheticCartpole.__init__(self, opts, seed)
        assert self.use_raw_pixels, "RasterCartpole renders pixels"
        self.H, self.W = int(opts.render_height), int(opts.render_width)
        self.C, self.R = int(opts.num_cameras), int(opts.action_repeats)
        self.blind_camera, self.substeps = bool(blind_camera), int(substeps)
        self.glint = float(glint)      # probability that a blind camera's frame has ONE off-colour pixel: a nearly constant channel
        self.yy, self.xx = np.mgrid[0:self.H, 0:self.W].astype(np.float64)
        self.lut = (np.arange(256) / 255.0).astype(np.float16).astype(np.float32)       # f16(k/255), as f32
        self.state = np.zeros(self.observation_space.shape, np.float32)
        self._reset_pose()

    def _reset_pose(self):
        self.cx, self.cz = self.rng.uniform(-0.3, 0.3), self.rng.uniform(-0.3, 0.3)      # cart on the ground plane (x, depth)
        self.vx, self.vz = 0.0, 0.0
        self.th, self.om = self.rng.uniform(-0.15, 0.15), 0.0                          # pole angle from vertical, rad

    def _advance(self, action):
        a = np.asarray(action, np.float64).reshape(-1)
        dt = 0.02
        for _ in range(self.substeps):
            self.vx += dt * (4.0 * a[0] - 0.5 * self.vx)
            self.vz += dt * (4.0 * a[1] - 0.5 * self.vz)
            self.cx = float(np.clip(self.cx + dt * self.vx, -0.9, 0.9))
            self.cz = float(np.clip(self.cz + dt * self.vz, -0.9, 0.9))
            self.om += dt * (9.0 * np.sin(self.th) - 1.5 * a[0] * np.cos(self.th) - 0.2 * self.om)
            self.th += dt * self.om

    def render_u8(self, camera_idx):
        """(H, W, 3) uint8: what p.renderImage hands back, minus alpha."""
        H, W = self.H, self.W
        img = np.empty((H, W, 3), np.uint8)
        if camera_idx == 1 and self.blind_camera:
            img[:] = self.BLIND
            if self.glint > 0.0 and self.rng.uniform() < self.glint:
                img[self.rng.randint(0, H), self.rng.randint(0, W)] = self.POLE
            return img
        horizon = int(round(0.55 * H))
        img[:horizon], img[horizon:] = self.SKY, self.GROUND
        # camera 0 sees the cart move sideways with x; camera 1 (along the track) sees it move sideways with depth and grow with x
        lateral, toward = (self.cx, self.cz) if camera_idx == 0 else (self.cz, self.cx)
        size = 1.0 + 0.35 * toward
        ccx, ccy = (0.5 + 0.4 * lateral) * W, 0.70 * H
        hw, hh = 0.11 * W * size, 0.05 * H * size
        cart = (np.abs(self.xx + 0.5 - ccx) <= hw) & (np.abs(self.yy + 0.5 - ccy) <= hh)
        img[cart] = self.CART
        # the pole: a segment from the top of the cart, length 0.45 H, half-thickness ~1 px
        lean = self.th if camera_idx == 0 else 0.3 * self.th
        x0, y0 = ccx, ccy - hh
        x1, y1 = x0 + 0.45 * H * size * np.sin(lean), y0 - 0.45 * H * size * np.cos(lean)
        px, py = self.xx + 0.5 - x0, self.yy + 0.5 - y0
        dx, dy = x1 - x0, y1 - y0
        t = np.clip((px * dx + py * dy) / (dx * dx + dy * dy), 0.0, 1.0)
        dist = np.hypot(px - t * dx, py - t * dy)
        img[dist <= max(0.9, 0.016 * W * size)] = self.POLE
        return img

    def _render_repeat(self, repeat):
        for cam in range(self.C):
            self.state[:, :, :, cam, repeat] = self.lut[self.render_u8(cam)]

    def _obs(self):
        return np.copy(self.state)

    def reset(self):
        self._reset_pose()
        for r in range(self.R):                          # bullet_cartpole.py:276-281: the reset state is R renders of the rest pose
            self._render_repeat(r)
        return SyntheticCartpole.reset(self)

    def step(self, action):
        for r in range(self.R):                          # :185-206: physics, then one render per repeat
            self._advance(action)
            self._render_repeat(r)
        state, reward, done, info = SyntheticCartpole.step(self, action)
        done = bool(done or abs(self.th) > 0.6)
        return state, reward, done, info


def play_episodes(env, rows, rng, max_len=50):
    """[(initial_state, [(action (1, 2), reward, state_2), ...]), ...] with `rows` transitions in all: random-policy episodes of
    `env`, in the form ReplayMemory.add_episode takes them (ddpg_cartpole.py:315-326)."""
    out, left = [], int(rows)
    while left > 0:
        first, seq = env.reset(), []
        for _ in range(min(left, max_len)):
            a = rng.uniform(-1, 1, (1, 2)).astype(np.float32)
            s2, r, done, _ = env.step(a)
            seq.append((a, r, s2))
            if done:
                break
        out.append((first, seq))
        left -= len(seq)
    return out
