"""A small pure-Python env with the gym interface (no gym / MuJoCo in the image) to exercise the host-env path:
the classic torque-limited pendulum swing-up.  obs = (cos th, sin th, th_dot), action in [-1, 1] scaled to the
+-2 N m torque limit, reward = -(th_norm^2 + 0.1 th_dot^2 + 0.001 u^2), 200-step episodes ended by a time limit."""
import numpy as np
from gym import spaces


class PendulumEnv:
    max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0
    _max_episode_steps = 200

    def __init__(self, seed=0):
        self.observation_space = spaces.Box(-np.inf, np.inf, (3,))
        self.action_space = spaces.Box(-1.0, 1.0, (1,))
        self.seed(seed)
        self.training = True

    def seed(self, seed):
        self._rng = np.random.RandomState(int(seed) % (2 ** 32))

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def close(self):
        pass

    def _obs(self):
        return np.array([np.cos(self._th), np.sin(self._th), self._thdot], dtype=np.float32)

    def reset(self):
        self._th, self._thdot = self._rng.uniform(-np.pi, np.pi), self._rng.uniform(-1.0, 1.0)
        self._t = 0
        return self._obs()

    def step(self, action):
        u = float(np.clip(np.asarray(action).reshape(-1)[0], -1.0, 1.0)) * self.max_torque
        th_norm = ((self._th + np.pi) % (2 * np.pi)) - np.pi
        cost = th_norm ** 2 + 0.1 * self._thdot ** 2 + 0.001 * u ** 2
        self._thdot = float(np.clip(self._thdot + (3 * self.g / (2 * self.l) * np.sin(self._th)
                                                  + 3.0 / (self.m * self.l ** 2) * u) * self.dt,
                                    -self.max_speed, self.max_speed))
        self._th += self._thdot * self.dt
        self._t += 1
        time_limit = self._t >= self._max_episode_steps
        return self._obs(), -cost, time_limit, {"time_limit": time_limit}
