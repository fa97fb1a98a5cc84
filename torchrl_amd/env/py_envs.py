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


class CartPoleEnv:
    """The classic cart-pole balancing task (Barto, Sutton & Anderson): 4 observations, 2 actions, reward 1 per step,
    terminated when the pole passes 12 degrees or the cart leaves +-2.4, 200-step time limit."""
    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    theta_limit, x_limit = 12 * 2 * np.pi / 360, 2.4
    _max_episode_steps = 200

    def __init__(self, seed=0):
        self.observation_space = spaces.Box(-np.inf, np.inf, (4,))
        self.action_space = spaces.Discrete(2)
        self.seed(seed)

    def seed(self, seed):
        self._rng = np.random.RandomState(int(seed) % (2 ** 32))

    def train(self):
        pass

    def eval(self):
        pass

    def close(self):
        pass

    def reset(self):
        self._s = self._rng.uniform(-0.05, 0.05, size=4)
        self._t = 0
        return self._s.astype(np.float32)

    def step(self, action):
        x, x_dot, th, th_dot = self._s
        force = self.force_mag if int(action) == 1 else -self.force_mag
        total, pml = self.masscart + self.masspole, self.masspole * self.length
        tmp = (force + pml * th_dot ** 2 * np.sin(th)) / total
        th_acc = (self.gravity * np.sin(th) - np.cos(th) * tmp) / \
            (self.length * (4.0 / 3.0 - self.masspole * np.cos(th) ** 2 / total))
        x_acc = tmp - pml * th_acc * np.cos(th) / total
        self._s = np.array([x + self.tau * x_dot, x_dot + self.tau * x_acc, th + self.tau * th_dot, th_dot + self.tau * th_acc])
        self._t += 1
        fell = abs(self._s[0]) > self.x_limit or abs(self._s[2]) > self.theta_limit
        time_limit = self._t >= self._max_episode_steps and not fell
        return self._s.astype(np.float32), 1.0, bool(fell or time_limit), {"time_limit": bool(time_limit)}
