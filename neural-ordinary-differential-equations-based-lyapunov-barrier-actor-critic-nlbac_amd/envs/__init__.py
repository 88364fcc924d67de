"""gym-free restatements of the reference's simulators: the same ``reset`` / ``step`` contracts,
constants and arithmetic as ``*/envs/*.py`` (numpy float64 on the host, one environment per object), on top of the
constants objects of ``nlbac_amd.envspec``.  They exist so that the whole training loop (``nlbac_amd.train``) runs
where ``gym`` is not installed; a real gym env with the same attributes is interchangeable.

    UnicycleEnv        U/envs/unicycle_env.py:57-323   (``barrier=True``: NU/envs/unicycle_env.py:105-161)
    SimulatedCarsEnv   C/envs/simulated_cars_env.py:37-180
    PvtolEnv           P/envs/pvtol_env.py:66-406       (``barrier=True``: hazard part of NP/envs/pvtol_env.py:144-220)
"""
import numpy as np

from ..envspec import PvtolSpec, SimulatedCarsSpec, UnicycleSpec
from ..synth import _pvtol_obs, _unicycle_obs


class UnicycleEnv(UnicycleSpec):
    reward_goal, goal_size, l_p = 500.0, 0.3, 0.03
    little_b, capital_b = 0.0, -20.0           # NU: barrier signal values

    def __init__(self, seed=0, barrier=False):
        super().__init__(seed)
        self.barrier = barrier
        np.random.seed(seed)
        self.reset()

    def _center(self):
        return self.state[:2] + self.l_p * np.array([np.cos(self.state[2]), np.sin(self.state[2])])

    def get_obs(self):
        return _unicycle_obs(self.state[None], self.goal_pos)[0]

    def reset(self):
        self.episode_step = 0
        self.state = np.array([-2.5, -2.5, 0.0])
        self.center = np.array([-2.47, -2.5])
        self.next_center = np.array([-2.47, -2.5])
        self.last_goal_dist = np.linalg.norm(self.goal_pos - self.next_center)
        return self.get_obs()

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        center_pos = self._center()
        th = self.state[2]
        self.state = self.state + self.dt * np.array([np.cos(th) * action[0], np.sin(th) * action[0], action[1]])
        th = self.state[2]                                       # small drag along the new heading
        self.state = self.state - self.dt * 0.1 * np.array([np.cos(th), np.sin(th), 0.0]) * np.cos(th)
        self.next_center = next_center = self._center()
        self.episode_step += 1
        info = {}
        dist_goal = np.linalg.norm(self.goal_pos - next_center)
        reward = -np.square(action[0] - 2.5) * 0.1 + (self.last_goal_dist - dist_goal) * 30
        self.last_goal_dist = dist_goal
        if dist_goal <= self.goal_size:
            info['goal_met'] = True
            reward += self.reward_goal
            done = True
        else:
            done = self.episode_step >= self.max_episode_steps
        barrier_signal = self.little_b
        d2 = np.sum((next_center - self.hazards_locations) ** 2, axis=1)
        for hit in np.nonzero(d2 < self.hazards_radius ** 2)[0]:
            barrier_signal = self.capital_b if barrier_signal == self.little_b else barrier_signal + self.capital_b
            info['num_safety_violation'] = info.get('num_safety_violation', 0) + 1
            info['safety_cost'] = info.get('safety_cost', 0.0) + (self.hazards_radius - np.sqrt(d2[hit])) / self.hazards_radius
        if self.barrier:
            return self.get_obs(), reward, dist_goal, barrier_signal, center_pos, next_center, done, info
        return self.get_obs(), reward, dist_goal, center_pos, next_center, done, info


class SimulatedCarsEnv(SimulatedCarsSpec):
    should_keep_thre, reward_goal = 0.5, 2.0

    def __init__(self, seed=0):
        super().__init__(seed)
        np.random.seed(seed)
        self.reset()

    def get_obs(self):
        o = self.state.copy()
        o[::2] /= 100.0
        o[1::2] /= 30.0
        return o

    def reset(self):
        self.t = 0.0
        self.state = np.zeros(10)
        self.state[::2] = [42.0, 34.0, 26.0, 18.0, 10.0]
        self.state[1::2] = 3.0 + np.random.normal(0, 0.5)
        self.state[7] = 3.0
        self.episode_step = 0
        return self.get_obs()

    def step(self, action):
        action = np.asarray(action, dtype=np.float64).reshape(-1)
        pos, vels = self.state[::2], self.state[1::2]
        vels_des = 3.0 * np.ones(5)
        vels_des[0] -= 4 * np.sin(self.t)
        acc = self.kp * (vels_des - vels)
        acc[1] += -self.k_brake * (pos[0] - pos[1]) * ((pos[0] - pos[1]) < 6.5)
        acc[2] += -self.k_brake * (pos[1] - pos[2]) * ((pos[1] - pos[2]) < 6.5)
        acc[3] = 0.0
        acc[4] += -self.k_brake * (pos[2] - pos[4]) * ((pos[2] - pos[4]) < 13.0)
        acc *= 1.1
        previous = self.state[4:8].copy()
        f = np.zeros(10)
        f[::2], f[1::2] = vels, acc
        f[7] = 0.0
        g = np.zeros(10)
        g[7] = 1.0
        self.state = self.state + self.dt * (f + g * action[0])
        self.t += self.dt
        self.episode_step += 1
        d34, d45 = self.state[4] - self.state[6], self.state[6] - self.state[8]
        reward = -0.5 * np.abs(action[0] ** 2) / self.max_episode_steps
        reached = int(abs(d34 - self.should_keep) < self.should_keep_thre)
        reward += self.reward_goal * reached
        info = dict(reached=reached, goal_met=False, num_safety_violation=int(d34 < 2.5) + int(d45 < 2.5),
                    safety_cost=abs(d34 - 2.5) * (d34 < 2.5) + abs(d45 - 2.5) * (d45 < 2.5))
        done = self.episode_step >= self.max_episode_steps
        return self.get_obs(), reward, abs(d34 - self.should_keep), previous, self.state[4:8].copy(), done, info


class PvtolEnv(PvtolSpec):
    reward_goal, goal_size = 1500.0, 3.5
    little_b, capital_b = 0.0, -0.1            # NP: barrier signal values

    def __init__(self, seed=0, barrier=False, **overrides):
        super().__init__(seed, **overrides)
        self.barrier = barrier
        np.random.seed(seed)
        self.reset()

    def get_obs(self):
        return _pvtol_obs(self.state[None], self.goal_pos)[0]

    def reset(self):
        self.episode_step = 0
        self.state = np.array([-4.5, -4.5, 0.0, 0.0, 0.0, 1.0, -4.5])
        self.last_goal_dist = np.linalg.norm(self.goal_pos - self.state[:2])
        return self.get_obs()

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        lya_pre_term = self.get_obs()
        x = self.state[:6].copy()
        f = np.array([x[3], x[4], 0.0, -np.sin(x[2]) * x[5], np.cos(x[2]) * x[5] - 1.0, 0.0])
        x = x + self.dt * (f + np.array([0.0, 0.0, action[1], 0.0, 0.0, action[0]]))
        op = self.state[6] + self.safety_operator_follow * (x[0] - self.state[6])
        self.state = np.concatenate((x, [op]))
        self.episode_step += 1
        info = {}
        dist_goal = np.linalg.norm(self.goal_pos - self.state[:2])
        reward = -1e-3 * dist_goal
        self.last_goal_dist = dist_goal
        if dist_goal <= self.goal_size:
            info['goal_met'] = True
            reward += self.reward_goal
            done = True
        else:
            done = self.episode_step >= self.max_episode_steps
        barrier_signal = self.little_b
        d2 = np.sum((self.state[:2] - self.hazard_locations) ** 2, axis=1)
        for hit in np.nonzero(d2 < self.hazards_radius ** 2)[0]:
            barrier_signal = self.capital_b if barrier_signal == self.little_b else barrier_signal + self.capital_b
            info['num_safety_violation_obstacles'] = info.get('num_safety_violation_obstacles', 0) + 1
            info['safety_cost_obstacles'] = info.get('safety_cost_obstacles', 0.0) + \
                (self.hazards_radius - np.sqrt(d2[hit])) / self.hazards_radius
        obs = self.get_obs()
        if self.barrier:
            return obs, reward, dist_goal, barrier_signal, lya_pre_term, obs, done, info
        return obs, reward, dist_goal, lya_pre_term, obs, done, info


def make(name, seed=0, **kw):
    """``Unicycle`` / ``SimulatedCars`` / ``Pvtol`` / ``UnicycleBarrier`` / ``PvtolBarrier``."""
    if name == "Unicycle":
        return UnicycleEnv(seed)
    if name == "UnicycleBarrier":
        return UnicycleEnv(seed, barrier=True)
    if name == "SimulatedCars":
        return SimulatedCarsEnv(seed)
    if name == "Pvtol":
        return PvtolEnv(seed, **kw)
    if name == "PvtolBarrier":
        return PvtolEnv(seed, barrier=True, **kw)
    raise Exception("Dynamics mode not supported.")
