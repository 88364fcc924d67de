"""Batched simulators on the device: ``n_envs`` independent copies of a reference environment,
advanced by ONE launch per step (``csrc/env_kernels.hip``, float64 like the reference's numpy simulators).  Same
``reset`` / ``step`` contract as the host simulators of ``nlbac_amd.envs`` with a leading batch axis and device tensors:

    obs, reward, constraint[, barrier_signal], lya_in, next_lya_in, done, info = env.step(action)   # action (n, n_u)

``info`` is a dict of (n,) tensors (``goal_met`` / ``reached``, ``num_safety_violation``, ``safety_cost``).  Finished
environments are NOT reset behind the caller's back: ``reset(mask)`` resets those selected (``done`` of the last step
by default), so the terminal observation is what ``step`` returned — the reference driver's semantics.

Checked against traces recorded from the reference's own env classes (tests/test_device_envs_gpu.py).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..arena import stream_ptr
from ..envspec import PvtolSpec, SimulatedCarsSpec, UnicycleSpec


def _dparams(*v):
    return (C.c_double * len(v))(*[float(x) for x in v])


class _DeviceEnv:
    state_dim = obs_dim = lya_dim = act_dim = 0

    def _common(self, n_envs, device):
        _lib.load()
        self.n, self.device = int(n_envs), torch.device(device)
        z = lambda *s, dtype=torch.float64: torch.zeros(*s, dtype=dtype, device=self.device)
        self.state = z(self.n, self.state_dim)
        self.ep_step = z(self.n, dtype=torch.int32)
        self.obs, self.reward, self.constraint, self.signal = z(self.n, self.obs_dim), z(self.n), z(self.n), z(self.n)
        self.lya_pre, self.lya_next = z(self.n, self.lya_dim), z(self.n, self.lya_dim)
        self.done = z(self.n, dtype=torch.int32)
        self.info = z(self.n, 3)
        self._all = torch.ones(self.n, dtype=torch.bool, device=self.device)

    def _action(self, action):
        a = torch.as_tensor(action, device=self.device).to(torch.float64).reshape(self.n, self.act_dim).contiguous()
        return a

    def _info(self, first):
        return {first: self.info[:, 0], "num_safety_violation": self.info[:, 1], "safety_cost": self.info[:, 2]}


class DeviceUnicycleEnv(UnicycleSpec, _DeviceEnv):
    """U/envs/unicycle_env.py:57-152 for ``n_envs`` environments (``barrier=True``: NU's barrier signal as well)."""
    state_dim, obs_dim, lya_dim, act_dim = 3, 7, 2, 2
    reward_goal, goal_size, l_p = 500.0, 0.3, 0.03
    little_b, capital_b = 0.0, -20.0

    def __init__(self, n_envs, seed=0, barrier=False, device="cuda"):
        UnicycleSpec.__init__(self, seed)
        self.barrier = barrier
        self._common(n_envs, device)
        self.last_dist = torch.zeros(self.n, dtype=torch.float64, device=self.device)
        self._hz = torch.tensor(np.asarray(self.hazards_locations), dtype=torch.float64, device=self.device).contiguous()
        self.reset()

    def reset(self, mask=None):
        m = self._all if mask is None else torch.as_tensor(mask, device=self.device).bool()
        self.state[m] = torch.tensor([-2.5, -2.5, 0.0], dtype=torch.float64, device=self.device)
        self.ep_step[m] = 0
        self.last_dist[m] = float(np.linalg.norm(self.goal_pos - np.array([-2.47, -2.5])))
        self.step_obs()
        return self.obs

    def step_obs(self):
        """observation of the current states (what ``reset`` returns)"""
        st = self.state
        rx, ry = self.goal_pos[0] - st[:, 0], self.goal_pos[1] - st[:, 1]
        c, s = torch.cos(st[:, 2]), torch.sin(st[:, 2])
        v0, v1 = rx * c + ry * s, -rx * s + ry * c
        n = torch.sqrt(v0 * v0 + v1 * v1) + 0.001
        self.obs.copy_(torch.stack([st[:, 0], st[:, 1], c, s, v0 / n, v1 / n, torch.exp(-torch.sqrt(rx * rx + ry * ry))], 1))

    def step(self, action):
        a = self._action(action)
        par = _dparams(self.dt, self.goal_pos[0], self.goal_pos[1], self.goal_size, self.reward_goal, self.hazards_radius,
                       self.l_p, self.little_b, self.capital_b)
        _lib.call("nlbac_unicycle_env_step", self.n, par, int(self.max_episode_steps), self._hz.data_ptr(),
                  self._hz.shape[0], a.data_ptr(), self.state.data_ptr(), self.ep_step.data_ptr(),
                  self.last_dist.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(), self.constraint.data_ptr(),
                  self.signal.data_ptr(), self.lya_pre.data_ptr(), self.lya_next.data_ptr(), self.done.data_ptr(),
                  self.info.data_ptr(), stream_ptr())
        out = (self.obs, self.reward, self.constraint) + ((self.signal,) if self.barrier else ())
        return out + (self.lya_pre, self.lya_next, self.done, self._info("goal_met"))


class DevicePvtolEnv(PvtolSpec, _DeviceEnv):
    """P/envs/pvtol_env.py:85-216 for ``n_envs`` environments (``barrier=True``: NP's barrier signal as well); the
    Lyapunov inputs are the observations before / after the step."""
    state_dim, obs_dim, lya_dim, act_dim = 7, 11, 11, 2
    reward_goal, goal_size = 1500.0, 3.5
    little_b, capital_b = 0.0, -0.1

    def __init__(self, n_envs, seed=0, barrier=False, device="cuda", **overrides):
        PvtolSpec.__init__(self, seed, **overrides)
        self.barrier = barrier
        self._common(n_envs, device)
        self._hz = torch.tensor(np.asarray(self.hazard_locations), dtype=torch.float64, device=self.device).contiguous()
        self.reset()

    def reset(self, mask=None):
        m = self._all if mask is None else torch.as_tensor(mask, device=self.device).bool()
        self.state[m] = torch.tensor([-4.5, -4.5, 0.0, 0.0, 0.0, 1.0, -4.5], dtype=torch.float64, device=self.device)
        self.ep_step[m] = 0
        st = self.state
        rx, ry = self.goal_pos[0] - st[:, 0], self.goal_pos[1] - st[:, 1]
        c, s = torch.cos(st[:, 2]), torch.sin(st[:, 2])
        v0, v1 = rx * c + ry * s, -rx * s + ry * c
        n = torch.sqrt(v0 * v0 + v1 * v1) + 0.001
        self.obs.copy_(torch.stack([st[:, 0], st[:, 1], c, s, st[:, 3], st[:, 4], st[:, 5], st[:, 6], v0 / n, v1 / n,
                                    torch.exp(-torch.sqrt(rx * rx + ry * ry))], 1))
        return self.obs

    def step(self, action):
        a = self._action(action)
        par = _dparams(self.dt, self.goal_pos[0], self.goal_pos[1], self.goal_size, self.reward_goal, self.hazards_radius,
                       self.safety_operator_follow, self.little_b, self.capital_b)
        _lib.call("nlbac_pvtol_env_step", self.n, par, int(self.max_episode_steps), self._hz.data_ptr(),
                  self._hz.shape[0], a.data_ptr(), self.state.data_ptr(), self.ep_step.data_ptr(), self.obs.data_ptr(),
                  self.reward.data_ptr(), self.constraint.data_ptr(), self.signal.data_ptr(), self.lya_pre.data_ptr(),
                  self.done.data_ptr(), self.info.data_ptr(), stream_ptr())
        out = (self.obs, self.reward, self.constraint) + ((self.signal,) if self.barrier else ())
        return out + (self.lya_pre, self.obs, self.done, self._info("goal_met"))


class DeviceSimulatedCarsEnv(SimulatedCarsSpec, _DeviceEnv):
    """C/envs/simulated_cars_env.py:66-146 for ``n_envs`` environments; the initial velocity noise of ``reset`` is drawn
    on the host from numpy's global generator, one draw per environment reset (as the reference does per reset)."""
    state_dim, obs_dim, lya_dim, act_dim = 10, 10, 4, 1
    should_keep_thre, reward_goal = 0.5, 2.0

    def __init__(self, n_envs, seed=0, device="cuda"):
        SimulatedCarsSpec.__init__(self, seed)
        np.random.seed(seed)
        self._common(n_envs, device)
        self.t = torch.zeros(self.n, dtype=torch.float64, device=self.device)
        self.reset()

    def reset(self, mask=None, noise=None):
        """``noise``: the N(0, 0.5) initial-velocity offsets to use, one per environment (default: one draw from numpy's
        global generator per environment that is reset, in index order)."""
        m = (np.ones(self.n, dtype=bool) if mask is None else torch.as_tensor(mask).bool().cpu().numpy())
        for i in np.nonzero(m)[0]:
            st = np.zeros(10)
            st[::2] = [42.0, 34.0, 26.0, 18.0, 10.0]
            st[1::2] = 3.0 + (np.random.normal(0, 0.5) if noise is None else float(noise[i]))
            st[7] = 3.0
            self.state[i] = torch.from_numpy(st).to(self.device)
        mt = torch.from_numpy(m).to(self.device)
        self.t[mt] = 0.0
        self.ep_step[mt] = 0
        o = self.state.clone()
        o[:, ::2] /= 100.0
        o[:, 1::2] /= 30.0
        self.obs.copy_(o)
        return self.obs

    def step(self, action):
        a = self._action(action)
        par = _dparams(self.dt, self.kp, self.k_brake, self.should_keep, self.should_keep_thre, self.reward_goal)
        _lib.call("nlbac_cars_env_step", self.n, par, int(self.max_episode_steps), a.data_ptr(), self.state.data_ptr(),
                  self.t.data_ptr(), self.ep_step.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                  self.constraint.data_ptr(), self.lya_pre.data_ptr(), self.lya_next.data_ptr(), self.done.data_ptr(),
                  self.info.data_ptr(), stream_ptr())
        return (self.obs, self.reward, self.constraint, self.lya_pre, self.lya_next, self.done, self._info("reached"))


def make(name, n_envs, seed=0, device="cuda", **kw):
    """``Unicycle`` / ``UnicycleBarrier`` / ``SimulatedCars`` / ``Pvtol`` / ``PvtolBarrier``, batched on the device."""
    if name in ("Unicycle", "UnicycleBarrier"):
        return DeviceUnicycleEnv(n_envs, seed, barrier=name.endswith("Barrier"), device=device)
    if name == "SimulatedCars":
        return DeviceSimulatedCarsEnv(n_envs, seed, device=device)
    if name in ("Pvtol", "PvtolBarrier"):
        return DevicePvtolEnv(n_envs, seed, barrier=name.endswith("Barrier"), device=device, **kw)
    raise Exception("Dynamics mode not supported.")
