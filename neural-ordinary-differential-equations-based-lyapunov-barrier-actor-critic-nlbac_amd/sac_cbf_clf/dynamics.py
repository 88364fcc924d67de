"""obs <-> state glue with the reference's calling convention
(``U/sac_cbf_clf/dynamics.py:12-77``): numpy in -> numpy out (float64),
tensor in -> tensor out (same dtype/device, detached).  The update path does
not call this (obs->state runs in ``nlbac_unicycle_state`` on the device);
the training driver does, once per env step.
"""
import numpy as np
import torch


class DynamicsModel:

    def __init__(self, env, args):
        self.env = env
        self.device = torch.device("cuda" if getattr(args, "cuda", False) else "cpu")

    def get_state(self, obs):
        is_tensor = torch.is_tensor(obs)
        if is_tensor:
            dtype, device = obs.dtype, obs.device
            obs = obs.detach().cpu().double().numpy()
        single = obs.ndim == 1
        o = obs[None] if single else obs
        if self.env.dynamics_mode == 'Unicycle':
            state = np.stack([o[:, 0], o[:, 1], np.arctan2(o[:, 3], o[:, 2])], axis=1).astype(np.float64)
        elif self.env.dynamics_mode == 'SimulatedCars':
            state = o.astype(np.float64).copy()
            state[:, ::2] *= 100.0
            state[:, 1::2] *= 30.0
        elif self.env.dynamics_mode == 'Pvtol':
            # P/sac_cbf_clf/dynamics.py:50-73: (state with the safety operator, the six dynamic states)
            state = np.stack([o[:, 0], o[:, 1], np.arctan2(o[:, 3], o[:, 2]), o[:, 4], o[:, 5], o[:, 6], o[:, 7]],
                             axis=1).astype(np.float64)
            dyn = state[:, :6]
            if single:
                state, dyn = state[0], dyn[0]
            if is_tensor:
                return (torch.from_numpy(state).type(dtype).to(device),
                        torch.from_numpy(np.ascontiguousarray(dyn)).type(dtype).to(device))
            return state, dyn
        else:
            raise Exception('Unknown dynamics')
        if single:
            state = state[0]
        return torch.from_numpy(state).type(dtype).to(device) if is_tensor else state

    def seed(self, s):
        torch.manual_seed(s)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(s)
