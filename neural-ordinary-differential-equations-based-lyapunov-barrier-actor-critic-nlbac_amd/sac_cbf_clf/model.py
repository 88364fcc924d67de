"""Network containers of the agent.

Same classes, constructor signatures, ``state_dict`` key names and weight
initialisation as the reference (``U/sac_cbf_clf/model.py:14-17, 37-133,
177-206``): actor / critic / Lyapunov layers are Xavier-uniform with zero
bias, the NODE keeps PyTorch's default ``nn.Linear`` init.  The modules only
*hold* parameters (as views into a flat HBM arena, see ``nlbac_amd.arena``);
all arithmetic runs in the HIP kernels — ``forward`` here is the thin
inference path used by ``select_action``.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..arena import MlpHandle, io_array, mlp_array, pack, stream_ptr

LOG_SIG_MAX = 2
LOG_SIG_MIN = -20
epsilon = 1e-6


def weights_init_(m):
    if isinstance(m, nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight, gain=1)
        torch.nn.init.constant_(m.bias, 0)


def _run_fwd(handles, ios, B, target=False):
    descs = [h.desc_target if target else h.desc for h in handles]
    _lib.call("nlbac_mlp_fwd", mlp_array(descs), ios, len(descs), B, stream_ptr())


class QNetwork(nn.Module):
    """Twin Q: linear1-3 and linear4-6 over cat(state, action)."""

    def __init__(self, num_inputs, num_actions, hidden_dim):
        super().__init__()
        i = num_inputs + num_actions
        self.linear1 = nn.Linear(i, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear3 = nn.Linear(hidden_dim, 1)
        self.linear4 = nn.Linear(i, hidden_dim)
        self.linear5 = nn.Linear(hidden_dim, hidden_dim)
        self.linear6 = nn.Linear(hidden_dim, 1)
        self.apply(weights_init_)

    def attach(self, arena):
        self.q1 = MlpHandle(arena, [(l.weight, l.bias) for l in (self.linear1, self.linear2, self.linear3)], "q1")
        self.q2 = MlpHandle(arena, [(l.weight, l.bias) for l in (self.linear4, self.linear5, self.linear6)], "q2")
        return [self.q1, self.q2]


class LyaNetwork(nn.Module):
    """Lyapunov critic (also the shape of the NBC variants' BarrierNetwork)."""

    def __init__(self, num_inputs, hidden_dim):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.linear3 = nn.Linear(hidden_dim, 1)
        self.apply(weights_init_)

    def attach(self, arena):
        self.net = MlpHandle(arena, [(l.weight, l.bias) for l in (self.linear1, self.linear2, self.linear3)], "lya")
        return [self.net]


class BarrierNetwork(LyaNetwork):
    """Learned barrier certificate B(obs, action) (NU/sac_cbf_clf/model.py:67-84): the Lyapunov critic's shape
    over cat(state, action)."""

    def __init__(self, num_inputs, num_actions, hidden_dim):
        super().__init__(num_inputs + num_actions, hidden_dim)

    def attach(self, arena):
        self.net = MlpHandle(arena, [(l.weight, l.bias) for l in (self.linear1, self.linear2, self.linear3)], "barrier")
        return [self.net]


class GaussianPolicy(nn.Module):
    """Squashed-Gaussian actor; mean / log_std heads share one skinny layer on device."""

    def __init__(self, num_inputs, num_actions, hidden_dim, action_space=None):
        super().__init__()
        self.linear1 = nn.Linear(num_inputs, hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, hidden_dim)
        self.mean_linear = nn.Linear(hidden_dim, num_actions)
        self.log_std_linear = nn.Linear(hidden_dim, num_actions)
        self.apply(weights_init_)
        self.num_actions = num_actions
        if action_space is None:
            self.action_scale = torch.ones(num_actions)
            self.action_bias = torch.zeros(num_actions)
        else:
            self.action_scale = torch.FloatTensor((action_space.high - action_space.low) / 2.)
            self.action_bias = torch.FloatTensor((action_space.high + action_space.low) / 2.)

    def attach(self, arena):
        self.net = MlpHandle(arena, [
            (self.linear1.weight, self.linear1.bias), (self.linear2.weight, self.linear2.bias),
            ([self.mean_linear.weight, self.log_std_linear.weight],
             [self.mean_linear.bias, self.log_std_linear.bias])], "policy")
        return [self.net]

    def to(self, device):
        self.action_scale = self.action_scale.to(device)
        self.action_bias = self.action_bias.to(device)
        return super().to(device)

    ACT_DRAWS = 4096          # N(0,1) rows drawn per refill of the latency path's noise block

    def act(self, state, evaluate=False):
        """One action for one observation — the driver's per-env-step call (``select_action``, U/main.py:106).  Latency
        path: ONE launch and no copies — the policy forward reads the observation from pinned host memory, draws the
        squashed-Gaussian sample itself (``nlbac_mlp_fwd_gauss``) and writes the action to pinned host memory; the
        N(0,1) draws come from a device block refilled every ``ACT_DRAWS`` calls.  ``evaluate``: the deterministic
        action tanh(mean)·scale + bias, i.e. the same head with a zero draw."""
        A = self.num_actions
        w = self.__dict__.get("_act_ws")
        if w is None:
            import types
            dev = self.net.arena.device
            obs = self.linear1.in_features
            w = types.SimpleNamespace(
                pin_in=torch.zeros(1, obs).pin_memory(), pin_out=torch.zeros(1, A).pin_memory(),
                heads=torch.zeros(1, 2 * A, device=dev), logp=torch.zeros(1, device=dev),
                block=torch.zeros(self.ACT_DRAWS, A, device=dev), zero=torch.zeros(1, A, device=dev), k=self.ACT_DRAWS,
                io=io_array(1), head=_lib.GaussHead(), ev=torch.cuda.Event())
            w.io[0].x0, w.io[0].x0_dim, w.io[0].x0_ld = w.pin_in.data_ptr(), obs, obs      # (the device reads host memory)
            w.io[0].y, w.io[0].y_ld = w.heads.data_ptr(), 2 * A
            w.head.scale, w.head.bias, w.head.n_u = self.action_scale.data_ptr(), self.action_bias.data_ptr(), A
            w.head.action, w.head.action_ld, w.head.logp = w.pin_out.data_ptr(), A, w.logp.data_ptr()
            w.np_in, w.np_out = w.pin_in.numpy(), w.pin_out.numpy()
            w.nets = mlp_array([self.net.desc])
            w.eps = w.zero
            self.__dict__["_act_ws"] = w
        if evaluate:
            w.eps = w.zero
        else:
            if w.k >= self.ACT_DRAWS:
                w.block.normal_()
                w.k = 0
            w.eps = w.block[w.k:w.k + 1]          # (the draw this call uses)
            w.k += 1
        w.head.eps = w.eps.data_ptr()
        w.np_in[0, :] = state                     # (float64 -> float32, as the reference's FloatTensor cast)
        _lib.call("nlbac_mlp_fwd_gauss", w.nets, w.io, 1, 1, C.byref(w.head), stream_ptr())
        w.ev.record()
        w.ev.synchronize()
        return w.np_out[0].copy()

    def sample(self, state, eps=None):
        """(action, log_prob, tanh(mean)*scale+bias) for a (n, obs) device tensor."""
        n, A = state.shape[0], self.num_actions
        state = state.contiguous().float()
        heads = torch.empty(n, 2 * A, device=state.device)
        io = io_array(1)
        io[0].x0, io[0].x0_dim, io[0].x0_ld = state.data_ptr(), state.shape[1], state.shape[1]
        io[0].y, io[0].y_ld = heads.data_ptr(), 2 * A
        _run_fwd([self.net], io, n)
        if eps is None:
            eps = torch.randn(n, A, device=state.device)
        action = torch.empty(n, A, device=state.device)
        logp = torch.empty(n, device=state.device)
        _lib.call("nlbac_gauss_sample_fwd", heads.data_ptr(), 2 * A, eps.data_ptr(), self.action_scale.data_ptr(),
                  self.action_bias.data_ptr(), A, n, action.data_ptr(), A, logp.data_ptr(), stream_ptr())
        mean = torch.tanh(heads[:, :A]) * self.action_scale + self.action_bias
        return action, logp.unsqueeze(1), mean


class NeuralODEModel(nn.Module):
    """Learned dynamics, both reference forms:

    ``NeuralODEModel(input_dim, output_dim1, output_dim2)`` — control-affine field f(x) + g(x) u with f_net 5
    and g_net 4 Linear layers of width 100 (U/sac_cbf_clf/model.py:177-206; Pvtol uses the same form);
    ``NeuralODEModel(input_dim, output_dim)`` — one net of 4 Linear layers of width 64 on [x, u, t]
    (C/sac_cbf_clf/model.py:179-194); the trailing ``input_dim - output_dim`` inputs are carried unchanged."""

    def __init__(self, input_dim, output_dim1, output_dim2=None, hidden_dim=None, f_depth=5, g_depth=4, depth=4,
                 normalizer=None):
        """``normalizer`` (single-net form only): ``(in_mean, in_std, out_mean, out_std)`` — states and actions are
        normalised before they enter the net and its outputs de-normalised before they are used as the prediction
        (the Quadrotor NODE, /root/reference/README.md:192; no reference code exists for it):
        ``dx/dt = out_mean + out_std * net(([x, u] - in_mean) / in_std)``.  Fixed constants, kept as buffers."""
        super().__init__()
        self.input_dim, self.output_dim1, self.output_dim2 = input_dim, output_dim1, output_dim2
        self.affine = output_dim2 is not None
        self.normalized = normalizer is not None
        if self.normalized:
            assert not self.affine, "normalisation is built for the single-net form"
            im, isd, om, osd = (torch.as_tensor(np.asarray(v), dtype=torch.float32).reshape(-1) for v in normalizer)
            assert im.numel() == isd.numel() == input_dim and om.numel() == osd.numel() == output_dim1
            for name, v in (("in_mean", im), ("in_std", isd), ("out_mean", om), ("out_std", osd)):
                self.register_buffer(name, v.clone(), persistent=False)   # env constants, not checkpoint state

        def seq(in_dim, depth, out, hid):
            layers = [nn.Linear(in_dim, hid), nn.ReLU()]
            for _ in range(depth - 2):
                layers += [nn.Linear(hid, hid), nn.ReLU()]
            layers += [nn.Linear(hid, out)]
            return nn.Sequential(*layers)
        if self.affine:
            hid = hidden_dim or 100
            self.n_s, self.n_u = output_dim1, output_dim2 // output_dim1
            self.f_net = seq(input_dim, f_depth, output_dim1, hid)
            self.g_net = seq(input_dim, g_depth, output_dim2, hid)
        else:
            hid = hidden_dim or 64
            self.output_dim = output_dim1
            self.n_s, self.n_carry = output_dim1, input_dim - output_dim1
            self.net = seq(input_dim, depth, output_dim1, hid)

    def device_handles(self):
        """The model's ``MlpHandle``s; a model built outside an agent attaches itself to an arena of its own here."""
        hs = [self.f, self.g] if (self.affine and hasattr(self, "f")) else \
            ([self.net_handle] if hasattr(self, "net_handle") else None)
        if hs is None:
            from ..arena import Arena
            arena = Arena("cuda", n_slabs=16)
            hs = self.attach(arena)
            arena.finalize()
        if hs[0].desc is None:
            for h in hs:
                h.bind()
            pack(hs)
        return hs

    def norm_device(self):
        """[in_mean | 1/in_std | out_mean | out_std] on the model's device (what the fused kernels read), or None."""
        if not self.normalized:
            return None
        t = self.__dict__.get("_norm_dev")
        if t is None:
            dev = self.device_handles()[0].arena.device
            isig = (1.0 / self.in_std.float())          # fp32 reciprocal, as the oracle forms it
            t = torch.cat([self.in_mean.float(), isig, self.out_mean.float(), self.out_std.float()]).to(dev).contiguous()
            self.__dict__["_norm_dev"] = t
        return t

    def refresh_device_weights(self):
        """Re-pack the MFMA-fragment copies of the weights (after anything but this build's own optimiser kernel
        has written the parameters, e.g. a ``torch.optim`` step or ``load_state_dict``)."""
        pack(self.device_handles())

    def forward(self, t, s):
        """The reference's field evaluation (U/model.py:208-217, C/model.py:196-203): ``s = [x | carried inputs]``
        -> ``[dx/dt | 0]``, on the device kernels.  Not recorded by autograd — gradients flow through
        ``nlbac_amd.odeint.odeint``, which differentiates whole solves."""
        from ..odeint import _solver_of
        sv = _solver_of(self)
        ns = sv.n_s
        s = s.detach().float().contiguous()
        n = s.shape[0]
        s_in = s
        if self.normalized:                   # (inference path: normalise / de-normalise around the net launch)
            nd = self.norm_device()
            d = self.input_dim
            s_in = (s - nd[:d]) * nd[d:2 * d]
        x, c = s_in[:, :ns].contiguous(), s_in[:, ns:].contiguous()
        k, g = torch.empty(n, ns, device=s.device), torch.empty(n, ns * max(1, sv.n_u), device=s.device)
        sv._eval(x, c, n, k, g)               # (fresh launch descriptors: the inputs are the caller's tensors)
        if self.normalized:
            k = k * nd[2 * d + ns:2 * d + 2 * ns] + nd[2 * d:2 * d + ns]
        return torch.cat([k, torch.zeros_like(s[:, ns:])], dim=1)

    def attach(self, arena):
        if self.affine:
            self.f = MlpHandle(arena, [(m.weight, m.bias) for m in self.f_net if isinstance(m, nn.Linear)], "f_net")
            self.g = MlpHandle(arena, [(m.weight, m.bias) for m in self.g_net if isinstance(m, nn.Linear)], "g_net")
            return [self.f, self.g]
        self.net_handle = MlpHandle(arena, [(m.weight, m.bias) for m in self.net if isinstance(m, nn.Linear)], "net")
        return [self.net_handle]


def train_step(model, state, action, next_state, *rest) -> float:
    """The reference's NODE regression step, same positional signature (U/model.py:221-260; the SimulatedCars copy
    passes ``time_batch`` before the optimizer, C/model.py:208-252):

        train_step(model, state, action, next_state[, time_batch], optimizer, loss_func, horizon, solver, time_interval)

    one solve over ``[0, time_interval]`` from ``[state | action (| time)]``, ``loss_func`` on the predicted state,
    backward through the solve, ``optimizer.step()``; returns ``loss / horizon``.  (``SAC_CBF_CLF.update_parameters``
    does not come through here: its NODE fit is the fused device path, ``fit_node_rows``.)"""
    from ..odeint import odeint
    if len(rest) == 6:
        time_batch, optimizer, loss_func, horizon, solver, time_interval = rest
    else:
        (optimizer, loss_func, horizon, solver, time_interval), time_batch = rest, None
    assert state.dim() == 2 and action.dim() == 2 and next_state.dim() == 2, (state.shape, action.shape, next_state.shape)
    dev = model.device_handles()[0].arena.device
    cols = [state, action] + ([time_batch] if time_batch is not None else [])
    y0 = torch.cat([torch.as_tensor(c, dtype=torch.float32).to(dev) for c in cols], dim=-1)
    model.train()
    optimizer.zero_grad()
    pred = odeint(model, y0, torch.tensor([0.0, float(time_interval)]), method=solver, atol=1e-7, rtol=1e-5)[-1]
    loss = loss_func(pred[:, :state.shape[1]], torch.as_tensor(next_state, dtype=torch.float32).to(dev))
    loss.backward()
    optimizer.step()
    model.refresh_device_weights()
    return float(loss.item()) / horizon
