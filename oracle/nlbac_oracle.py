"""ORACLE — test infrastructure, NOT product code.

CPU (PyTorch fp32 + autograd) restatement of the NLBAC hot path for the
Unicycle and SimulatedCars agents.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product
package never does.

Pinned by ``tests/golden/{unicycle,cars}_*.npz`` which ``oracle/gen_golden.py``
produced by importing the reference's own modules in the build container
(Euler: the only solver configuration the reference executes, exact).
rk4 / dopri5 follow torchdiffeq 0.2.3 *from recollection* (the package is
not in the container) — PARITY UNPINNED for those two solvers.

Pvtol (P = NLBAC_pvtol_RL_training/Pvtol_RL_training, fixtures ``pvtol_*.npz``) is restated by
``OraclePvtolAgent``.  The learned-barrier-certificate Unicycle copy (NU, fixtures ``nbc_unicycle_*.npz``) is restated by
``OracleUnicycleBarrierAgent``.

SimulatedCars (C = NLBAC_SimulatedCarsFollowing_RL_training/Simulated_Car_Following_RL_training) is
restated by ``OracleCarsAgent``: non-affine NODE C/sac_cbf_clf/model.py:179-205, two-step rollout and
relative-degree-2 CBFs C/sac_cbf_clf/sac_cbf_clf.py:412-555 / 557-681, get_state/get_obs
C/sac_cbf_clf/dynamics.py:25-96, train_step C/sac_cbf_clf/model.py:208-252.

Reference lines restated (U = NLBAC_Unicycle_RL_training/Unicycle_RL_training):
  update_parameters        U/sac_cbf_clf/sac_cbf_clf.py:181-319
  get_policy_loss_2        U/sac_cbf_clf/sac_cbf_clf.py:408-530
  backup_get_policy_loss_2 U/sac_cbf_clf/sac_cbf_clf.py:532-640
  QNetwork/LyaNetwork/GaussianPolicy  U/sac_cbf_clf/model.py:37-133
  NeuralODEModel.forward   U/sac_cbf_clf/model.py:208-217
  train_step               U/sac_cbf_clf/model.py:221-260
  get_state                U/sac_cbf_clf/dynamics.py:27-69
  soft_update              U/sac_cbf_clf/utils.py:75-79
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOG_SIG_MAX, LOG_SIG_MIN, EPS = 2.0, -20.0, 1e-6
L_P = 0.03


# ---------------------------------------------------------------------------
# networks (functional; parameters live in dicts keyed like the reference)
# ---------------------------------------------------------------------------

def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def qnet(sd, obs, act):
    """model.py:53-64"""
    xu = torch.cat([obs, act], 1)
    x1 = _lin(sd, "linear3", F.relu(_lin(sd, "linear2", F.relu(_lin(sd, "linear1", xu)))))
    x2 = _lin(sd, "linear6", F.relu(_lin(sd, "linear5", F.relu(_lin(sd, "linear4", xu)))))
    return x1, x2


def lyanet(sd, x):
    """model.py:77-83"""
    return _lin(sd, "linear3", F.relu(_lin(sd, "linear2", F.relu(_lin(sd, "linear1", x)))))


def policy_sample(sd, obs, eps, scale, bias):
    """model.py:108-128 with the N(0,1) draw supplied by the caller."""
    x = F.relu(_lin(sd, "linear2", F.relu(_lin(sd, "linear1", obs))))
    mean = _lin(sd, "mean_linear", x)
    log_std = torch.clamp(_lin(sd, "log_std_linear", x), min=LOG_SIG_MIN, max=LOG_SIG_MAX)
    std = log_std.exp()
    x_t = mean + eps * std
    y_t = torch.tanh(x_t)
    action = y_t * scale + bias
    var = std ** 2
    log_prob = -((x_t - mean) ** 2) / (2 * var) - log_std - math.log(math.sqrt(2 * math.pi))
    log_prob = log_prob - torch.log(scale * (1 - y_t.pow(2)) + EPS)
    log_prob = log_prob.sum(1, keepdim=True)
    return action, log_prob, torch.tanh(mean) * scale + bias


class AffineNode:
    """model.py:177-217: ds/dt = f(x) + g(x) u ; action columns carried."""

    def __init__(self, sd, n_s=3, n_u=2, f_layers=5, g_layers=4):
        self.sd, self.n_s, self.n_u = sd, n_s, n_u
        self.f_names = ["f_net.%d" % (2 * i) for i in range(f_layers)]
        self.g_names = ["g_net.%d" % (2 * i) for i in range(g_layers)]
        self.nfe = 0

    def _mlp(self, names, x):
        for n in names[:-1]:
            x = F.relu(_lin(self.sd, n, x))
        return _lin(self.sd, names[-1], x)

    def __call__(self, t, s):
        self.nfe += 1
        x, u = s[..., :self.n_s], s[..., self.n_s:self.n_s + self.n_u]
        f = self._mlp(self.f_names, x)
        g = self._mlp(self.g_names, x).reshape(-1, self.n_s, self.n_u)
        ds = f + torch.bmm(g, u.reshape(-1, self.n_u, 1)).squeeze(-1)
        return torch.cat((ds, torch.zeros_like(u)), -1)


class ConcatNode:
    """C/sac_cbf_clf/model.py:179-205: ds/dt = net([x, u, t]); the action and time columns are carried."""

    def __init__(self, sd, n_s=10, n_carry=2, depth=4, norm=None):
        """``norm`` = (in_mean, in_std, out_mean, out_std): the Quadrotor form (/root/reference/README.md:192, prose
        only — no reference code): states and actions are normalised before they enter the net, its outputs are
        de-normalised before they are used as the prediction."""
        self.sd, self.n_s, self.n_carry = sd, n_s, n_carry
        self.names = ["net.%d" % (2 * i) for i in range(depth)]
        self.nfe = 0
        self.norm = None
        if norm is not None:
            im, isd, om, osd = (torch.as_tensor(np.asarray(v), dtype=torch.float32) for v in norm)
            self.norm = (im, 1.0 / isd, om, osd)

    def __call__(self, t, s):
        self.nfe += 1
        x = s
        if self.norm is not None:
            x = (s - self.norm[0]) * self.norm[1]
        for n in self.names[:-1]:
            x = F.relu(_lin(self.sd, n, x))
        ds = _lin(self.sd, self.names[-1], x)
        if self.norm is not None:
            ds = ds * self.norm[3] + self.norm[2]
        return torch.cat((ds, torch.zeros_like(s[:, self.n_s:])), -1)


# ---------------------------------------------------------------------------
# odeint restatement (torchdiffeq 0.2.3 semantics; see module docstring)
# ---------------------------------------------------------------------------

_DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_C_SOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_DP_C_ERR = [
    35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
    -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.,
]
_DP_C_MID = [
    6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2,
    -2691868925 / 45128329728 / 2, 187940372067 / 1594534317056 / 2,
    -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2,
]


def _rms(x):
    return float(x.detach().double().pow(2).mean().sqrt())


def dopri5_initial_step(func, t0, y0, f0, rtol, atol, order=4):
    """Hairer's rule as in torchdiffeq ``_select_initial_step``; returns a
    python float (step sizes carry no gradient in this restatement)."""
    with torch.no_grad():
        scale = atol + y0.abs() * rtol
        d0, d1 = _rms(y0 / scale), _rms(f0 / scale)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        y1 = y0 + np.float32(h0) * f0
        f1 = func(t0 + h0, y1)
        d2 = _rms((f1 - f0) / scale) / h0
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = max(1e-6, h0 * 1e-3)
        else:
            h1 = (0.01 / max(d1, d2)) ** (1.0 / float(order + 1))
        return min(100 * h0, h1)


def dopri5_step(func, t0, h, y0, f0):
    """One Dormand–Prince 5(4) step (FSAL): returns y1, f1, err, k list."""
    hf = np.float32(h)
    k = [f0]
    yi = y0
    for a, beta in zip(_DP_ALPHA, _DP_BETA):
        yi = y0
        for bj, kj in zip(beta, k):
            if bj != 0:
                yi = yi + kj * (np.float32(bj) * hf)
        k.append(func(t0 + a * h, yi))
    y1, f1 = yi, k[-1]
    err = sum(kj * (np.float32(c) * hf) for c, kj in zip(_DP_C_ERR, k))
    return y1, f1, err, k


def dopri5_interp(y0, y1, k, h, x):
    """4th-order interpolant through (y0, y_mid, y1) evaluated at
    x = (t - t0)/h (torchdiffeq ``_interp_fit`` / ``_interp_evaluate``)."""
    hf = np.float32(h)
    y_mid = y0 + sum(kj * (np.float32(c) * hf) for c, kj in zip(_DP_C_MID, k) if c != 0)
    f0, f1 = k[0], k[-1]
    a = 2 * hf * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
    b = hf * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
    c = hf * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
    d = hf * f0
    xf = np.float32(x)
    return y0 + xf * (d + xf * (c + xf * (b + xf * a)))


def odeint(func, y0, t, method="euler", atol=1e-7, rtol=1e-5, info=None):
    """IVP solve on the grid ``t`` (only ``t=[t0,t1]`` is used on this path).

    euler / rk4: one fixed step over [t0,t1] (fixed-grid solvers step on the
    output grid when no ``step_size`` is given).  dopri5: adaptive, one shared
    step size for the whole batch tensor, steps are not clipped to t1 — the
    result is the interpolant at t1.
    """
    assert len(t) == 2
    t0, t1 = float(t[0]), float(t[1])
    dt = (t[1] - t[0]).to(y0.dtype) if torch.is_tensor(t) else np.float32(t1 - t0)
    if method == "euler":
        y1 = y0 + dt * func(t0, y0)
    elif method == "rk4":  # 3/8 rule
        k1 = func(t0, y0)
        k2 = func(t0 + (t1 - t0) / 3, y0 + dt * k1 * (1.0 / 3.0))
        k3 = func(t0 + (t1 - t0) * 2 / 3, y0 + dt * (k2 - k1 * (1.0 / 3.0)))
        k4 = func(t1, y0 + dt * (k1 - k2 + k3))
        y1 = y0 + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
    elif method == "dopri5":
        f0 = func(t0, y0)
        h = dopri5_initial_step(func, t0, y0, f0, rtol, atol)
        tc, yc, fc = t0, y0, f0
        steps = []
        n = 0
        while True:
            assert n < 1000, "max_num_steps exceeded"
            n += 1
            yn, fn, err, k = dopri5_step(func, tc, h, yc, fc)
            with torch.no_grad():
                tol = atol + rtol * torch.max(yc.abs(), yn.abs())
                ratio = _rms(err / tol)
            accept = ratio <= 1
            steps.append((h, ratio, accept))
            if ratio == 0:
                fac = 10.0
            else:
                dfac = 1.0 if ratio < 1 else 0.2
                fac = min(10.0, max(0.9 / ratio ** 0.2, dfac))
            h_next = h * fac
            if accept:
                if tc + h >= t1:
                    y1 = dopri5_interp(yc, yn, k, h, (t1 - tc) / h)
                    break
                tc, yc, fc = tc + h, yn, fn
            h = h_next
        if info is not None:
            info["steps"] = steps
    else:
        raise ValueError(method)
    return torch.stack([y0, y1])


# ---------------------------------------------------------------------------
# odeint_adjoint restatement — torchdiffeq 0.2.3 ``OdeintAdjointMethod`` (torchdiffeq/_impl/adjoint.py) from the
# published algorithm; the reference never calls it and the package is not in the container, so this is
# SPEC FROM THE PUBLISHED ALGORITHM, PARITY UNPINNED.  What it restates:
#   forward : ``odeint`` under no_grad, only y(t1) is kept
#   backward: the augmented state (y, adj_y, adj_params) is integrated from t1 back to t0 with the same method and
#             tolerances; d/dt (y, adj_y, adj_p) = (f, -adj_y^T df/dy, -adj_y^T df/dp); the decreasing time grid is
#             handled the way ``_check_inputs`` does (t -> -t, f -> -f): in s = -t every right-hand side flips sign
#   norm    : ``handle_adjoint_norm_`` default: max(rms(y part), rms(adj_y part), max_i rms(adj_param_i)) of the
#             scaled quantity — a mixed norm over the tuple, every parameter tensor on its own
#   result  : adaptive: the dopri5 interpolant at t0 for every component (steps are not clipped); fixed grid: one step
# ---------------------------------------------------------------------------

def _tuple_norm(parts):
    return max([_rms(p) for p in parts] + [0.0])


def _tuple_axpy(z0, ks, coefs, hf):
    out = []
    for i, z in enumerate(z0):
        a = z
        for c, k in zip(coefs, ks):
            if c != 0:
                a = a + k[i] * (np.float32(c) * hf)
        out.append(a)
    return out


def odeint_tuple(G, z0, T, method, atol, rtol, info=None):
    """Integrate dz/ds = G(z) over s in [0, T] for a list-of-tensors state with the mixed norm above.  Same solver
    arithmetic as ``odeint`` (which is the one-tensor case with a plain RMS norm)."""
    hT = np.float32(T)
    if method == "euler":
        return _tuple_axpy(z0, [G(z0)], [1.0], hT)
    if method == "rk4":          # 3/8 rule in tableau form (the same numbers as ``odeint``'s rk4 up to rounding order)
        ks = [G(z0)]
        for beta in ([1 / 3], [-1 / 3, 1.0], [1.0, -1.0, 1.0]):
            ks.append(G(_tuple_axpy(z0, ks, beta, hT)))
        return _tuple_axpy(z0, ks, [1 / 8, 3 / 8, 3 / 8, 1 / 8], hT)
    assert method == "dopri5", method
    with torch.no_grad():
        f0 = G(z0)
        scale = [atol + z.abs() * rtol for z in z0]
        d0 = _tuple_norm([z / s for z, s in zip(z0, scale)])
        d1 = _tuple_norm([f / s for f, s in zip(f0, scale)])
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        f1 = G(_tuple_axpy(z0, [f0], [1.0], np.float32(h0)))
        d2 = _tuple_norm([(b - a) / s for a, b, s in zip(f0, f1, scale)]) / h0
        h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
        h = min(100 * h0, h1)
        tc, zc, fc, steps = 0.0, z0, f0, []
        for n in range(1000):
            hf = np.float32(h)
            ks = [fc]
            zi = zc
            for beta in _DP_BETA:
                zi = _tuple_axpy(zc, ks, beta, hf)
                ks.append(G(zi))
            zn, fn = zi, ks[-1]
            err = [sum(k[i] * (np.float32(c) * hf) for c, k in zip(_DP_C_ERR, ks)) for i in range(len(zc))]
            ratio = _tuple_norm([e / (atol + rtol * torch.max(a.abs(), b.abs())) for e, a, b in zip(err, zc, zn)])
            accept = ratio <= 1
            steps.append((h, ratio, accept))
            fac = 10.0 if ratio == 0 else min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1 else 0.2))
            if accept:
                if tc + h >= T:
                    x = (T - tc) / h
                    out = [dopri5_interp(zc[i], zn[i], [k[i] for k in ks], h, x) for i in range(len(zc))]
                    if info is not None:
                        info["steps"] = steps
                    return out
                tc, zc, fc = tc + h, zn, fn
            h = h * fac
    raise AssertionError("max_num_steps exceeded")


class _OdeintAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, func, method, atol, rtol, info, n_params, y0, t, *params):
        with torch.no_grad():
            ans = odeint(func, y0, t, method=method, atol=atol, rtol=rtol, info=info)
        ctx.func, ctx.method, ctx.atol, ctx.rtol, ctx.info = func, method, atol, rtol, info
        ctx.save_for_backward(t, ans, *params)
        return ans

    @staticmethod
    def backward(ctx, grad_y):
        t, y, *params = ctx.saved_tensors
        func = ctx.func
        params = tuple(params)

        def G(z):           # s-time right-hand side: minus the augmented dynamics of adjoint.py
            yv, adj = z[0], z[1]
            with torch.enable_grad():
                yv = yv.detach().requires_grad_(True)
                fe = func(t[0], yv)
                vj = torch.autograd.grad(fe, (yv,) + params, -adj, allow_unused=True)
            vj = [torch.zeros_like(x) if v is None else v for v, x in zip(vj, (yv,) + params)]
            return [-fe.detach()] + [-v for v in vj]

        z1 = [y[-1], grad_y[-1]] + [torch.zeros_like(p) for p in params]
        binfo = {}
        with torch.no_grad():
            z0 = odeint_tuple(G, z1, float(t[1]) - float(t[0]), ctx.method, ctx.atol, ctx.rtol, binfo)
        if ctx.info is not None:
            ctx.info["adjoint_steps"] = binfo.get("steps")
        adj_y = z0[1] + grad_y[0]
        return (None, None, None, None, None, None, adj_y, None, *z0[2:])


def odeint_adjoint(func, y0, t, method="euler", atol=1e-7, rtol=1e-5, adjoint_params=None, info=None):
    """``torchdiffeq.odeint_adjoint`` on ``t = [t0, t1]`` (adjoint tolerances / method default to the forward ones,
    as in torchdiffeq).  ``adjoint_params``: the tensors the parameter adjoint is integrated for — default: every
    tensor of ``func.sd`` (torchdiffeq: ``find_parameters(func)``); ``()`` leaves the parameter adjoint out of the
    augmented state and of the step-size norm."""
    if adjoint_params is None:
        adjoint_params = tuple(func.sd.values())
    adjoint_params = tuple(p for p in adjoint_params if p.requires_grad)
    return _OdeintAdjoint.apply(func, method, atol, rtol, info, len(adjoint_params), y0, t, *adjoint_params)


# ---------------------------------------------------------------------------
# the agent
# ---------------------------------------------------------------------------

def _leafify(sd_np):
    return {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=True)
            for k, v in sd_np.items()}


def _flat(ts):
    return torch.cat([t.reshape(-1) for t in ts])


class Args:
    """Defaults of U/main.py:191-239 that the agent reads."""
    gamma = 0.99
    gamma_b = 50.0
    tau = 0.005
    alpha = 0.2
    lr = 3e-4
    batch_size = 128
    hidden_size = 256
    target_update_interval = 1
    Lagrangian_multiplier_update_interval = 8
    automatic_entropy_tuning = True
    policy = "Gaussian"
    seed = 0
    cuda = False
    backup_update_interval = 20     # Pvtol only (P/main.py:291)

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class OracleAgentBase:
    """Common part of the restated ``SAC_CBF_CLF`` (explicit noise); env copies subclass it."""
    LAM_HI, RATIO_MIN, N_EPS = 400.0, None, 3
    EPS_BACKUP = 2            # index of the backup controller's draw in ``eps``
    OWN_BACKUP_RHO = False    # Pvtol keeps a separate backup_augmented_term

    def __init__(self, env, args, weights, solver="euler"):
        self.env, self.args, self.solver = env, args, solver
        self.gamma, self.gamma_b, self.tau = args.gamma, args.gamma_b, args.tau
        self.alpha = args.alpha
        self.backup_alpha = args.alpha
        self.batch_size = args.batch_size
        self.critic = _leafify(weights["critic"])
        self.critic_target = {k: v.detach().clone() for k, v in self.critic.items()}
        self.lya = _leafify(weights["lyapunov"])
        self.lya_target = {k: v.detach().clone() for k, v in self.lya.items()}
        self.policy = _leafify(weights["policy"])
        self.backup = _leafify(weights["backup_policy"])
        self.node = _leafify(weights["node"])
        self.log_alpha = torch.zeros(1, requires_grad=True)
        self.backup_log_alpha = torch.zeros(1, requires_grad=True)
        A = torch.optim.Adam
        self.opt = dict(
            critic=A(self.critic.values(), lr=4e-4), lya=A(self.lya.values(), lr=4e-4),
            policy=A(self.policy.values(), lr=args.lr), backup=A(self.backup.values(), lr=args.lr),
            alpha=A([self.log_alpha], lr=args.lr), backup_alpha=A([self.backup_log_alpha], lr=args.lr),
            node=A(self.node.values(), lr=1e-3))
        hi = torch.tensor(env.action_space.high, dtype=torch.float32)
        lo = torch.tensor(env.action_space.low, dtype=torch.float32)
        self.scale, self.bias = (hi - lo) / 2.0, (hi + lo) / 2.0
        self.target_entropy = -float(env.action_space.shape[0])
        self.num_cbfs = self._num_cbfs(env)
        self.num_constraints = self.num_cbfs + 1
        self.lambda_values = [0.0] * self.num_constraints
        self.backup_lambda_values = [0.0] * self.num_cbfs
        self.augmented_term, self.augmented_ratio = 1.0, 1.0005
        self.backup_augmented_term = 1.0
        self.cost_limit = 0.0
        self._setup_task(env)

    adjoint = False           # True: every NODE solve is differentiated by ``odeint_adjoint`` (BASELINE configs[3])

    def _ode(self, y0, t, info, fit=False):
        """The solver call of the rollouts / the NODE fit.  With ``adjoint`` the rollouts integrate no parameter
        adjoint (``adjoint_params=()``: the policy loss needs d/d action only), the fit integrates all of them."""
        if self.adjoint:
            return odeint_adjoint(self.node_fn, y0, t, method=self.solver, atol=1e-7, rtol=1e-5, info=info,
                                  adjoint_params=None if fit else ())
        return odeint(self.node_fn, y0, t, method=self.solver, atol=1e-7, rtol=1e-5, info=info)

    def _backup_due(self, updates):
        return True

    def _lya_train_inputs(self, batch):
        return batch["center"], batch["next_center"]

    def _set_grads(self, params, grads):
        for p, g in zip(params, grads):
            p.grad = g

    def _auglag(self, required, lambdas, updates, with_clf, backup=False):
        """sac_cbf_clf.py:506-528 (primary) / :623-638 (backup)."""
        out = {}
        rho_attr = "backup_augmented_term" if (backup and self.OWN_BACKUP_RHO) else "augmented_term"
        interval = self.args.Lagrangian_multiplier_update_interval * (self._backup_lam_mul() if backup else 1)
        ratio = 1.0
        if with_clf:
            other = torch.abs(torch.mean(required[:-1] - self.cost_limit))
            lyac = torch.abs(required[-1] - self.cost_limit)
            ratio = float(other / lyac)
            if self.RATIO_MIN is not None and ratio < self.RATIO_MIN:
                ratio = self.RATIO_MIN
        req_d = required.detach()
        if updates % interval == 0:
            for i in range(len(lambdas)):
                new = torch.as_tensor(lambdas[i], dtype=torch.float32) + getattr(self, rho_attr) * req_d[i]
                lambdas[i] = float(torch.clamp(new, 0.01, self.LAM_HI))
        setattr(self, rho_attr, min(getattr(self, rho_attr) * self.augmented_ratio, 200))
        rho = getattr(self, rho_attr)
        n_cbf = len(required) - (1 if with_clf else 0)
        loss = 0.0
        for i in range(n_cbf):
            g = required[i] - self.cost_limit
            loss = loss + float(lambdas[i]) * g + rho / 2.0 * g * g
        if with_clf:
            g = required[-1] - self.cost_limit
            loss = loss + float(lambdas[-1]) * ratio * g + ratio * ratio * rho / 2.0 * g * g
        out.update(ratio=ratio, loss=loss)
        return out

    def _backup_lam_mul(self):
        return 1

    # -- one update -----------------------------------------------------------
    def update(self, batch, eps, updates, node_batch=None):
        """``batch``: dict of float32 tensors (obs, action, reward, constraint,
        center, next_center, next_obs, mask); ``eps``: three (B,2) draws in the
        reference order (next_obs sample, obs sample, backup sample)."""
        R = {}
        if node_batch is not None:
            R["node_loss"], R["g_node"] = self.train_step(*node_batch)
        obs, nobs, act = batch["obs"], batch["next_obs"], batch["action"]
        rew, con = batch["reward"].unsqueeze(1), batch["constraint"].unsqueeze(1)
        (cen, ncen), mask = self._lya_train_inputs(batch), batch["mask"].unsqueeze(1)
        dt = self.env.dt
        do_backup = self._backup_due(updates)

        with torch.no_grad():
            na, nlogp, _ = policy_sample(self.policy, nobs, eps[0], self.scale, self.bias)
            q1t, q2t = qnet(self.critic_target, nobs, na)
            next_q = rew + mask * self.gamma * (torch.min(q1t, q2t) - self.alpha * nlogp)
            next_l = con + mask * self.gamma * lyanet(self.lya_target, ncen)
        q1, q2 = qnet(self.critic, obs, act)
        qf1_loss, qf2_loss = F.mse_loss(q1, next_q), F.mse_loss(q2, next_q)
        lf_loss = F.mse_loss(lyanet(self.lya, cen), next_l)
        gc = torch.autograd.grad(qf1_loss + qf2_loss, list(self.critic.values()))
        gl = torch.autograd.grad(lf_loss, list(self.lya.values()))
        self._set_grads(self.critic.values(), gc)
        self.opt["critic"].step()
        self._set_grads(self.lya.values(), gl)
        self.opt["lya"].step()
        R.update(g_critic=_flat(gc), g_lya=_flat(gl), next_q=next_q, next_l=next_l)

        pi, log_pi, _ = policy_sample(self.policy, obs, eps[1], self.scale, self.bias)
        min_q_pi = torch.min(*qnet(self.critic, obs, pi))
        policy_loss_1 = ((self.alpha * log_pi) - min_q_pi).mean()
        if do_backup:
            bpi, blog_pi, _ = policy_sample(self.backup, obs, eps[self.EPS_BACKUP], self.scale, self.bias)
            bmin_q = torch.min(*qnet(self.critic, obs, bpi))
            backup_loss_1 = ((self.backup_alpha * blog_pi) - bmin_q).mean()

        # primary: CLF + CBFs ; backup: CBFs only  (env-specific terms)
        matr, extra = self._primary_terms(batch, pi, eps)
        required = torch.where(matr > 0, matr, torch.zeros_like(matr)).sum(0) / self.batch_size
        al = self._auglag(required, self.lambda_values, updates, True)
        R.update(matr=matr.detach(), required=required.detach(), ratio=al["ratio"], **extra)
        policy_loss = policy_loss_1 + al["loss"]
        gp = torch.autograd.grad(policy_loss, list(self.policy.values()))
        self._set_grads(self.policy.values(), gp)
        self.opt["policy"].step()
        R.update(g_policy=_flat(gp), policy_loss_2=float(al["loss"]))
        if do_backup:
            bmatr, bextra = self._backup_terms(batch, bpi, eps)
            brequired = torch.where(bmatr > 0, bmatr, torch.zeros_like(bmatr)).sum(0) / self.batch_size
            bal = self._auglag(brequired, self.backup_lambda_values, updates, False, backup=True)
            R.update(bmatr=bmatr.detach(), brequired=brequired.detach(), **bextra)
            backup_loss = backup_loss_1 + bal["loss"]
            gb = torch.autograd.grad(backup_loss, list(self.backup.values()))
            self._set_grads(self.backup.values(), gb)
            self.opt["backup"].step()
            R.update(g_backup=_flat(gb), backup_policy_loss_2=float(bal["loss"]),
                     backup_policy_loss_1=float(backup_loss_1))

        alpha_loss = -(self.log_alpha * (log_pi + self.target_entropy).detach()).mean()
        self.log_alpha.grad = torch.autograd.grad(alpha_loss, self.log_alpha)[0]
        self.opt["alpha"].step()
        self.alpha = float(self.log_alpha.exp())
        if do_backup:
            balpha_loss = -(self.backup_log_alpha * (blog_pi + self.target_entropy).detach()).mean()
            self.backup_log_alpha.grad = torch.autograd.grad(balpha_loss, self.backup_log_alpha)[0]
            self.opt["backup_alpha"].step()
            self.backup_alpha = float(self.backup_log_alpha.exp())

        if updates % self.args.target_update_interval == 0:
            with torch.no_grad():
                for tgt, src in ((self.critic_target, self.critic), (self.lya_target, self.lya)):
                    for k in tgt:
                        tgt[k].copy_(tgt[k] * (1.0 - self.tau) + src[k] * self.tau)

        R["ret"] = (float(qf1_loss), float(qf2_loss), float(lf_loss), float(policy_loss_1),
                    float(alpha_loss), float(self.alpha))
        R.update(backup_alpha=self.backup_alpha,
                 lambdas=list(self.lambda_values), backup_lambdas=list(self.backup_lambda_values),
                 augmented_term=self.augmented_term, log_pi=log_pi.detach(), pi=pi.detach())
        return R


class OracleUnicycleAgent(OracleAgentBase):
    """Unicycle copy (U/sac_cbf_clf/sac_cbf_clf.py): 7 CBFs + 1 CLF, control-affine NODE, one-step rollout."""

    def _num_cbfs(self, env):
        return len(env.hazards_locations)

    def _setup_task(self, env):
        self.hazards = torch.tensor(np.asarray(env.hazards_locations), dtype=torch.float32)
        self.node_fn = AffineNode(self.node)

    @staticmethod
    def get_state(obs):
        """dynamics.py:53-58 — atan2 in float64 on the host, cast back."""
        o = obs.detach().double().numpy()
        st = np.zeros((o.shape[0], 3))
        st[:, 0], st[:, 1], st[:, 2] = o[:, 0], o[:, 1], np.arctan2(o[:, 3], o[:, 2])
        return torch.from_numpy(st).float()

    def _rollout(self, state, action, fit=False):
        y0 = torch.cat((state, action), -1)
        t = torch.tensor([0, self.env.dt])
        info = {}
        y = self._ode(y0, t, info, fit)[-1]
        return y[:, :3], info

    def _lookahead(self, st):
        th = st[:, 2]
        return torch.stack([st[:, 0] + L_P * torch.cos(th), st[:, 1] + L_P * torch.sin(th)], 1)

    def _cbf_terms(self, ps, ps_next):
        r = 1.05 * self.env.hazards_radius
        hs = 0.5 * (((ps[:, None, :] - self.hazards[None]) ** 2).sum(2) - r ** 2)
        hn = 0.5 * (((ps_next[:, None, :] - self.hazards[None]) ** 2).sum(2) - r ** 2)
        return -((hn - hs) / self.env.dt) - self.gamma_b * hs

    def train_step(self, node_obs, node_action, node_next_obs):
        """model.py:221-260 via sac_cbf_clf.py:205-219."""
        st, nst = self.get_state(node_obs), self.get_state(node_next_obs)
        self.opt["node"].zero_grad()
        pred, _ = self._rollout(st, node_action, fit=True)
        loss = F.mse_loss(pred, nst)
        g = torch.autograd.grad(loss, list(self.node.values()))
        self._set_grads(self.node.values(), g)
        self.opt["node"].step()
        return float(loss), _flat(g)

    def _primary_terms(self, batch, pi, eps):
        """sac_cbf_clf.py:364-386, 408-530"""
        obs, cen, dt = batch["obs"], batch["center"], self.env.dt
        state = self.get_state(obs)
        V = lyanet(self.lya, cen).detach()
        ps = self._lookahead(state)
        x_next, info = self._rollout(state, pi)
        ps_next = self._lookahead(x_next)
        V_next = lyanet(self.lya, ps_next)
        lya_term = ((V_next - V) / dt) + 1.0 * V
        matr = torch.cat((self._cbf_terms(ps, ps_next), lya_term), 1)
        self._ps = ps
        return matr, dict(x_next=x_next.detach(), ode_info=info)

    def _backup_terms(self, batch, bpi, eps):
        """sac_cbf_clf.py:388-406, 532-640"""
        state = self.get_state(batch["obs"])
        bx_next, binfo = self._rollout(state, bpi)
        return self._cbf_terms(self._ps, self._lookahead(bx_next)), dict(bx_next=bx_next.detach(), bode_info=binfo)


class OracleCarsAgent(OracleAgentBase):
    """SimulatedCars copy (C/sac_cbf_clf/sac_cbf_clf.py): two relative-degree-2 CBFs + 1 CLF, non-affine NODE
    on [x, u, t], two-step rollout with a re-sampled (detached) second action."""
    LAM_HI, RATIO_MIN, N_EPS = 300.0, 0.002, 5

    def _num_cbfs(self, env):
        return 2

    def _setup_task(self, env):
        self.node_fn = ConcatNode(self.node)

    @staticmethod
    def get_state(obs):
        """C/dynamics.py:59-62 (numpy float64 on the host, cast back)."""
        o = obs.detach().double().numpy().copy()
        o[:, ::2] *= 100.0
        o[:, 1::2] *= 30.0
        return torch.from_numpy(o).float()

    @staticmethod
    def get_obs(state):
        """C/dynamics.py:88-91 (torch, in place on a clone)."""
        o = state.clone()
        o[:, ::2] /= 100.0
        o[:, 1::2] /= 30.0
        return o

    def _rollout(self, state, action, t, fit=False):
        y0 = torch.cat((state, action, t), -1)
        ts = torch.tensor([0, self.env.dt])
        info = {}
        y = self._ode(y0, ts, info, fit)[-1]
        return y[:, :10], info

    def train_step(self, node_obs, node_action, node_next_obs, node_t):
        """C/model.py:208-252 via C/sac_cbf_clf.py:201-217."""
        st, nst = self.get_state(node_obs), self.get_state(node_next_obs)
        pred, _ = self._rollout(st, node_action, node_t.reshape(-1, 1), fit=True)
        loss = F.mse_loss(pred, nst)
        g = torch.autograd.grad(loss, list(self.node.values()))
        self._set_grads(self.node.values(), g)
        self.opt["node"].step()
        return float(loss), _flat(g)

    def _terms(self, batch, action, sd_policy, eps_next, with_clf):
        gamma_b, gamma_l, radius = self.gamma_b, 0.15, 4.5
        state = self.get_state(batch["obs"])
        t, nt = batch["t"].reshape(-1, 1), batch["next_t"].reshape(-1, 1)
        x1, info1 = self._rollout(state, action, t)
        with torch.no_grad():
            pi_next, _, _ = policy_sample(sd_policy, self.get_obs(x1.detach()), eps_next, self.scale, self.bias)
        x2, info2 = self._rollout(x1, pi_next, nt)

        def h(x):
            return (x[:, 4:5] - x[:, 6:7]) - radius, (x[:, 6:7] - x[:, 8:9]) - radius
        h23_0, h34_0 = h(state)
        h23_1, h34_1 = h(x1)
        h23_2, h34_2 = h(x2)
        l1_23 = h23_1 - h23_0 + gamma_b * h23_0
        l1_34 = h34_1 - h34_0 + gamma_b * h34_0
        l2_23 = h23_2 - h23_1 + gamma_b * h23_1
        l2_34 = h34_2 - h34_1 + gamma_b * h34_1
        cbf23 = -(l2_23 - l1_23) - gamma_b * l1_23
        cbf34 = -(l2_34 - l1_34) - gamma_b * l1_34
        cols = [cbf23, cbf34]
        if with_clf:
            V = lyanet(self.lya, batch["center"]).detach()
            V1 = lyanet(self.lya, x1[:, 4:8])
            cols.append((V1 - V) + gamma_l * V)
        return torch.cat(cols, 1), x1.detach(), x2.detach(), (info1, info2)

    def _primary_terms(self, batch, pi, eps):
        matr, x1, x2, info = self._terms(batch, pi, self.policy, eps[3], True)
        return matr, dict(x_next=x1, x_next2=x2, ode_info=info[0])

    def _backup_terms(self, batch, bpi, eps):
        bmatr, x1, x2, info = self._terms(batch, bpi, self.backup, eps[4], False)
        return bmatr, dict(bx_next=x1, bx_next2=x2, bode_info=info[0])


class OraclePvtolAgent(OracleAgentBase):
    """Pvtol copy (P = NLBAC_pvtol_RL_training/Pvtol_RL_training, sac_cbf_clf/sac_cbf_clf.py): control-affine NODE on
    the 6 dynamic states, three-step rollout with two re-sampled detached actions, relative-degree-3 CBFs (5 hazards,
    2 safety-operator distances, y_max, y_min) + 1 CLF on the predicted observation; the backup controller is
    updated every ``backup_update_interval`` updates with its own augmented term (:282-305, 1033-1034)."""
    LAM_HI, RATIO_MIN, N_EPS = 400.0, 0.002, 7
    EPS_BACKUP, OWN_BACKUP_RHO = 4, True
    GOAL = (4.5, 4.5)

    def _num_cbfs(self, env):
        return len(env.hazard_locations) + 4

    def _setup_task(self, env):
        self.hazards = torch.tensor(np.asarray(env.hazard_locations), dtype=torch.float32)
        self.node_fn = AffineNode(self.node, n_s=6, n_u=2)
        self.backup_update_interval = int(getattr(self.args, "backup_update_interval", 20))

    def _backup_due(self, updates):
        return updates % self.backup_update_interval == 0

    def _backup_lam_mul(self):
        return self.backup_update_interval

    def _lya_train_inputs(self, batch):
        return batch["obs"], batch["next_obs"]          # P:243-252: the Lyapunov critic is trained on observations

    @staticmethod
    def get_state(obs):
        """P/sac_cbf_clf/dynamics.py:50-66 — float64 on the host, cast back; (state7, state_dynamics6)."""
        o = obs.detach().double().numpy()
        st = np.zeros((o.shape[0], 7))
        st[:, 0], st[:, 1], st[:, 2] = o[:, 0], o[:, 1], np.arctan2(o[:, 3], o[:, 2])
        st[:, 3], st[:, 4], st[:, 5], st[:, 6] = o[:, 4], o[:, 5], o[:, 6], o[:, 7]
        st = torch.from_numpy(st).float()
        return st, st[:, :6]

    def get_obs(self, st7):
        """P/sac_cbf_clf/dynamics.py:97-153 (differentiable)."""
        c, s_ = torch.cos(st7[:, 2]), torch.sin(st7[:, 2])
        rx, ry = self.GOAL[0] - st7[:, 0], self.GOAL[1] - st7[:, 1]
        dist = torch.sqrt(rx * rx + ry * ry)
        v0, v1 = c * rx + s_ * ry, -s_ * rx + c * ry
        div = torch.sqrt(v0 * v0 + v1 * v1) + 0.001
        return torch.stack([st7[:, 0], st7[:, 1], c, s_, st7[:, 3], st7[:, 4], st7[:, 5], st7[:, 6], v0 / div,
                            v1 / div, torch.exp(-dist)], 1)

    def _rollout(self, state6, action, fit=False):
        y0 = torch.cat((state6, action), -1)
        t = torch.tensor([0, self.env.dt])
        info = {}
        y = self._ode(y0, t, info, fit)[-1]
        return y[:, :6], info

    def train_step(self, node_obs, node_action, node_next_obs):
        """P/model.py:224-266 via P/sac_cbf_clf.py:205-219."""
        (_, st), (_, nst) = self.get_state(node_obs), self.get_state(node_next_obs)
        pred, _ = self._rollout(st, node_action, fit=True)
        loss = F.mse_loss(pred, nst)
        g = torch.autograd.grad(loss, list(self.node.values()))
        self._set_grads(self.node.values(), g)
        self.opt["node"].step()
        return float(loss), _flat(g)

    def _step7(self, prev7, x6):
        """Append the safety operator's next x-position (P:462-470)."""
        op = prev7[:, 6] + self.env.safety_operator_follow * (x6[:, 0] - prev7[:, 6])
        return torch.cat((x6, op.unsqueeze(1)), 1)

    def _three_steps(self, obs, a0, policy, eps_a, eps_b):
        st7, st6 = self.get_state(obs)
        x1, info = self._rollout(st6, a0)
        s1 = self._step7(st7, x1)
        obs1 = self.get_obs(s1)
        a1, _, _ = policy_sample(policy, obs1.detach(), eps_a, self.scale, self.bias)
        x2, _ = self._rollout(x1, a1.detach())
        s2 = self._step7(s1, x2)
        a2, _, _ = policy_sample(policy, self.get_obs(s2).detach(), eps_b, self.scale, self.bias)
        x3, _ = self._rollout(x2, a2.detach())
        s3 = self._step7(s2, x3)
        return (st7, s1, s2, s3), obs1, info

    def _rd3(self, h0, h1, h2, h3):
        """Relative-degree-3 composition, in the reference's operation order (P:575-587)."""
        gb = self.gamma_b
        t1 = h3 - h2 + gb * h2
        t2 = h2 - h1 + gb * h1
        t3 = h1 - h0 + gb * h0
        return -(t1 - t2 + gb * t2 - (t2 - t3 + gb * t3) + gb * (t2 - t3 + gb * t3))

    def _cbf_terms(self, states):
        env = self.env
        r = 1.2 * env.hazards_radius
        hz = [0.5 * (((s_[:, None, :2] - self.hazards[None]) ** 2).sum(2) - r ** 2) for s_ in states]
        d = 0.9 * env.operator_dist
        h1 = [(s_[:, 0] - s_[:, 6] + d).unsqueeze(1) for s_ in states]
        h2 = [(-s_[:, 0] + s_[:, 6] + d).unsqueeze(1) for s_ in states]
        h3 = [(-s_[:, 1] + env.y_max - 10.0).unsqueeze(1) for s_ in states]
        h4 = [(s_[:, 1] - env.y_min - 10.0).unsqueeze(1) for s_ in states]
        return torch.cat([self._rd3(*h) for h in (hz, h1, h2, h3, h4)], 1)

    def _primary_terms(self, batch, pi, eps):
        """P:376-399, 424-757"""
        states, obs1, info = self._three_steps(batch["obs"], pi, self.policy, eps[2], eps[3])
        V = lyanet(self.lya, batch["center"]).detach()
        lya_term = ((lyanet(self.lya, obs1) - V) / 1.0) + 0.1 * V
        matr = torch.cat((self._cbf_terms(states), lya_term), 1)
        return matr, dict(x_next=states[1][:, :6].detach(), x_next2=states[2][:, :6].detach(),
                          x_next3=states[3][:, :6].detach(), ode_info=info)

    def _backup_terms(self, batch, bpi, eps):
        """P:401-422, 759-1048"""
        states, _, binfo = self._three_steps(batch["obs"], bpi, self.backup, eps[5], eps[6])
        return self._cbf_terms(states), dict(bx_next=states[1][:, :6].detach(), bx_next3=states[3][:, :6].detach(),
                                             bode_info=binfo)


class OracleUnicycleBarrierAgent(OracleUnicycleAgent):
    """Learned-barrier-certificate Unicycle copy (NU = neural_barrier_certificate/
    neural_barrier_certificate_NLBAC_Unicycle_RL_training/Unicycle_RL_training): one policy, a BarrierNetwork
    B(obs, a) trained TD-style on the env's barrier signal (NU/sac_cbf_clf/sac_cbf_clf.py:214-246), one learned
    CBF term -(B(obs', a') - B) - gamma_b B with obs' = get_obs(x') differentiable and a' re-sampled and detached
    (:404-420), CLF as in U, no ratio in the loss (:474-475)."""
    N_EPS = 3
    GOAL = (2.5, 2.5)
    CLF_DT, GAMMA_L, RATIO_MIN = None, 1.0, None      # CLF divisor (None: env.dt), class-K coefficient, ratio clamp

    def __init__(self, env, args, weights, solver="euler"):
        self.env, self.args, self.solver = env, args, solver
        self.gamma, self.gamma_b, self.tau = args.gamma, args.gamma_b, args.tau
        self.alpha, self.batch_size = args.alpha, args.batch_size
        self.critic, self.lya, self.barrier = (_leafify(weights[k]) for k in ("critic", "lyapunov", "barrier"))
        clone = lambda sd: {k: v.detach().clone() for k, v in sd.items()}
        self.critic_target, self.lya_target, self.barrier_target = clone(self.critic), clone(self.lya), clone(self.barrier)
        self.policy, self.node = _leafify(weights["policy"]), _leafify(weights["node"])
        self.log_alpha = torch.zeros(1, requires_grad=True)
        A = torch.optim.Adam
        self.opt = dict(critic=A(self.critic.values(), lr=4e-4), lya=A(self.lya.values(), lr=4e-4),
                        barrier=A(self.barrier.values(), lr=4e-4), policy=A(self.policy.values(), lr=args.lr),
                        alpha=A([self.log_alpha], lr=args.lr), node=A(self.node.values(), lr=1e-3))
        hi = torch.tensor(env.action_space.high, dtype=torch.float32)
        lo = torch.tensor(env.action_space.low, dtype=torch.float32)
        self.scale, self.bias = (hi - lo) / 2.0, (hi + lo) / 2.0
        self.target_entropy = -float(env.action_space.shape[0])
        self.num_cbfs, self.num_constraints = 1, 2
        self.lambda_values = [0.0, 0.0]
        self.augmented_term, self.augmented_ratio, self.cost_limit = 1.0, 1.0005, 0.0
        self._setup_task(env)

    def get_obs(self, st):
        """NU/sac_cbf_clf/dynamics.py:92-135 (differentiable): [x, y, cos, sin, compass(2), exp(-dist)]."""
        c, s_ = torch.cos(st[:, 2]), torch.sin(st[:, 2])
        rx, ry = self.GOAL[0] - st[:, 0], self.GOAL[1] - st[:, 1]
        dist = torch.sqrt(rx * rx + ry * ry)
        v0, v1 = c * rx + s_ * ry, -s_ * rx + c * ry
        div = torch.sqrt(v0 * v0 + v1 * v1) + 0.001
        return torch.stack([st[:, 0], st[:, 1], c, s_, v0 / div, v1 / div, torch.exp(-dist)], 1)

    def _predict(self, obs, pi):
        """(x', get_obs(x') differentiable, input of V', solver info): one NODE step under ``pi``."""
        x_next, info = self._rollout(self.get_state(obs), pi)
        return x_next, self.get_obs(x_next), self._lookahead(x_next), info

    def _lya_inputs(self, batch):
        return batch["center"], batch["next_center"]

    def update(self, batch, eps, updates, node_batch=None):
        """NU/sac_cbf_clf/sac_cbf_clf.py:155-280; ``eps``: (next_obs sample, obs sample, sample on the predicted
        next observation inside the loss)."""
        R = {}
        if node_batch is not None:
            R["node_loss"], R["g_node"] = self.train_step(*node_batch)
        obs, nobs, act = batch["obs"], batch["next_obs"], batch["action"]
        rew, con, sig = (batch[k].unsqueeze(1) for k in ("reward", "constraint", "barrier_signal"))
        (cen, ncen), mask = self._lya_inputs(batch), batch["mask"].unsqueeze(1)
        dt = self.CLF_DT or self.env.dt
        with torch.no_grad():
            na, nlogp, _ = policy_sample(self.policy, nobs, eps[0], self.scale, self.bias)
            q1t, q2t = qnet(self.critic_target, nobs, na)
            next_q = rew + mask * self.gamma * (torch.min(q1t, q2t) - self.alpha * nlogp)
            next_l = con + mask * self.gamma * lyanet(self.lya_target, ncen)
            next_b = sig + mask * self.gamma * lyanet(self.barrier_target, torch.cat([nobs, na], 1))
        q1, q2 = qnet(self.critic, obs, act)
        qf1_loss, qf2_loss = F.mse_loss(q1, next_q), F.mse_loss(q2, next_q)
        lf_loss = F.mse_loss(lyanet(self.lya, cen), next_l)
        bf_loss = F.mse_loss(lyanet(self.barrier, torch.cat([obs, act], 1)), next_b)
        for name, loss, sd in (("critic", qf1_loss + qf2_loss, self.critic), ("lya", lf_loss, self.lya),
                               ("barrier", bf_loss, self.barrier)):
            g = torch.autograd.grad(loss, list(sd.values()))
            self._set_grads(sd.values(), g)
            self.opt[name].step()
            R["g_" + name] = _flat(g)
        R.update(next_q=next_q, next_l=next_l, next_b=next_b, barrier_loss=float(bf_loss))

        pi, log_pi, _ = policy_sample(self.policy, obs, eps[1], self.scale, self.bias)
        policy_loss_1 = ((self.alpha * log_pi) - torch.min(*qnet(self.critic, obs, pi))).mean()
        # get_cbf_clf_part / get_policy_loss_2 (:339-477)
        V = lyanet(self.lya, batch["center"]).detach()
        x_next, obs_pred, v_in, info = self._predict(obs, pi)
        V_next = lyanet(self.lya, v_in)
        lya_term = ((V_next - V) / dt) + self.GAMMA_L * V
        Bv = lyanet(self.barrier, torch.cat([obs, pi], 1)).detach()
        pi_next, _, _ = policy_sample(self.policy, obs_pred.detach(), eps[2], self.scale, self.bias)
        B_next = lyanet(self.barrier, torch.cat([obs_pred, pi_next.detach()], 1))
        barrier_term = -(B_next - Bv) - self.gamma_b * Bv
        matr = torch.cat((barrier_term, lya_term), 1)
        required = torch.where(matr > 0, matr, torch.zeros_like(matr)).sum(0) / self.batch_size
        req_d = required.detach()
        if updates % self.args.Lagrangian_multiplier_update_interval == 0:
            for i in range(2):
                new = torch.as_tensor(self.lambda_values[i], dtype=torch.float32) + self.augmented_term * req_d[i]
                self.lambda_values[i] = float(torch.clamp(new, 0.01, 400.0))
        self.augmented_term = min(self.augmented_term * self.augmented_ratio, 200)
        rho = self.augmented_term
        ratio = 1.0
        if self.RATIO_MIN is not None:          # NP:440-447; NU has no ratio
            ratio = max(float(torch.abs(torch.mean(required[:-1] - self.cost_limit)) /
                              torch.abs(required[-1] - self.cost_limit)), self.RATIO_MIN)
        g0, g1 = required[0] - self.cost_limit, required[1] - self.cost_limit
        loss2 = float(self.lambda_values[0]) * g0 + rho / 2.0 * g0 * g0
        loss2 = loss2 + float(self.lambda_values[1]) * ratio * g1 + ratio * ratio * rho / 2.0 * g1 * g1
        gp = torch.autograd.grad(policy_loss_1 + loss2, list(self.policy.values()))
        self._set_grads(self.policy.values(), gp)
        self.opt["policy"].step()
        alpha_loss = -(self.log_alpha * (log_pi + self.target_entropy).detach()).mean()
        self.log_alpha.grad = torch.autograd.grad(alpha_loss, self.log_alpha)[0]
        self.opt["alpha"].step()
        self.alpha = float(self.log_alpha.exp())
        if updates % self.args.target_update_interval == 0:
            with torch.no_grad():
                for tgt, src in ((self.critic_target, self.critic), (self.lya_target, self.lya),
                                 (self.barrier_target, self.barrier)):
                    for k in tgt:
                        tgt[k].copy_(tgt[k] * (1.0 - self.tau) + src[k] * self.tau)
        R["ret"] = (float(qf1_loss), float(qf2_loss), float(lf_loss), float(policy_loss_1), float(alpha_loss),
                    float(self.alpha))
        R.update(g_policy=_flat(gp), policy_loss_2=float(loss2), matr=matr.detach(), required=req_d,
                 lambdas=list(self.lambda_values), augmented_term=self.augmented_term, x_next=x_next.detach(),
                 ode_info=info, log_pi=log_pi.detach(), pi=pi.detach(), obs_pred=obs_pred.detach(),
                 pi_next=pi_next.detach())
        return R


class OraclePvtolBarrierAgent(OracleUnicycleBarrierAgent):
    """Learned-barrier-certificate Pvtol copy (NP = neural_barrier_certificate/
    neural_barrier_certificate_NLBAC_pvtol_RL_training/Pvtol_RL_training/sac_cbf_clf/sac_cbf_clf.py:150-480): the NU
    update on Pvtol's state / observation maps, one NODE step, CLF (V' - V)/1 + 0.1 V on the predicted observation,
    ratio clamped at 0.002; the Lyapunov critic is regressed on observations (:206-217)."""
    GOAL = (4.5, 4.5)
    CLF_DT, GAMMA_L, RATIO_MIN = 1.0, 0.1, 0.002
    get_state = staticmethod(OraclePvtolAgent.get_state)
    get_obs = OraclePvtolAgent.get_obs
    _rollout = OraclePvtolAgent._rollout
    _step7 = OraclePvtolAgent._step7
    train_step = OraclePvtolAgent.train_step

    def _setup_task(self, env):
        self.node_fn = AffineNode(self.node, n_s=6, n_u=2)

    def _lya_inputs(self, batch):
        return batch["obs"], batch["next_obs"]

    def _predict(self, obs, pi):
        st7, st6 = self.get_state(obs)
        x_next, info = self._rollout(st6, pi)
        obs_pred = self.get_obs(self._step7(st7, x_next))
        return x_next, obs_pred, obs_pred, info


class OracleQuadrotorLikeAgent(OracleUnicycleBarrierAgent):
    """BASELINE configs[4] as far as the reference's prose goes (/root/reference/README.md:66-72, 190-192; the code is
    an empty submodule, so there is NOTHING to pin this against — PARITY UNPINNED): the NP update (one controller,
    learned barrier on the signals D1 / D2, CLF (V' - V)/1 + 0.1 V, ratio clamped at 0.002, Lyapunov critic on
    observations) over a single-net NODE on normalised [state (6) | action (2)] with de-normalised outputs; the
    observation is the state."""
    CLF_DT, GAMMA_L, RATIO_MIN = 1.0, 0.1, 0.002

    def _setup_task(self, env):
        self.node_fn = ConcatNode(self.node, n_s=6, n_carry=2, depth=4, norm=env.node_normalizer)

    @staticmethod
    def get_state(obs):
        return obs.detach().double().float()

    def get_obs(self, st):
        return st

    def _rollout(self, state, action, fit=False):
        y0 = torch.cat((state, action), -1)
        t = torch.tensor([0, self.env.dt])
        info = {}
        y = self._ode(y0, t, info, fit)[-1]
        return y[:, :6], info

    def train_step(self, node_obs, node_action, node_next_obs):
        st, nst = self.get_state(node_obs), self.get_state(node_next_obs)
        pred, _ = self._rollout(st, node_action, fit=True)
        loss = F.mse_loss(pred, nst)
        g = torch.autograd.grad(loss, list(self.node.values()))
        self._set_grads(self.node.values(), g)
        self.opt["node"].step()
        return float(loss), _flat(g)

    def _lya_inputs(self, batch):
        return batch["obs"], batch["next_obs"]

    def _predict(self, obs, pi):
        x_next, info = self._rollout(self.get_state(obs), pi)
        return x_next, x_next, x_next, info


def make_oracle(env, args, weights, solver="euler", adjoint=False):
    kind = env.dynamics_mode + ("Barrier" if "barrier" in weights else "")
    cls = {"QuadrotorBarrier": OracleQuadrotorLikeAgent, "Unicycle": OracleUnicycleAgent, "SimulatedCars": OracleCarsAgent, "Pvtol": OraclePvtolAgent,
            "PvtolBarrier": OraclePvtolBarrierAgent, "UnicycleBarrier": OracleUnicycleBarrierAgent}[kind]
    agent = cls(env, args, weights, solver)
    agent.adjoint = bool(adjoint)
    return agent
