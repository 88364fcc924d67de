"""Environment-specific halves of the update (the reference keeps one copy of
``sac_cbf_clf.py`` per environment; here the shared SAC / Lyapunov machinery
lives in ``sac_cbf_clf.SAC_CBF_CLF`` and each environment contributes a task):

  * minibatch row layout dims, NODE model shape, number of policy-noise draws;
  * the NODE fit (``train_step``);
  * the rollout of the learned dynamics under both controllers, the CBF / CLF
    terms, and the gradient of the augmented-Lagrangian loss w.r.t. the actions.

``UnicycleTask``   U/sac_cbf_clf/sac_cbf_clf.py:364-640, U/sac_cbf_clf/model.py:177-260
``CarsTask``       C/sac_cbf_clf/sac_cbf_clf.py:364-681, C/sac_cbf_clf/model.py:179-252
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..arena import io_array, mlp_array, stream_ptr
from ..odeint import AffineNodeSolver, ConcatNodeSolver
from . import _layout as SC
from .model import NeuralODEModel


class _Task:
    name = None
    rollout_waits = 1        # adaptive solves chained inside one update's rollout: each ends in a host wait (dopri5)
    obs_dim = act_dim = lya_dim = n_s = 0
    n_eps = 3                 # N(0,1) draws per update: next-obs sample, obs sample, backup sample [, ...]
    lam_hi = 400.0
    ratio_mode = 1            # 1 plain ratio (U), 2 clamped at 0.002 (C)
    graph_ok = False          # whole update replayable as hipGraphs
    n_pol = 2                 # controllers trained per update (primary + backup); 1 in the learned-barrier copies
    backup_mode = 1           # nlbac_auglag: 0 no backup, 1 backup shares rho, 2 own rho
    has_signal = False        # replay rows carry a barrier signal (learned-barrier copies)
    n_extra_critics = 0       # critic-type nets trained beside Q1, Q2, L (BarrierNet)
    backup_interval = 1       # the backup controller is trained every n-th update (Pvtol: 20)
    eps_order = None          # device noise slot -> index in the reference's draw order (None: identical)

    def __init__(self, agent, env, args):
        self.agent, self.env = agent, env

    def z(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.agent.device)

    def reserve(self, solver, n, P):
        """Pre-allocate a solver's buffers for (n rows, P problems, the agent's current method) once."""
        if not self.agent.fold_launches:
            solver.interp_fold = False        # (NLBAC_FOLD=0: the interpolation launches too, as every other folded step)
            solver.norm_defer = False         # ... and the norms with their elections / as a launch over the error rows
        key = (id(solver), n, P, self.agent.solver)
        seen = self.__dict__.setdefault("_reserved", set())
        if key not in seen:
            seen.add(key)
            solver.reserve(n, P, self.agent.solver)

    def policy_sample(self, ws, key, nets, io, n_nets, B, heads, eps, n_u, action, action_ld, logp):
        """pi(. | obs) of ``n_nets`` stacked policies on B rows each + GaussianPolicy.sample of every row (model.py:116-128):
        one launch — the MLP launch applies the head itself (nlbac_gauss_head) — or, with the launch folds off
        (NLBAC_FOLD=0), the forward and nlbac_gauss_sample_fwd on its (n_nets * B, 2 n_u) output ``heads``."""
        a, s, pol = self.agent, stream_ptr(), self.agent.policy
        p_scale, p_bias = pol.action_scale.data_ptr(), pol.action_bias.data_ptr()
        if not a.fold_launches:
            _lib.call("nlbac_mlp_fwd", nets, io, n_nets, B, s)
            _lib.call("nlbac_gauss_sample_fwd", heads.data_ptr(), 2 * n_u, eps.data_ptr(), p_scale, p_bias, n_u, n_nets * B,
                      action.data_ptr(), action_ld, logp.data_ptr(), s)
            return
        hs = ws.__dict__.setdefault("_gauss_heads", {})
        gh = hs.get(key)
        if gh is None:
            gh = hs[key] = _lib.GaussHead()
            gh.eps, gh.scale, gh.bias, gh.n_u = eps.data_ptr(), p_scale, p_bias, n_u
            gh.action, gh.action_ld, gh.logp = action.data_ptr(), action_ld, logp.data_ptr()
        _lib.call("nlbac_mlp_fwd_gauss", nets, io, n_nets, B, C.byref(gh), s)

    def n_pol_now(self, updates):
        return self.n_pol

    def backup_lam_due(self, updates, interval):
        return 1 if updates % interval == 0 else 0

    def fit_due(self, i_episode):
        return True

    def lya_train_cols(self, lay):
        """Columns of the minibatch row the Lyapunov critic is regressed on: (input, next input)."""
        return lay.lya, lay.nlya

    # value-only nets riding in the Q(s, pi) launch besides V(current Lyapunov input)
    def extra_value_nets(self):
        return []

    def extra_value_io(self, ws, io, i):
        pass


# =====================================================================================
class UnicycleTask(_Task):
    name = "Unicycle"
    obs_dim, act_dim, lya_dim, n_s = 7, 2, 2, 3
    graph_ok = True
    l_p = 0.03

    def __init__(self, agent, env, args):
        super().__init__(agent, env, args)
        self.num_cbfs = len(env.hazards_locations)
        self.gamma_l = 1.0

    def build_node(self):
        return NeuralODEModel(3, 3, 6)

    def setup(self):
        a = self.agent
        self.hazards = torch.tensor(np.asarray(self.env.hazards_locations), dtype=torch.float32,
                                    device=a.device).contiguous()
        self.solver = AffineNodeSolver(a.neural_ode_model, a.device)      # policy-loss rollouts (2B rows)
        self.solver.keep_acts = False                                     # differentiated w.r.t. the actions only
        self.fit_solver = AffineNodeSolver(a.neural_ode_model, a.device)  # NODE fit rollouts
        self.solvers = [self.solver, self.fit_solver]

    def alloc(self, ws):
        B, z, H = ws.B, self.z, self.agent.hidden
        ws.ps = z(B, 2)
        ws.y0_2 = z(2 * B, 3)
        ws.V, ws.Vn, ws.dVn = z(B), z(B), z(B)
        ws.acts_vn = z(2, B, H)
        ws.ps_next2, ws.dps_next2, ws.dps_v2 = z(2 * B, 2), z(2 * B, 2), z(2 * B, 2)
        ws.matr, ws.bmatr = z(B, self.num_cbfs + 1), z(B, self.num_cbfs)
        ws.part_c = z(ws.nblk, 2 * self.num_cbfs + 1)
        # (the constraint head of the V(p(x')) forward: column sums per 16-row tile, two-level election)
        n16 = (B + 15) // 16
        ws.part_c16 = z(n16, 2 * self.num_cbfs + 1)
        ws.tickets_c = torch.zeros(2 + n16 // 16 + 1, dtype=torch.int32, device=self.agent.device)
        ws.dx_next2 = z(2 * B, 3)

    def plan(self, ws, P):
        a, lay = self.agent, self.agent.lay
        P.n_l = mlp_array([a.h_l.desc])
        io = P.io_vn = io_array(1)                 # V(p(x')) forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.ps_next2.data_ptr(), 2, 2
        io[0].y, io[0].y_ld = ws.Vn.data_ptr(), 1
        io[0].acts = ws.acts_vn.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dVn.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dps_v2.data_ptr(), 2
        P.io_vc = io_array(1)                      # V(centre), value only
        P.io_vc[0].x0, P.io_vc[0].x0_dim, P.io_vc[0].x0_ld = ws.mb.data_ptr() + 4 * lay.lya, 2, lay.LD
        P.io_vc[0].y, P.io_vc[0].y_ld = ws.V.data_ptr(), 1
        # the data backward of V(p(x')) and of the Q(s, pi) nets do not depend on each other and are both due when the
        # constraints' gradients exist: one launch (the Q nets' dL/dq from the dy head, V's from io.dy) instead of a
        # 128-tile launch on 256 CUs followed by a second one
        n2 = 2 * P.NP
        P.n_q5v = mlp_array([a.h_q1.desc, a.h_q2.desc] * P.NP + [a.h_l.desc])
        P.io_q5v = io_array(n2 + 1)
        for i in range(n2):
            C.memmove(C.byref(P.io_q5v, i * C.sizeof(_lib.MlpIO)), C.byref(P.io_q5, i * C.sizeof(_lib.MlpIO)), C.sizeof(_lib.MlpIO))
        C.memmove(C.byref(P.io_q5v, n2 * C.sizeof(_lib.MlpIO)), C.byref(P.io_vn, 0), C.sizeof(_lib.MlpIO))
        # likewise forward: Q(s, pi) [+ V(centre)] is the last piece of part 1 and V(p(x')) the first launch behind the
        # rollout — when that piece is still pending at that point, both go out as one launch
        nq = P.n_q5_count
        P.n_q5f = mlp_array([a.h_q1.desc, a.h_q2.desc] * P.NP + [a.h_l.desc] * (nq - n2) + [a.h_l.desc])
        P.io_q5f = io_array(nq + 1)
        for i in range(nq):
            C.memmove(C.byref(P.io_q5f, i * C.sizeof(_lib.MlpIO)), C.byref(P.io_q5, i * C.sizeof(_lib.MlpIO)), C.sizeof(_lib.MlpIO))
        C.memmove(C.byref(P.io_q5f, nq * C.sizeof(_lib.MlpIO)), C.byref(P.io_vn, 0), C.sizeof(_lib.MlpIO))

    # V(centre) rides in the 5-net launch of the shared part: tell it where to write
    def value_now_io(self, ws, io, i):
        lay = self.agent.lay
        io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.mb.data_ptr() + 4 * lay.lya, self.lya_dim, lay.LD
        io[i].y, io[i].y_ld = ws.V.data_ptr(), 1

    # -- rollout under both controllers ------------------------------------------------------
    def rollout_begin(self, ws, P):
        a, s = self.agent, stream_ptr()
        B, LD = ws.B, a.lay.LD
        p_obs = ws.mb.data_ptr()
        # state (twice: primary and backup rows of the rollout) and look-ahead point: formed by the rollout's first
        # launch (fused solver), else by a launch of their own
        if self.solver.fused and a.fold_launches:
            self.solver.set_in_map(1, ws.mb, LD, self.l_p, ws.ps)
        else:
            _lib.call("nlbac_unicycle_state", p_obs, LD, B, self.l_p, ws.y0_2.data_ptr(), 2, ws.ps.data_ptr(), s)
        self.reserve(self.solver, 2 * B, 2)
        # the look-ahead point of x(t + dt) and its backward ride in the solver's interpolation launches where those
        # exist (device-driven dopri5); elsewhere this task launches them (loss_and_backward)
        if a.fold_launches:
            self.solver.set_out_map(1, self.l_p, ws.ps_next2, ws.dps_next2, ws.dps_v2)
        else:
            self.solver._out_map = None
        self.solver.forward_begin(ws.y0_2, ws.pi2, 2, B, a.solver, float(self.env.dt), a.atol, a.rtol)

    def loss_and_backward(self, ws, P, lam_upd, assume_single):
        """Constraint terms, augmented-Lagrangian scalars and d loss / d actions (2B, act_dim)."""
        a, s, call = self.agent, stream_ptr(), _lib.call
        B, sc, dt = ws.B, a.sc.data_ptr(), float(self.env.dt)
        x_next2 = self.solver.forward_finish(assume_single_step=assume_single)
        mapped = self.solver.out_mapped
        # the constraint terms ride in V(p(x'))'s own launch where its kernels evaluate them (below); that launch is then
        # never merged with the pending Q(s, pi) forward: V(c), which the CLF term needs, would be computed by another
        # net's workgroups of the same launch — and eager and captured updates keep the same launches' arithmetic
        use_head = self._constraint_head_ok(P.n_l, 1, ws)
        merged = mapped and a.world == 1 and a.fold_launches and not a.h_extra and len(a._fill) == 1 and not use_head
        if merged:
            a._fill.clear()      # (the pending piece is exactly the Q(s, pi) forward: it rides with V(p(x')) below)
        a.drain_fill()           # what is left of part 1 (critic step, Q(s, pi)): everything below uses the stepped nets
        if not mapped:
            call("nlbac_unicycle_lookahead", x_next2.data_ptr(), 2 * B, self.l_p, ws.ps_next2.data_ptr(), s)
        r_coll = 1.05 * float(self.env.hazards_radius)
        nets, io, cnt = (P.n_q5f, P.io_q5f, P.n_q5_count + 1) if merged else (P.n_l, P.io_vn, 1)
        P.cf_job = None
        if use_head:
            # the constraint terms, their column sums and the augmented-Lagrangian step are the epilogue of V(p(x'))'s
            # workgroups in this launch (nlbac_gauss_head::cf_kind 1): no nlbac_unicycle_constraints_fwd launch
            A = a.auglag_fused(ws, self.num_cbfs, lam_upd)[0]._obj
            G = P.__dict__.get("cf_head")       # (built once per plan: ~40 ctypes field stores sit between the accept
            if G is None:                       #  decision and this launch; only the lambda-update flags change per update)
                G = P.cf_head = _lib.GaussHead()
                G.cf_kind, G.cf_net, G.cf_nh = 1, cnt - 1, self.num_cbfs
                G.cf_ps, G.cf_ps_next, G.cf_V, G.cf_hazards = ws.ps.data_ptr(), ws.ps_next2.data_ptr(), ws.V.data_ptr(), self.hazards.data_ptr()
                rc = float(np.float32(r_coll))           # (r_coll^2 as nlbac_unicycle_constraints_fwd forms it from its float argument)
                G.cf_r2, G.cf_dt, G.cf_gamma_b, G.cf_gamma_l = float(np.float32(rc * rc)), dt, float(a.gamma_b), self.gamma_l
                G.cf_matr, G.cf_bmatr = ws.matr.data_ptr(), ws.bmatr.data_ptr()
                G.cf_partials, G.cf_tickets, G.cf_sc = ws.part_c16.data_ptr(), ws.tickets_c.data_ptr(), sc
                G.cf_n_cbf, G.cf_n_clf, G.cf_batch_size = A.n_cbf, A.n_clf, A.batch_size
                G.cf_ratio_mode, G.cf_backup_mode, G.cf_lam_lo, G.cf_lam_hi = A.ratio_mode, A.backup_mode, A.lam_lo, A.lam_hi
            assert G.cf_net == cnt - 1
            G.cf_do_lambda_update, G.cf_do_backup_lambda_update = A.do_lambda_update, A.do_backup_lambda_update
            cf_defer = a._sums_defer()
            if cf_defer:
                # ... without the election and the step: the tiles' column sums go out, the workgroups of the constraint
                # backward (below) sum them and run the step on a private copy of the scalars block, a workgroup of the
                # actors' data backward commits it (nlbac_gauss_head::cf_defer, nlbac_dy_head::cb_defer, nlbac_head_sums 4)
                G.cf_defer, G.cf_tiles = 1, ws.sums_tiles.data_ptr() + 8
                P.cf_job = (ws.part_c16.data_ptr(), ws.sums_tiles.data_ptr() + 8, A, sc, ws.sc_stage.data_ptr())
            call("nlbac_mlp_fwd_head", nets, io, cnt, B, C.byref(G), s)
        else:
            call("nlbac_mlp_fwd", nets, io, cnt, B, s)
            call("nlbac_unicycle_constraints_fwd", ws.ps.data_ptr(), ws.ps_next2.data_ptr(), ws.V.data_ptr(),
                 ws.Vn.data_ptr(), self.hazards.data_ptr(), self.num_cbfs, r_coll, dt, float(a.gamma_b), self.gamma_l, B,
                 ws.matr.data_ptr(), ws.bmatr.data_ptr(), ws.part_c.data_ptr(), *a.auglag_fused(ws, self.num_cbfs, lam_upd), s)
        a.auglag(ws, self.num_cbfs, lam_upd)
        if a.world == 1 and a.fold_launches:
            # dV_next -> d ps_next (rows [0,B)), together with the Q(s, pi) nets' dx; the constraint backward itself (d ps_next
            # of both controllers from the CBF terms, dV_next from the CLF term) is the prologue of V's workgroups in that
            # launch (nlbac_dy_head::cb_kind 1): no nlbac_unicycle_constraints_bwd launch
            NP = ws.np_now
            H = P.__dict__.get("head_actor_q_cb")
            if H is None:
                H = P.head_actor_q_cb = _lib.DyHead.from_buffer_copy(a._actor_q_head(ws, P, NP, B * a.world))
                H.cb_kind, H.cb_nh = 1, self.num_cbfs
                H.cb_ps_next, H.cb_matr, H.cb_bmatr = ws.ps_next2.data_ptr(), ws.matr.data_ptr(), ws.bmatr.data_ptr()
                H.cb_hazards, H.cb_sc, H.cb_dt, H.cb_batch = self.hazards.data_ptr(), sc, dt, float(a.batch_size)
                H.cb_dps_next, H.cb_dV = ws.dps_next2.data_ptr(), ws.dVn.data_ptr()
            job = P.__dict__.get("cf_job") if use_head else None
            H.cb_defer = 1 if job else 0
            if job:
                H.cb_partials, H.cb_tiles, H.cb_stage = job[0], job[1], job[4]
                C.memmove(C.byref(H.cb_auglag), C.byref(job[2]), C.sizeof(_lib.AuglagArgs))
            call("nlbac_mlp_bwd_data_head", P.n_q5v, P.io_q5v, 2 * NP + 1, B, C.byref(H), s)
            ws.q5_bwd_done = True
        else:
            call("nlbac_unicycle_constraints_bwd", ws.ps_next2.data_ptr(), ws.matr.data_ptr(), ws.bmatr.data_ptr(),
                 self.hazards.data_ptr(), self.num_cbfs, dt, float(a.batch_size), B, sc, ws.dps_next2.data_ptr(),
                 ws.dVn.data_ptr(), s)
            call("nlbac_mlp_bwd_data", P.n_l, P.io_vn, 1, B, s)
        if mapped:
            du2, _ = self.solver.backward(None, need_du=True)
        else:
            call("nlbac_unicycle_lookahead_bwd", x_next2.data_ptr(), ws.dps_next2.data_ptr(), ws.dps_v2.data_ptr(), 2 * B,
                 self.l_p, ws.dx_next2.data_ptr(), s)
            du2, _ = self.solver.backward(ws.dx_next2, need_du=True)
        return du2, self.act_dim

    def _constraint_head_ok(self, nets, cnt, ws):
        """Single GPU, launch folds on, both controllers in the update, and the forward kernels that serve these nets
        evaluate constraint heads (``nlbac_mlp_fwd_head_ok``)."""
        a = self.agent
        if not (a.world == 1 and a.fold_launches and ws.np_now == 2 and self.num_cbfs == 7):
            return False
        ok = self.__dict__.get("_cf_ok")
        if ok is None:
            ok = self._cf_ok = bool(_lib.load().nlbac_mlp_fwd_head_ok(nets, cnt))
        return ok

    def first_step_done(self):
        return self.solver.first_step_done()

    # -- NODE fit (U/model.py:221-260 via U/sac_cbf_clf.py:205-219) -------------------------------
    def fit_inputs(self, rows):
        """(obs ptr, ld, action (N,2), next_obs ptr, ld, N) of minibatch-layout rows."""
        lay = self.agent.lay
        return (rows.data_ptr(), rows.shape[1], rows[:, lay.act:lay.act + 2], rows.data_ptr() + 4 * lay.nobs,
                rows.shape[1], rows.shape[0])

    def fit_ws(self, N):
        z = self.z
        return dict(st=z(N, 3), nst=z(N, 3), dpred=z(N, 3), part=z((N + 255) // 256), u=z(N, 2))

    def fit_part1(self, w, p_obs, obs_ld, p_nobs, nobs_ld, N):
        a, s = self.agent, stream_ptr()
        _lib.call("nlbac_unicycle_state", p_obs, obs_ld, N, self.l_p, w["st"].data_ptr(), 1, None, s)
        _lib.call("nlbac_unicycle_state", p_nobs, nobs_ld, N, self.l_p, w["nst"].data_ptr(), 1, None, s)
        if isinstance(self.fit_solver, AffineNodeSolver) and not isinstance(self.fit_solver, ConcatNodeSolver):
            self.reserve(self.fit_solver, N, 1)
        self.fit_solver.forward_begin(w["st"], w["u"], 1, N, a.solver, self.env.dt, a.atol, a.rtol)


# =====================================================================================
class UnicycleBarrierTask(UnicycleTask):
    """Learned-barrier-certificate Unicycle (NU/sac_cbf_clf/sac_cbf_clf.py:339-477): one controller; the CBF is a
    network B(obs, a) trained with the critics; its term needs the predicted next observation get_obs(x')
    (differentiable) and a re-sampled, detached next action; no ratio in the loss."""
    name = "UnicycleBarrier"
    n_pol, backup_mode, has_signal, n_extra_critics = 1, 0, True, 1
    ratio_mode = 0
    n_eps = 3                 # next-obs sample, obs sample, sample on the predicted next observation
    graph_ok = True
    GOAL = (2.5, 2.5)         # NU/sac_cbf_clf/dynamics.py:104-105

    def __init__(self, agent, env, args):
        super().__init__(agent, env, args)
        self.num_cbfs = 1

    def alloc(self, ws):
        B, z, H = ws.B, self.z, self.agent.hidden
        ws.y0 = z(B, 3)
        ws.V, ws.Vn, ws.dVn = z(B), z(B), z(B)
        ws.acts_vn = z(2, B, H)
        ws.ps_next, ws.dps_v = z(B, 2), z(B, 2)
        ws.Bv, ws.Bn, ws.dBn = z(B), z(B), z(B)
        ws.acts_bn = z(2, B, H)
        ws.obs_pred, ws.heads_nx, ws.pi_next, ws.logp_nx = z(B, 7), z(B, 4), z(B, 2), z(B)
        ws.dxb = z(B, 9)                               # d B(obs', a') / d [obs', a']
        ws.matr = z(B, 2)
        ws.part_c = z(ws.nblk, 2)
        ws.dx_next = z(B, 3)

    def extra_value_nets(self):
        return [self.agent.h_extra[0]]

    def extra_value_io(self, ws, io, i):               # B(obs, pi), value only (detached in the reference)
        lay = self.agent.lay
        io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.mb.data_ptr() + 4 * lay.obs, 7, lay.LD
        io[i].x1, io[i].x1_dim, io[i].x1_ld = ws.pi2.data_ptr(), 2, 2
        io[i].y, io[i].y_ld = ws.Bv.data_ptr(), 1

    def plan(self, ws, P):
        a = self.agent
        P.n_l = mlp_array([a.h_l.desc])
        io = P.io_vn = io_array(1)                     # V(p(x')) forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.ps_next.data_ptr(), 2, 2
        io[0].y, io[0].y_ld = ws.Vn.data_ptr(), 1
        io[0].acts = ws.acts_vn.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dVn.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dps_v.data_ptr(), 2
        P.n_pi = mlp_array([a.h_p.desc])
        io = P.io_nx = io_array(1)                     # policy on the predicted next observation
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 7, 7
        io[0].y, io[0].y_ld = ws.heads_nx.data_ptr(), 4
        P.n_bar = mlp_array([a.h_extra[0].desc])
        io = P.io_bn = io_array(1)                     # B(obs', a') forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 7, 7
        io[0].x1, io[0].x1_dim, io[0].x1_ld = ws.pi_next.data_ptr(), 2, 2
        io[0].y, io[0].y_ld = ws.Bn.data_ptr(), 1
        io[0].acts = ws.acts_bn.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dBn.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dxb.data_ptr(), 9

    def rollout_begin(self, ws, P):
        a, s = self.agent, stream_ptr()
        if self.solver.fused and a.fold_launches:       # (the state is formed by the rollout's first launch: UnicycleTask)
            self.solver.set_in_map(1, ws.mb, a.lay.LD, self.l_p, None)
        else:
            _lib.call("nlbac_unicycle_state", ws.mb.data_ptr(), a.lay.LD, ws.B, self.l_p, ws.y0.data_ptr(), 1, None, s)
        self.reserve(self.solver, ws.B, 1)
        self.solver._out_map = None
        self.solver.forward_begin(ws.y0, ws.pi2, 1, ws.B, a.solver, float(self.env.dt), a.atol, a.rtol)

    def loss_and_backward(self, ws, P, lam_upd, assume_single):
        a, s, call = self.agent, stream_ptr(), _lib.call
        B, sc, dt = ws.B, a.sc.data_ptr(), float(self.env.dt)
        pol = a.policy
        gx, gy = self.GOAL
        x_next = self.solver.forward_finish(assume_single_step=assume_single)
        a.drain_fill()
        call("nlbac_unicycle_lookahead", x_next.data_ptr(), B, self.l_p, ws.ps_next.data_ptr(), s)
        call("nlbac_mlp_fwd", P.n_l, P.io_vn, 1, B, s)
        call("nlbac_unicycle_obs_fwd", x_next.data_ptr(), B, gx, gy, ws.obs_pred.data_ptr(), 7, s)
        self.policy_sample(ws, "nx", P.n_pi, P.io_nx, 1, B, ws.heads_nx, ws.eps[2], 2, ws.pi_next, 2, ws.logp_nx)
        call("nlbac_mlp_fwd", P.n_bar, P.io_bn, 1, B, s)
        call("nlbac_barrier_constraints_fwd", ws.Bv.data_ptr(), ws.Bn.data_ptr(), ws.V.data_ptr(), ws.Vn.data_ptr(),
             dt, float(a.gamma_b), self.gamma_l, B, ws.matr.data_ptr(), ws.part_c.data_ptr(),
             *a.auglag_fused(ws, 1, lam_upd), s)
        a.auglag(ws, 1, lam_upd)
        call("nlbac_barrier_constraints_bwd", ws.matr.data_ptr(), dt, float(a.batch_size), B, sc, ws.dBn.data_ptr(),
             ws.dVn.data_ptr(), s)
        call("nlbac_mlp_bwd_data", P.n_l, P.io_vn, 1, B, s)          # dV' -> d p(x')
        call("nlbac_mlp_bwd_data", P.n_bar, P.io_bn, 1, B, s)        # dB' -> d [obs', a'] (a' is detached)
        call("nlbac_unicycle_lookahead_bwd", x_next.data_ptr(), ws.dps_v.data_ptr(), None, B, self.l_p,
             ws.dx_next.data_ptr(), s)
        call("nlbac_unicycle_obs_bwd", x_next.data_ptr(), ws.dxb.data_ptr(), 9, B, gx, gy, ws.dx_next.data_ptr(), 1, s)
        du, _ = self.solver.backward(ws.dx_next, need_du=True)
        return du, self.act_dim


# =====================================================================================
class CarsTask(_Task):
    """SimulatedCars: two-step rollout of a non-affine NODE on [x, u, t]; the second action is re-sampled from
    the (detached) predicted observation and carries no gradient; relative-degree-2 CBFs between cars 3-4 and
    4-5, CLF on (x3, v3, x4, v4)."""
    rollout_waits = 2
    name = "SimulatedCars"
    obs_dim, act_dim, lya_dim, n_s = 10, 1, 4, 10
    n_eps = 5
    lam_hi = 300.0
    ratio_mode = 2
    collision_radius = 4.5

    def __init__(self, agent, env, args):
        super().__init__(agent, env, args)
        self.num_cbfs = 2
        self.gamma_l = 0.15

    def build_node(self):
        return NeuralODEModel(12, 10)

    def setup(self):
        a = self.agent
        self.solver1 = ConcatNodeSolver(a.neural_ode_model, a.device)     # x_t   -> x_t+1 (2B rows)
        self.solver2 = ConcatNodeSolver(a.neural_ode_model, a.device)     # x_t+1 -> x_t+2
        self.solver1.keep_acts = self.solver2.keep_acts = False           # differentiated w.r.t. state / carried inputs only
        self.fit_solver = ConcatNodeSolver(a.neural_ode_model, a.device)
        self.solvers = [self.solver1, self.solver2, self.fit_solver]

    def alloc(self, ws):
        B, z, H = ws.B, self.z, self.agent.hidden
        ws.state = z(B, 10)
        ws.y0_2 = z(2 * B, 10)
        ws.c1, ws.c2 = z(2 * B, 2), z(2 * B, 2)        # carried [action, time] of the two steps
        ws.x1_2, ws.obs1_2 = z(2 * B, 10), z(2 * B, 10)
        ws.heads_nx, ws.logp_nx = z(2 * B, 2), z(2 * B)
        ws.V, ws.V1, ws.dV1 = z(B), z(B), z(B)
        ws.acts_v1 = z(2, B, H)
        ws.dlya = z(B, 4)
        ws.matr, ws.bmatr = z(B, 3), z(B, 2)
        ws.part_c = z(ws.nblk, 5)
        ws.dx1, ws.dx2 = z(2 * B, 10), z(2 * B, 10)
        ws.du2 = z(2 * B, 1)

    def plan(self, ws, P):
        a, B = self.agent, ws.B
        P.n_l = mlp_array([a.h_l.desc])
        io = P.io_v1 = io_array(1)                 # V(x_t+1[4:8]) forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.x1_2.data_ptr() + 4 * 4, 4, 10
        io[0].y, io[0].y_ld = ws.V1.data_ptr(), 1
        io[0].acts = ws.acts_v1.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dV1.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dlya.data_ptr(), 4
        io = P.io_nx = io_array(2)                 # both policies on the predicted next observation
        for i in range(2):
            io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.obs1_2[i * B:].data_ptr(), 10, 10
            io[i].y, io[i].y_ld = ws.heads_nx[i * B:].data_ptr(), 2

    def value_now_io(self, ws, io, i):
        lay = self.agent.lay
        io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.mb.data_ptr() + 4 * lay.lya, self.lya_dim, lay.LD
        io[i].y, io[i].y_ld = ws.V.data_ptr(), 1

    def rollout_begin(self, ws, P):
        a, s = self.agent, stream_ptr()
        B, lay = ws.B, a.lay
        # state, its primary / backup copies and the carried [action, time] inputs of both steps: one launch
        _lib.call("nlbac_cars_rollout_inputs", ws.mb.data_ptr(), lay.LD, lay.t, lay.nt, ws.pi2.data_ptr(), B,
                  ws.state.data_ptr(), ws.y0_2.data_ptr(), ws.c1.data_ptr(), ws.c2.data_ptr(), s)
        self.solver1.out_into = ws.x1_2 if a.fold_launches else None     # x_t+1 lands where the next launches read it
        self.solver1.forward_begin(ws.y0_2, ws.c1, 2, B, a.solver, float(self.env.dt), a.atol, a.rtol)

    def loss_and_backward(self, ws, P, lam_upd, assume_single):
        a, s, call = self.agent, stream_ptr(), _lib.call
        B, sc, dt = ws.B, a.sc.data_ptr(), float(self.env.dt)
        pol = a.policy
        x1 = self.solver1.forward_finish()
        if x1.data_ptr() != ws.x1_2.data_ptr():
            ws.x1_2.copy_(x1)         # (normally the solver has written there itself: out_into, rollout_begin)
        # u_(t+1) ~ pi(. | get_obs(x_t+1)), detached (C/sac_cbf_clf.py:441-451, 585-595)
        call("nlbac_cars_obs", ws.x1_2.data_ptr(), 2 * B, ws.obs1_2.data_ptr(), s)
        self.policy_sample(ws, "nx", P.n_act, P.io_nx, 2, B, ws.heads_nx, ws.eps[3:5], 1, ws.c2, 2, ws.logp_nx)
        x2 = self.solver2.forward(ws.x1_2, ws.c2, 2, B, a.solver, dt, a.atol, a.rtol)
        a.drain_fill()
        call("nlbac_mlp_fwd", P.n_l, P.io_v1, 1, B, s)
        call("nlbac_cars_constraints_fwd", ws.state.data_ptr(), ws.x1_2.data_ptr(), x2.data_ptr(), ws.V.data_ptr(),
             ws.V1.data_ptr(), float(a.gamma_b), self.gamma_l, self.collision_radius, B, ws.matr.data_ptr(),
             ws.bmatr.data_ptr(), ws.part_c.data_ptr(), *a.auglag_fused(ws, self.num_cbfs, lam_upd), s)
        a.auglag(ws, self.num_cbfs, lam_upd)
        call("nlbac_cars_constraints_bwd", ws.matr.data_ptr(), ws.bmatr.data_ptr(), float(a.gamma_b),
             float(a.batch_size), B, sc, ws.dx1.data_ptr(), ws.dx2.data_ptr(), ws.dV1.data_ptr(), s)
        call("nlbac_mlp_bwd_data", P.n_l, P.io_v1, 1, B, s)               # dV1 -> d x1[0:B, 4:8]
        # x_t+2 depends on the first action only through x_t+1
        _, dy0 = self.solver2.backward(ws.dx2, need_du=False, need_dy0=True)
        # d/dx_t+1: the constraints' own (dx1) + V(x_t+1)'s on columns 4..7 of the primary rows + the second solve's
        call("nlbac_add_cols_plus", ws.dx1.data_ptr(), 10, 4, ws.dlya.data_ptr(), 4, 4, B, dy0.data_ptr(), 2 * B, s)
        dc, _ = self.solver1.backward(ws.dx1, need_du=True)               # (2B, 2): d/d[action, time]
        return dc, 2

    def first_step_done(self):
        return self.solver1.first_step_done()

    # -- NODE fit (C/model.py:208-252 via C/sac_cbf_clf.py:201-217) --------------------------------
    def fit_inputs(self, rows):
        lay = self.agent.lay
        c = torch.stack((rows[:, lay.act], rows[:, lay.t]), 1)
        return (rows.data_ptr(), rows.shape[1], c, rows.data_ptr() + 4 * lay.nobs, rows.shape[1], rows.shape[0])

    def fit_ws(self, N):
        z = self.z
        return dict(st=z(N, 10), nst=z(N, 10), dpred=z(N, 10), part=z((N + 255) // 256), u=z(N, 2))

    def fit_part1(self, w, p_obs, obs_ld, p_nobs, nobs_ld, N):
        a, s = self.agent, stream_ptr()
        _lib.call("nlbac_cars_state", p_obs, obs_ld, N, w["st"].data_ptr(), s)
        _lib.call("nlbac_cars_state", p_nobs, nobs_ld, N, w["nst"].data_ptr(), s)
        if isinstance(self.fit_solver, AffineNodeSolver) and not isinstance(self.fit_solver, ConcatNodeSolver):
            self.reserve(self.fit_solver, N, 1)
        self.fit_solver.forward_begin(w["st"], w["u"], 1, N, a.solver, self.env.dt, a.atol, a.rtol)


# =====================================================================================
class PvtolTask(_Task):
    """Pvtol (P/sac_cbf_clf/sac_cbf_clf.py:376-1048): control-affine NODE on the six dynamic states, three-step
    rollout (the second and third actions are re-sampled from the predicted observations and detached), the safety
    operator's position follows the predicted x, relative-degree-3 CBFs (5 hazards, 2 operator distances, y_max,
    y_min) + CLF on the predicted observation.  The backup controller is trained every ``backup_update_interval``
    updates (P:282) with its own augmented term and Adam state."""
    rollout_waits = 3
    name = "Pvtol"
    obs_dim, act_dim, lya_dim, n_s = 11, 2, 11, 6
    n_eps = 7
    # reference draw order: next-obs, obs, pi_next, pi_next_next [, backup, backup pi_next, backup pi_next_next];
    # device slots keep each step's (primary, backup) draws adjacent
    eps_order = [0, 1, 4, 2, 5, 3, 6]
    lam_hi, ratio_mode, backup_mode = 400.0, 2, 2
    GOAL = (4.5, 4.5)

    def __init__(self, agent, env, args):
        super().__init__(agent, env, args)
        self.num_cbfs = len(env.hazard_locations) + 4
        self.gamma_l = 0.1
        self.backup_interval = int(getattr(args, "backup_update_interval", 20))
        agent.backup_update_interval = self.backup_interval

    def build_node(self):
        return NeuralODEModel(6, 6, 12)

    def n_pol_now(self, updates):
        return 2 if updates % self.backup_interval == 0 else 1

    def backup_lam_due(self, updates, interval):
        return 1 if updates % (interval * self.backup_interval) == 0 else 0

    def fit_due(self, i_episode):
        return i_episode is None or i_episode <= 100

    def lya_train_cols(self, lay):
        return lay.obs, lay.nobs          # P:243-252: the Lyapunov critic is regressed on observations

    def setup(self):
        a = self.agent
        self.hazards = torch.tensor(np.asarray(self.env.hazard_locations), dtype=torch.float32,
                                    device=a.device).contiguous()
        self.steps = [AffineNodeSolver(a.neural_ode_model, a.device) for _ in range(3)]
        for sv in self.steps:
            sv.keep_acts = False                                          # differentiated w.r.t. state / actions only
        self.fit_solver = AffineNodeSolver(a.neural_ode_model, a.device)
        self.solvers = self.steps + [self.fit_solver]

    def alloc(self, ws):
        B, z, H = ws.B, self.z, self.agent.hidden
        ws.st6, ws.op0 = z(B, 6), z(B)
        ws.y0 = z(2 * B, 6)
        ws.x1, ws.x2, ws.x3 = z(2 * B, 6), z(2 * B, 6), z(2 * B, 6)
        ws.obs1, ws.obs2 = z(2 * B, 11), z(2 * B, 11)
        ws.op1, ws.op2 = z(2 * B), z(2 * B)
        ws.heads_n1, ws.heads_n2 = z(2 * B, 4), z(2 * B, 4)
        ws.a1, ws.a2, ws.logp_nx = z(2 * B, 2), z(2 * B, 2), z(2 * B)
        ws.V, ws.V1, ws.dV1 = z(B), z(B), z(B)
        ws.acts_v1 = z(2, B, H)
        ws.dobs1 = z(B, 11)
        ws.matr, ws.bmatr = z(B, self.num_cbfs + 1), z(B, self.num_cbfs)
        ws.part_c = z(ws.nblk, 2 * self.num_cbfs + 1)
        ws.dx1, ws.dx2, ws.dx3 = z(2 * B, 6), z(2 * B, 6), z(2 * B, 6)

    def value_now_io(self, ws, io, i):
        lay = self.agent.lay
        io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.mb.data_ptr() + 4 * lay.lya, self.lya_dim, lay.LD
        io[i].y, io[i].y_ld = ws.V.data_ptr(), 1

    def plan(self, ws, P):
        a, B, NP = self.agent, ws.B, P.NP
        P.n_l = mlp_array([a.h_l.desc])
        io = P.io_v1 = io_array(1)                 # V(obs(x_t+1)) forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs1.data_ptr(), 11, 11
        io[0].y, io[0].y_ld = ws.V1.data_ptr(), 1
        io[0].acts = ws.acts_v1.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dV1.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dobs1.data_ptr(), 11
        P.n_pols = mlp_array([h.desc for h in a.h_pols[:NP]])
        P.io_nx = []
        for obs, heads in ((ws.obs1, ws.heads_n1), (ws.obs2, ws.heads_n2)):
            io = io_array(NP)                      # each controller on its own rows of the predicted observation
            for i in range(NP):
                io[i].x0, io[i].x0_dim, io[i].x0_ld = obs[i * B:].data_ptr(), 11, 11
                io[i].y, io[i].y_ld = heads[i * B:].data_ptr(), 4
            P.io_nx.append(io)

    def rollout_begin(self, ws, P):
        a, s = self.agent, stream_ptr()
        B, NP = ws.B, P.NP
        # the state, once per controller's rows of the rollout's initial state (st6 = the first block: no D2D copies —
        # a torch .copy_ is ~20 us of host time, more than the 5 us launch it replaces)
        ws.st6 = ws.y0[:B]
        for p in range(NP):
            _lib.call("nlbac_pvtol_state", ws.mb.data_ptr(), a.lay.LD, B, ws.y0[p * B:].data_ptr(),
                      ws.op0.data_ptr() if p == 0 else None, s)
        for sv in self.steps:
            self.reserve(sv, NP * B, NP)
        self.steps[0].forward_begin(ws.y0[:NP * B], ws.pi2[:NP * B], NP, B, a.solver, float(self.env.dt), a.atol,
                                    a.rtol)

    def loss_and_backward(self, ws, P, lam_upd, assume_single):
        a, s, call = self.agent, stream_ptr(), _lib.call
        B, NP, sc, dt, env = ws.B, P.NP, a.sc.data_ptr(), float(self.env.dt), self.env
        n = NP * B
        pol = a.policy
        p_scale, p_bias = pol.action_scale.data_ptr(), pol.action_bias.data_ptr()
        follow, (gx, gy) = float(env.safety_operator_follow), self.GOAL
        s1, s2, s3 = self.steps
        # (x_t+1 .. x_t+3 are read where the solvers left them — three solvers, three output buffers, none re-used before
        #  the update ends: no D2D copies between the chained solves)
        ws.x1 = s1.forward_finish()
        # u_(t+1), u_(t+2) ~ pi(. | get_obs(x)), detached (P:474-526)
        call("nlbac_pvtol_obs_fwd", ws.x1.data_ptr(), ws.op0.data_ptr(), B, follow, gx, gy, n, ws.obs1.data_ptr(), 11,
             ws.op1.data_ptr(), s)
        self.policy_sample(ws, ("n1", NP), P.n_pols, P.io_nx[0], NP, B, ws.heads_n1, ws.eps[3:3 + NP], 2, ws.a1, 2, ws.logp_nx)
        ws.x2 = s2.forward(ws.x1[:n], ws.a1[:n], NP, B, a.solver, dt, a.atol, a.rtol)
        call("nlbac_pvtol_obs_fwd", ws.x2.data_ptr(), ws.op1.data_ptr(), n, follow, gx, gy, n, ws.obs2.data_ptr(), 11,
             ws.op2.data_ptr(), s)
        self.policy_sample(ws, ("n2", NP), P.n_pols, P.io_nx[1], NP, B, ws.heads_n2, ws.eps[5:5 + NP], 2, ws.a2, 2, ws.logp_nx)
        ws.x3 = s3.forward(ws.x2[:n], ws.a2[:n], NP, B, a.solver, dt, a.atol, a.rtol)
        a.drain_fill()
        call("nlbac_mlp_fwd", P.n_l, P.io_v1, 1, B, s)
        hz = self.hazards.data_ptr()
        call("nlbac_pvtol_constraints_fwd", ws.st6.data_ptr(), ws.op0.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(),
             ws.x3.data_ptr(), ws.V.data_ptr(), ws.V1.data_ptr(), hz, len(env.hazard_locations),
             1.2 * float(env.hazards_radius), 0.9 * float(env.operator_dist), float(env.y_max), float(env.y_min),
             follow, float(a.gamma_b), self.gamma_l, B, NP, ws.matr.data_ptr(), ws.bmatr.data_ptr(),
             ws.part_c.data_ptr(), *a.auglag_fused(ws, self.num_cbfs, lam_upd), s)
        a.auglag(ws, self.num_cbfs, lam_upd)
        call("nlbac_pvtol_constraints_bwd", ws.matr.data_ptr(), ws.bmatr.data_ptr(), ws.x1.data_ptr(),
             ws.x2.data_ptr(), ws.x3.data_ptr(), hz, len(env.hazard_locations), follow, float(a.gamma_b),
             float(a.batch_size), B, NP, sc, ws.dx1.data_ptr(), ws.dx2.data_ptr(), ws.dx3.data_ptr(),
             ws.dV1.data_ptr(), s)
        call("nlbac_mlp_bwd_data", P.n_l, P.io_v1, 1, B, s)               # dV1 -> d obs(x_t+1) (primary rows)
        call("nlbac_pvtol_obs_bwd", ws.x1.data_ptr(), ws.dobs1.data_ptr(), 11, follow, gx, gy, B, ws.dx1.data_ptr(),
             1, s)
        # x_t+3 and x_t+2 depend on the first action only through the state handed from step to step
        _, dy0 = s3.backward(ws.dx3[:n], need_du=False, need_dy0=True)
        call("nlbac_axpby", 1.0, ws.dx2.data_ptr(), 1.0, dy0.data_ptr(), n * 6, ws.dx2.data_ptr(), s)
        _, dy0 = s2.backward(ws.dx2[:n], need_du=False, need_dy0=True)
        call("nlbac_axpby", 1.0, ws.dx1.data_ptr(), 1.0, dy0.data_ptr(), n * 6, ws.dx1.data_ptr(), s)
        du, _ = s1.backward(ws.dx1[:n], need_du=True)
        return du, self.act_dim

    def first_step_done(self):
        return self.steps[0].first_step_done()

    # -- NODE fit (P/model.py:224-266 via P/sac_cbf_clf.py:205-219) --------------------------------
    def fit_inputs(self, rows):
        lay = self.agent.lay
        return (rows.data_ptr(), rows.shape[1], rows[:, lay.act:lay.act + 2], rows.data_ptr() + 4 * lay.nobs,
                rows.shape[1], rows.shape[0])

    def fit_ws(self, N):
        z = self.z
        return dict(st=z(N, 6), nst=z(N, 6), dpred=z(N, 6), part=z((N + 255) // 256), u=z(N, 2))

    def fit_part1(self, w, p_obs, obs_ld, p_nobs, nobs_ld, N):
        a, s = self.agent, stream_ptr()
        _lib.call("nlbac_pvtol_state", p_obs, obs_ld, N, w["st"].data_ptr(), None, s)
        _lib.call("nlbac_pvtol_state", p_nobs, nobs_ld, N, w["nst"].data_ptr(), None, s)
        if isinstance(self.fit_solver, AffineNodeSolver) and not isinstance(self.fit_solver, ConcatNodeSolver):
            self.reserve(self.fit_solver, N, 1)
        self.fit_solver.forward_begin(w["st"], w["u"], 1, N, a.solver, self.env.dt, a.atol, a.rtol)


# =====================================================================================
class PvtolBarrierTask(PvtolTask):
    """Learned-barrier-certificate Pvtol (NP/sac_cbf_clf/sac_cbf_clf.py:334-480): one controller, one NODE step, the
    learned CBF term on get_obs(x') with a re-sampled detached next action, CLF (V' - V)/1 + 0.1 V on the predicted
    observation, ratio clamped at 0.002."""
    rollout_waits = 1
    name = "PvtolBarrier"
    n_pol, backup_mode, has_signal, n_extra_critics = 1, 0, True, 1
    n_eps, eps_order = 3, None
    lam_hi, ratio_mode = 400.0, 2

    def __init__(self, agent, env, args):
        _Task.__init__(self, agent, env, args)
        self.num_cbfs = 1
        self.gamma_l = 0.1
        self.backup_interval = 1

    def n_pol_now(self, updates):
        return 1

    def backup_lam_due(self, updates, interval):
        return 0

    def setup(self):
        a = self.agent
        self.solver = AffineNodeSolver(a.neural_ode_model, a.device)
        self.solver.keep_acts = False
        self.fit_solver = AffineNodeSolver(a.neural_ode_model, a.device)
        self.solvers = [self.solver, self.fit_solver]

    def alloc(self, ws):
        B, z, H = ws.B, self.z, self.agent.hidden
        ws.st6, ws.op0 = z(B, 6), z(B)
        ws.V, ws.V1, ws.dV1 = z(B), z(B), z(B)
        ws.acts_v1 = z(2, B, H)
        ws.obs_pred, ws.dobs1 = z(B, 11), z(B, 11)
        ws.heads_nx, ws.pi_next, ws.logp_nx = z(B, 4), z(B, 2), z(B)
        ws.Bv, ws.Bn, ws.dBn = z(B), z(B), z(B)
        ws.acts_bn = z(2, B, H)
        ws.dxb = z(B, 13)                              # d B(obs', a') / d [obs', a']
        ws.matr = z(B, 2)
        ws.part_c = z(ws.nblk, 2)
        ws.dx_next = z(B, 6)

    def extra_value_nets(self):
        return [self.agent.h_extra[0]]

    def extra_value_io(self, ws, io, i):               # B(obs, pi), value only (detached in the reference)
        lay = self.agent.lay
        io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.mb.data_ptr() + 4 * lay.obs, 11, lay.LD
        io[i].x1, io[i].x1_dim, io[i].x1_ld = ws.pi2.data_ptr(), 2, 2
        io[i].y, io[i].y_ld = ws.Bv.data_ptr(), 1

    def plan(self, ws, P):
        a = self.agent
        P.n_l = mlp_array([a.h_l.desc])
        io = P.io_v1 = io_array(1)                     # V(obs(x')) forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 11, 11
        io[0].y, io[0].y_ld = ws.V1.data_ptr(), 1
        io[0].acts = ws.acts_v1.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dV1.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dobs1.data_ptr(), 11
        P.n_pi = mlp_array([a.h_p.desc])
        io = P.io_nx = io_array(1)                     # policy on the predicted next observation
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 11, 11
        io[0].y, io[0].y_ld = ws.heads_nx.data_ptr(), 4
        P.n_bar = mlp_array([a.h_extra[0].desc])
        io = P.io_bn = io_array(1)                     # B(obs', a') forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 11, 11
        io[0].x1, io[0].x1_dim, io[0].x1_ld = ws.pi_next.data_ptr(), 2, 2
        io[0].y, io[0].y_ld = ws.Bn.data_ptr(), 1
        io[0].acts = ws.acts_bn.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dBn.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dxb.data_ptr(), 13

    def rollout_begin(self, ws, P):
        a, s = self.agent, stream_ptr()
        _lib.call("nlbac_pvtol_state", ws.mb.data_ptr(), a.lay.LD, ws.B, ws.st6.data_ptr(), ws.op0.data_ptr(), s)
        self.reserve(self.solver, ws.B, 1)
        self.solver.forward_begin(ws.st6, ws.pi2, 1, ws.B, a.solver, float(self.env.dt), a.atol, a.rtol)

    def loss_and_backward(self, ws, P, lam_upd, assume_single):
        a, s, call = self.agent, stream_ptr(), _lib.call
        B, sc, env = ws.B, a.sc.data_ptr(), self.env
        pol = a.policy
        follow, (gx, gy) = float(env.safety_operator_follow), self.GOAL
        x1 = self.solver.forward_finish(assume_single_step=assume_single)
        a.drain_fill()
        call("nlbac_pvtol_obs_fwd", x1.data_ptr(), ws.op0.data_ptr(), B, follow, gx, gy, B, ws.obs_pred.data_ptr(), 11,
             None, s)
        call("nlbac_mlp_fwd", P.n_l, P.io_v1, 1, B, s)
        self.policy_sample(ws, "nx", P.n_pi, P.io_nx, 1, B, ws.heads_nx, ws.eps[2], 2, ws.pi_next, 2, ws.logp_nx)
        call("nlbac_mlp_fwd", P.n_bar, P.io_bn, 1, B, s)
        call("nlbac_barrier_constraints_fwd", ws.Bv.data_ptr(), ws.Bn.data_ptr(), ws.V.data_ptr(), ws.V1.data_ptr(),
             1.0, float(a.gamma_b), self.gamma_l, B, ws.matr.data_ptr(), ws.part_c.data_ptr(),
             *a.auglag_fused(ws, 1, lam_upd), s)
        a.auglag(ws, 1, lam_upd)
        call("nlbac_barrier_constraints_bwd", ws.matr.data_ptr(), 1.0, float(a.batch_size), B, sc, ws.dBn.data_ptr(),
             ws.dV1.data_ptr(), s)
        call("nlbac_mlp_bwd_data", P.n_l, P.io_v1, 1, B, s)          # dV' -> d obs'
        call("nlbac_mlp_bwd_data", P.n_bar, P.io_bn, 1, B, s)        # dB' -> d [obs', a'] (a' is detached)
        call("nlbac_pvtol_obs_bwd", x1.data_ptr(), ws.dobs1.data_ptr(), 11, follow, gx, gy, B, ws.dx_next.data_ptr(), 0, s)
        call("nlbac_pvtol_obs_bwd", x1.data_ptr(), ws.dxb.data_ptr(), 13, follow, gx, gy, B, ws.dx_next.data_ptr(), 1, s)
        du, _ = self.solver.backward(ws.dx_next, need_du=True)
        return du, self.act_dim

    def first_step_done(self):
        return self.solver.first_step_done()

# =====================================================================================
class QuadrotorBarrierTask(PvtolBarrierTask):
    """BASELINE configs[4] "Quadrotor + neural barrier certificate" as far as the reference describes it
    (/root/reference/README.md:66-72, 190-192; its code is an empty submodule — NO REFERENCE PARITY, checked against
    the oracle only): the learned-barrier agent pattern of NP (one controller, BarrierNetwork trained on the barrier
    signal D1 = -1 / D2 = -10, one NODE step, CLF (V' - V)/1 + 0.1 V, ratio clamped at 0.002) on a NON-affine
    single-net NODE  dx/dt = out_mu + out_sig * net(([x | u] - in_mu) / in_sig)  (8 -> 6, inputs normalised, outputs
    de-normalised inside the fused RK kernels).  The observation is the state, so get_state / get_obs are identities."""
    name = "QuadrotorBarrier"
    obs_dim, act_dim, lya_dim, n_s = 6, 2, 6, 6
    NODE_HIDDEN = 128

    def build_node(self):
        return NeuralODEModel(8, 6, hidden_dim=self.NODE_HIDDEN, normalizer=self.env.node_normalizer)

    def setup(self):
        a = self.agent
        self.solver = ConcatNodeSolver(a.neural_ode_model, a.device)       # carried inputs = the action
        self.solver.keep_acts = False                                      # differentiated w.r.t. the action only
        self.fit_solver = ConcatNodeSolver(a.neural_ode_model, a.device)
        self.solvers = [self.solver, self.fit_solver]

    def alloc(self, ws):
        B, z, H = ws.B, self.z, self.agent.hidden
        ws.st6 = z(B, 6)
        ws.V, ws.V1, ws.dV1 = z(B), z(B), z(B)
        ws.acts_v1 = z(2, B, H)
        ws.obs_pred = z(B, 6)
        ws.heads_nx, ws.pi_next, ws.logp_nx = z(B, 4), z(B, 2), z(B)
        ws.Bv, ws.Bn, ws.dBn = z(B), z(B), z(B)
        ws.acts_bn = z(2, B, H)
        ws.dxb = z(B, 8)                               # d B(obs', a') / d [obs', a']
        ws.matr = z(B, 2)
        ws.part_c = z(ws.nblk, 2)
        ws.dx_next = z(B, 6)

    def extra_value_io(self, ws, io, i):               # B(obs, pi), value only
        lay = self.agent.lay
        io[i].x0, io[i].x0_dim, io[i].x0_ld = ws.mb.data_ptr() + 4 * lay.obs, 6, lay.LD
        io[i].x1, io[i].x1_dim, io[i].x1_ld = ws.pi2.data_ptr(), 2, 2
        io[i].y, io[i].y_ld = ws.Bv.data_ptr(), 1

    def plan(self, ws, P):
        a = self.agent
        P.n_l = mlp_array([a.h_l.desc])
        io = P.io_v1 = io_array(1)                     # V(obs') forward + data backward, d obs' lands in dx_next
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 6, 6
        io[0].y, io[0].y_ld = ws.V1.data_ptr(), 1
        io[0].acts = ws.acts_v1.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dV1.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dx_next.data_ptr(), 6
        P.n_pi = mlp_array([a.h_p.desc])
        io = P.io_nx = io_array(1)                     # policy on the predicted next observation
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 6, 6
        io[0].y, io[0].y_ld = ws.heads_nx.data_ptr(), 4
        P.n_bar = mlp_array([a.h_extra[0].desc])
        io = P.io_bn = io_array(1)                     # B(obs', a') forward + data backward
        io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.obs_pred.data_ptr(), 6, 6
        io[0].x1, io[0].x1_dim, io[0].x1_ld = ws.pi_next.data_ptr(), 2, 2
        io[0].y, io[0].y_ld = ws.Bn.data_ptr(), 1
        io[0].acts = ws.acts_bn.data_ptr()
        io[0].dy, io[0].dy_ld = ws.dBn.data_ptr(), 1
        io[0].dx, io[0].dx_ld = ws.dxb.data_ptr(), 8

    def _obs_cols(self, src, src_ld, n, dst):
        """the state IS the observation: its six columns of a minibatch-layout row block, made contiguous"""
        _lib.call("nlbac_copy_blocks", src, src_ld, dst.data_ptr(), 6, 6, n, stream_ptr())

    def rollout_begin(self, ws, P):
        a = self.agent
        self._obs_cols(ws.mb.data_ptr() + 4 * a.lay.obs, a.lay.LD, ws.B, ws.st6)
        self.solver.forward_begin(ws.st6, ws.pi2, 1, ws.B, a.solver, float(self.env.dt), a.atol, a.rtol)

    def loss_and_backward(self, ws, P, lam_upd, assume_single):
        a, s, call = self.agent, stream_ptr(), _lib.call
        B, sc = ws.B, a.sc.data_ptr()
        pol = a.policy
        x1 = self.solver.forward_finish(assume_single_step=assume_single)
        a.drain_fill()
        call("nlbac_copy_blocks", x1.data_ptr(), 6 * B, ws.obs_pred.data_ptr(), 6 * B, 6 * B, 1, s)   # obs' = x'
        call("nlbac_mlp_fwd", P.n_l, P.io_v1, 1, B, s)
        self.policy_sample(ws, "nx", P.n_pi, P.io_nx, 1, B, ws.heads_nx, ws.eps[2], 2, ws.pi_next, 2, ws.logp_nx)
        call("nlbac_mlp_fwd", P.n_bar, P.io_bn, 1, B, s)
        call("nlbac_barrier_constraints_fwd", ws.Bv.data_ptr(), ws.Bn.data_ptr(), ws.V.data_ptr(), ws.V1.data_ptr(),
             1.0, float(a.gamma_b), self.gamma_l, B, ws.matr.data_ptr(), ws.part_c.data_ptr(),
             *a.auglag_fused(ws, 1, lam_upd), s)
        a.auglag(ws, 1, lam_upd)
        call("nlbac_barrier_constraints_bwd", ws.matr.data_ptr(), 1.0, float(a.batch_size), B, sc, ws.dBn.data_ptr(),
             ws.dV1.data_ptr(), s)
        call("nlbac_mlp_bwd_data", P.n_l, P.io_v1, 1, B, s)          # dV' -> d obs' (written to dx_next)
        call("nlbac_mlp_bwd_data", P.n_bar, P.io_bn, 1, B, s)        # dB' -> d [obs', a'] (a' is detached)
        call("nlbac_add_cols", ws.dx_next.data_ptr(), 6, 0, ws.dxb.data_ptr(), 8, 6, B, s)
        dc, _ = self.solver.backward(ws.dx_next, need_du=True)       # (B, 2): d / d action (the carried inputs)
        return dc, self.act_dim

    # -- NODE fit: one solve from (obs, action) against next_obs ---------------------------------------
    def fit_inputs(self, rows):
        lay = self.agent.lay
        return (rows.data_ptr() + 4 * lay.obs, rows.shape[1], rows[:, lay.act:lay.act + 2],
                rows.data_ptr() + 4 * lay.nobs, rows.shape[1], rows.shape[0])

    def fit_part1(self, w, p_obs, obs_ld, p_nobs, nobs_ld, N):
        a = self.agent
        self._obs_cols(p_obs, obs_ld, N, w["st"])
        self._obs_cols(p_nobs, nobs_ld, N, w["nst"])
        self.fit_solver.forward_begin(w["st"], w["u"], 1, N, a.solver, self.env.dt, a.atol, a.rtol)


TASKS = {"Unicycle": UnicycleTask, "SimulatedCars": CarsTask, "UnicycleBarrier": UnicycleBarrierTask,
         "Pvtol": PvtolTask, "PvtolBarrier": PvtolBarrierTask, "QuadrotorBarrier": QuadrotorBarrierTask}
