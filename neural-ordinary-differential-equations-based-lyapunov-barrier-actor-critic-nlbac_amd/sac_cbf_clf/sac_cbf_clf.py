"""``SAC_CBF_CLF`` — drop-in for the reference agent class (``U/sac_cbf_clf/sac_cbf_clf.py:26-656`` and its
per-environment copies ``C/``, ``P/``; the learned-barrier copies ``NU/``, ``NP/`` are the subclass in
``nlbac_amd/neural_barrier_certificate/``) whose ``update_parameters`` runs entirely on one MI355X through the C ABI
in ``include/nlbac_hip.h``.

Same constructor arguments, public methods, return values, checkpoint files and ``state_dict`` key names as the
reference.  Differences that are visible to a caller:
  * CUDA(HIP)-only: ``args.cuda`` must be true (no CPU fallback);
  * ``self.solver`` may be ``'euler'`` (reference default), ``'rk4'`` or ``'dopri5'``; the NODE-fit and CBF/CLF
    rollouts use it;
  * Lagrange multipliers / augmented terms / temperatures live on the device (``lambda_values`` etc. are read-back
    properties);
  * policy noise comes from the device generator unless ``set_noise`` is given pre-drawn N(0,1) samples (parity tests);
  * a replay object with ``sample_rows`` (``DeviceReplayMemory``) is gathered on the device.

The environment-specific half of an update (row layout, NODE form, rollout, CBF / CLF terms, NODE fit) is a task
object (``tasks.py``) chosen from ``env.dynamics_mode``; this file holds the shared SAC / Lyapunov machinery:
``_upd_part1`` (targets, critic step, actor forward, rollout start) and ``_upd_part2`` (constraints, actor backward,
actor step), see DESIGN.md §3.  Host<->device traffic per update: the minibatch upload (or 8 bytes per row of
indices), one 512-byte scalars read-back and, for dopri5, one 256-byte control block per attempted step.
"""
import collections
import ctypes as C
import os
import random
import types

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..odeint import HOST_COPY
from ..arena import Arena, bwd_weights, io_array, mlp_array, pack, skinny_partials_ws, stream_ptr
from . import _layout as SC
from .model import BarrierNetwork, GaussianPolicy, LyaNetwork, QNetwork
from .tasks import TASKS
from .utils import to_tensor

DYNAMICS_MODE = {'Unicycle': {'n_s': 3, 'n_u': 2}, 'SimulatedCars': {'n_s': 10, 'n_u': 1},
                 'Pvtol': {'n_s': 6, 'n_u': 2}, 'Quadrotor': {'n_s': 6, 'n_u': 2}}
l_p = 0.03


class PoseLoss(nn.Module):
    """MSE('mean') — kept for API parity; the fit computes it on device."""

    def forward(self, predicted_state, true_state):
        return nn.functional.mse_loss(predicted_state, true_state)


class _Layout:
    """Column offsets of one minibatch row in HBM: the fields of ``ReplayMemory.sample``
    (replay_memory.py:24-25) side by side, row stride padded to 16 bytes."""

    def __init__(self, task):
        self.obs_dim, self.act_dim, self.lya_dim = task.obs_dim, task.act_dim, task.lya_dim
        c = 0
        fields = [("obs", task.obs_dim), ("act", task.act_dim), ("rew", 1), ("con", 1), ("lya", task.lya_dim),
                  ("nlya", task.lya_dim), ("nobs", task.obs_dim), ("mask", 1), ("t", 1), ("nt", 1)]
        if task.has_signal:          # learned-barrier copies store a barrier signal after the constraint
            fields.insert(4, ("sig", 1))
        self.sig = None
        for name, w in fields:
            setattr(self, name, c)
            c += w
        self.width = c
        self.LD = (c + 3) // 4 * 4


class _Workspace:
    """Per-batch-size device buffers (allocated once, reused every update); the task adds its own."""

    def __init__(self, B, H, dev, lay, task):
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        A, Do = lay.act_dim, lay.obs_dim
        NP, NX = task.n_pol, task.n_extra_critics      # controllers; critic-type nets beyond Q1, Q2, L
        self.B = B
        self.mb = z(B, lay.LD)                   # minibatch rows (see _Layout)
        self.eps = z(task.n_eps, B, A)
        # policy samples of one update live side by side: rows [0,B) pi(s'), then pi(s) (and pi_backup(s)), so that
        # one forward launch and one sampling launch serve all of them
        self.heads3, self.act3, self.logp3 = z((1 + NP) * B, 2 * A), z((1 + NP) * B, A), z((1 + NP) * B)
        self.heads_n, self.na, self.nlogp = self.heads3[:B], self.act3[:B], self.logp3[:B]
        self.q6 = z(6 + 2 * NX, B)               # q1t q2t lt q1 q2 lf [xt x]...
        self.dq3 = z(3 + NX, B)
        self.next_q, self.next_l = z(B), z(B)
        self.acts_c = z(3 + NX, 2, B, H)         # Q1,Q2,L[,extras] saved activations
        self.dz_c = z(3 + NX, 2, B, H)
        self.nblk = (B + 255) // 256
        self.part_td = z(self.nblk, 3)
        self.n_tiles = (B + _lib.MLP_TILE_MIN - 1) // _lib.MLP_TILE_MIN
        self.part_td32 = z(self.n_tiles, 4)            # per-tile sums of the fused dy heads (nlbac_dy_head; 16-row tiles at most)
        # tickets of the heads' two-level elections: 1 + ceil(workgroups / 16) words each (TD head: <= 4 nets; actor head)
        self.tickets_td = torch.zeros(2 + self.n_tiles * 4 // 16 + 1, dtype=torch.int32, device=dev)
        self.tickets_q = torch.zeros(2 + self.n_tiles * NP // 16 + 1, dtype=torch.int32, device=dev)
        self.sums_tiles = torch.zeros(4, dtype=torch.int32, device=dev)    # nlbac_dy_head::sums_tiles of the td / actor-q heads
        self.sc_stage = z(SC.SC_SIZE)                                        # nlbac_dy_head::cb_stage
        self.part_tdx = z(max(NX, 1), self.nblk)
        self.heads2, self.pi2, self.logp2 = self.heads3[B:], self.act3[B:], self.logp3[B:]
        self.acts_p = z(NP, 2, B, H)
        self.dz_p = z(NP, 2, B, H)
        self.plan = None
        self.graphs, self.warm = {}, 0
        self.qpi = z(2, NP * B)
        self.acts_q = z(2 * NP, 2, B, H)
        self.dq_pi = z(2, NP * B)
        self.part_q = z(NP, self.nblk, 2)
        self.part_q32 = z(NP, self.n_tiles, 2)
        self.dxq = z(2, NP * B, Do + A)
        self.dheads2 = z(NP * B, 2 * A)
        task.alloc(self)


class SAC_CBF_CLF(object):
    variant = ""         # "Barrier" in the learned-barrier-certificate copies (neural_barrier_certificate/)

    def __init__(self, num_inputs, action_space, env, args):
        self.gamma = args.gamma
        self.gamma_b = args.gamma_b
        self.tau = args.tau
        self.policy_type = args.policy
        self.batch_size = args.batch_size
        self.target_update_interval = args.target_update_interval
        self.Lagrangian_multiplier_update_interval = args.Lagrangian_multiplier_update_interval
        self.automatic_entropy_tuning = args.automatic_entropy_tuning
        self.action_space = action_space
        self.action_space.seed(args.seed)
        if not args.cuda or not torch.cuda.is_available():
            raise RuntimeError("nlbac_amd.SAC_CBF_CLF runs on an MI355X only (pass --cuda on a GPU box); "
                               "there is no CPU fallback")
        if self.policy_type != "Gaussian":
            raise NotImplementedError("only the Gaussian policy is on the device path")
        if env.dynamics_mode + self.variant not in TASKS:
            raise Exception('Dynamics mode not supported.')
        _lib.load()
        self.device = torch.device("cuda")
        self.env = env
        self.task = task = TASKS[env.dynamics_mode + self.variant](self, env, args)
        self.lay = _Layout(task)
        if num_inputs != task.obs_dim or action_space.shape[0] != task.act_dim:
            raise ValueError("%s expects %d observations / %d actions" % (task.name, task.obs_dim, task.act_dim))
        self.center_pos_num = task.lya_dim       # inputs of the Lyapunov network
        self.critic_lyapunov_lr = 0.0004
        self.lr = args.lr
        hidden = args.hidden_size
        n_act = action_space.shape[0]
        self.num_inputs, self.n_act, self.hidden = num_inputs, n_act, hidden
        # gradient slabs of the actor / critic arenas (row ranges whose weight-gradient partials the optimiser sums).  8 for
        # the usual batches; 16 from batch 16384 on: a slab is then still >= 1024 rows and mlp_bwd_wide has twice the
        # workgroups (measured at B = 32768: 184 -> 159 us per launch; 128 x 128 output tiles instead — half the operand
        # re-reads — were slower than that, 170 us, and were dropped)
        self.n_grad_slabs = int(getattr(args, "grad_slabs", 16 if int(getattr(args, "batch_size", 0) or 0) >= 16384 else 8))
        if os.environ.get("NLBAC_GRAD_SLABS"):           # (experiments)
            self.n_grad_slabs = int(os.environ["NLBAC_GRAD_SLABS"])

        # --- same construction order (and RNG consumption) as the reference ---
        self.critic = QNetwork(num_inputs, n_act, hidden)
        self.lyapunovNet = LyaNetwork(self.center_pos_num, hidden)
        self.BarrierNet = BarrierNetwork(num_inputs, n_act, hidden) if task.has_signal else None
        QNetwork(num_inputs, n_act, hidden)          # the reference builds target nets here
        LyaNetwork(self.center_pos_num, hidden)      # (then hard-copies): consume the same RNG
        if task.has_signal:
            BarrierNetwork(num_inputs, n_act, hidden)
        self.cost_limit = 0.0
        self.augmented_ratio = 1.0005
        if args.seed >= 0:
            env.seed(args.seed)
            random.seed(args.seed)
            env.action_space.seed(args.seed)
            torch.manual_seed(args.seed)
            np.random.seed(args.seed)
            torch.cuda.manual_seed_all(args.seed)
        self.target_entropy = -float(np.prod(action_space.shape))
        self.log_alpha = nn.Parameter(torch.zeros(1))
        self.backup_log_alpha = nn.Parameter(torch.zeros(1))
        self.policy = GaussianPolicy(num_inputs, n_act, hidden, action_space)
        self.backup_policy = GaussianPolicy(num_inputs, n_act, hidden, action_space) if task.n_pol == 2 else None

        self.num_cbfs = task.num_cbfs
        self.l_p = l_p
        self.action_dim = env.action_space.shape[0]
        self.u_min, self.u_max = self.get_control_bounds()
        self.num_constraints = self.num_cbfs + 1
        self.neural_ode_model = task.build_node()
        self.solver = 'euler'
        self.model_loss_func = PoseLoss()
        self.atol, self.rtol = 1e-7, 1e-5

        # --- flat HBM arenas, one per optimiser group --------------------------
        dev = self.device
        self.ar_c = Arena(dev, self.n_grad_slabs, with_target=True)     # critic + Lyapunov (lr 4e-4)
        self.ar_a = Arena(dev, self.n_grad_slabs)                        # policies + log alphas (lr args.lr)
        self.n_fit_slabs = int(getattr(args, "fit_grad_slabs", 51))      # per accepted RK step of a NODE fit (row slabs = workgroups;
        # f_net + g_net have five hid x hid layers: 5 x 51 = 255 long workgroups for 256 CUs, mlp_dw16_kernels.hip)
        self.ar_n = Arena(dev, self.n_fit_slabs * 2)                     # NODE (lr 1e-3)
        self.h_q1, self.h_q2 = self.critic.attach(self.ar_c)
        (self.h_l,) = self.lyapunovNet.attach(self.ar_c)
        self.h_extra = list(self.BarrierNet.attach(self.ar_c)) if task.has_signal else []
        (self.h_p,) = self.policy.attach(self.ar_a)
        self.h_pols = [self.h_p]
        # a backup controller stepped on its own schedule (Pvtol: every 20th update) needs its own Adam state
        self.ar_b = Arena(dev, self.n_grad_slabs) if (task.n_pol == 2 and task.backup_interval > 1) else None
        ar_backup = self.ar_b if self.ar_b is not None else self.ar_a
        if task.n_pol == 2:
            (self.h_b,) = self.backup_policy.attach(ar_backup)
            self.h_pols.append(self.h_b)
        self.ar_a.add_group([self.log_alpha])
        if task.n_pol == 2:
            ar_backup.add_group([self.backup_log_alpha])
        self.h_node = list(self.neural_ode_model.attach(self.ar_n))
        self.arenas = [a for a in (self.ar_c, self.ar_a, self.ar_b, self.ar_n) if a is not None]
        for ar in self.arenas:
            ar.finalize()
        self.ar_c.hard_update_target()
        self.policy.to(dev)
        if self.backup_policy is not None:
            self.backup_policy.to(dev)
        self.h_crit = [self.h_q1, self.h_q2, self.h_l] + self.h_extra       # one Adam group, lr 4e-4
        for h in self.h_crit + self.h_pols + self.h_node:
            h.bind()
        la_off = self.ar_a.offset_of[id(self.log_alpha)]
        G = types.SimpleNamespace
        if self.ar_b is not None:       # two Adam groups: [policy, log_alpha] and [backup policy, backup log_alpha]
            self.actor_groups = [G(arena=self.ar_a, first=0, count=1, la_off=la_off, la_stride=0),
                                 G(arena=self.ar_b, first=1, count=1, la_stride=0,
                                   la_off=self.ar_b.offset_of[id(self.backup_log_alpha)])]
        else:
            stride = (self.ar_a.offset_of[id(self.backup_log_alpha)] - la_off) if task.n_pol == 2 else 0
            self.actor_groups = [G(arena=self.ar_a, first=0, count=task.n_pol, la_off=la_off, la_stride=stride)]
        self.pol_arena = [self.ar_a] + ([ar_backup] if task.n_pol == 2 else [])
        # target networks as modules over the Polyak buffer (state_dict / inspection)
        self.critic_target = _TargetView(self.critic, self.ar_c)
        self.lyapunovNet_target = _TargetView(self.lyapunovNet, self.ar_c)
        if task.has_signal:
            self.BarrierNet_target = _TargetView(self.BarrierNet, self.ar_c)
        self.repack_all()
        for ar in self.arenas:          # (built here, not inside a captured update)
            ar.scatter_tables()

        # --- device scalars: alpha, lambdas, augmented term --------------------
        self.sc = torch.zeros(SC.SC_SIZE, dtype=torch.float32, device=dev)
        sc_host = np.zeros(SC.SC_SIZE, dtype=np.float32)
        sc_host[SC.SC_ALPHA] = sc_host[SC.SC_BALPHA] = args.alpha
        sc_host[SC.SC_RHO_F64:SC.SC_RHO_F64 + 2] = np.array([1.0], dtype=np.float64).view(np.float32)
        sc_host[SC.SC_BRHO_F64:SC.SC_BRHO_F64 + 2] = np.array([1.0], dtype=np.float64).view(np.float32)
        self.sc.copy_(torch.from_numpy(sc_host))
        task.setup()
        self._tickets = torch.zeros(16, dtype=torch.int32, device=dev)     # last-workgroup tickets of the fused launches
        self._ws = {}
        self._noise = None
        self._fit_ws = {}
        self._fill = collections.deque()
        for sv in self.task.solvers:          # independent launches go in just before a solver waits for a decision
            sv.before_wait = self._fill_one
        self.use_graphs = False  # replay the update as hipGraphs (single GPU; see update_on_device)
        # per-row steps evaluated inside the MLP / solver launches that produce or consume them (nlbac_gauss_head,
        # nlbac_dy_head, nlbac_in_map / nlbac_out_map, results written to pinned memory by the kernels): False (or
        # NLBAC_FOLD=0) runs every step as the launch of its own it was — same numbers, for A/B runs and the tests
        self.fold_launches = os.environ.get("NLBAC_FOLD", "1") != "0"
        self.adjoint = bool(getattr(args, "adjoint", False))
        self.dp = None          # nlbac_amd.parallel.DataParallel when sharded over GPUs
        self._xb = {}

    @property
    def adjoint(self):
        """True: every NODE solve of the update is differentiated by the continuous adjoint (``odeint_adjoint``,
        BASELINE configs[3]) — the rollouts without a parameter adjoint (the policy loss needs d/d action only), the
        NODE fit with it — instead of back-propagating through the solver's steps."""
        return self._adjoint

    @adjoint.setter
    def adjoint(self, on):
        self._adjoint = bool(on)
        for sv in self.task.solvers:
            sv.adjoint = bool(on)

    def _graphs_on(self):
        # (a dopri5 adjoint solve reads its control block on the host: not capturable)
        return self.use_graphs and self.world == 1 and self.task.graph_ok and not (self._adjoint and self.solver == "dopri5")

    @property
    def node_solver(self):
        return self.task.solvers[0]

    @property
    def fit_solver(self):
        return self.task.fit_solver

    # ------------------------------------------------------------ data parallel
    def enable_data_parallel(self, dist, group=None, always_collective=False, step_control="global"):
        """Shard minibatches over the ranks of ``dist`` (one process per GPU).  Every rank then passes its
        own rows to ``update_*``; losses are normalised by the global batch (``args.batch_size`` must be
        the global size) and gradients / constraint sums are all-reduced (nlbac_amd/parallel.py).

        ``step_control`` (dopri5 only): "global" — the squared error norms of every attempted step are all-reduced, so
        all ranks share the step sizes and accept decisions the single-device run over the global batch takes (what the
        parity tests pin; one small collective per norm, inside the solve).  "shard" — every rank controls the steps of
        its own rows: no collective inside a solve, the ranks' solves run free of each other and meet at the gradient
        all-reduces only; the result is the single-device run with ``row_groups = world`` on its solvers (each shard an
        adaptive solve of its own, as if the reference had been handed the shards one by one), not bit-comparable
        with the global-norm run: the two differ by what a change of step size within rtol / atol changes."""
        from ..parallel import DataParallel
        assert step_control in ("global", "shard")
        self.dp = DataParallel(dist, group, always_collective)
        self.dp_step_control = step_control
        for ar in self.arenas:
            self.dp.broadcast_(ar.theta)
        self.ar_c.hard_update_target()
        self.dp.broadcast_(self.sc)
        self.repack_all()
        for sv in self.task.solvers:
            sv.comm = self.dp if step_control == "global" else None
        return self

    @property
    def world(self):
        return self.dp.world if self.dp is not None else 1

    def _exchange_buf(self, name, n):
        if name not in self._xb:
            self._xb[name] = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self._xb[name]

    def _adam(self, arena, lr, n_slabs, extra=None, target=None, tau=-1.0, before_step=None, alpha=None, mirror=None):
        """Adam on a whole arena.  One GPU: slab sum fused into the step.  Data parallel: local slab sum ->
        flat buffer (+ ``extra`` scalars riding along) -> all-reduce -> step on the reduced gradient."""
        s = stream_ptr()
        a = arena
        # one launch: ++step, slab sum, Adam, Polyak targets and the refresh of the nets' MFMA-fragment weight copies
        scat, scat_t = a.scatter_tables()
        scat_t = scat_t.data_ptr() if (scat_t is not None and target is not None and tau >= 0) else None
        # alpha = exp(log_alpha) refreshed by the step itself: (offsets of the log_alpha entries, where alpha goes)
        n_al = len(alpha[0]) if alpha else 0
        al_off = (C.c_long * 2)(*(list(alpha[0]) + [0, 0])[:2]) if alpha else None
        al_dst = (C.c_void_p * 2)(*(list(alpha[1]) + [None, None])[:2]) if alpha else None
        if self.world == 1:
            if before_step is not None:
                before_step(a.grad.data_ptr())
            # mirror: (pinned host block) the step's last workgroup writes the scalars block to (see _returns)
            _lib.call("nlbac_adam_fused", a.theta.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), a.grad.data_ptr(),
                      n_slabs, a.n, a.n, a.state.data_ptr(), lr, target, tau, scat.data_ptr(), scat_t, a.scatter_slots,
                      n_al, al_off, al_dst, self.sc.data_ptr() if mirror is not None else None,
                      mirror.data_ptr() if mirror is not None else None, SC.SC_SIZE if mirror is not None else 0, s)
            return
        n_extra = 0 if extra is None else extra.numel()
        xb = self._exchange_buf("g%d" % id(a), a.n + 4)
        _lib.call("nlbac_reduce_slabs", xb.data_ptr(), a.grad.data_ptr(), n_slabs, a.n, a.n, s)
        if n_extra:
            xb[a.n:a.n + n_extra].copy_(extra)
        self.dp.all_reduce_(xb)
        if n_extra:
            extra.copy_(xb[a.n:a.n + n_extra])
        if before_step is not None:
            before_step(xb.data_ptr())
        _lib.call("nlbac_adam_fused", a.theta.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), xb.data_ptr(), 1, a.n, a.n,
                  a.state.data_ptr(), lr, target, tau, scat.data_ptr(), scat_t, a.scatter_slots, n_al, al_off, al_dst, None,
                  None, 0, s)

    # ------------------------------------------------------------------ utils
    def repack_all(self):
        pack(self.h_crit + self.h_pols + self.h_node)
        pack(self.h_crit, target=True)

    def set_noise(self, eps_list):
        """Pre-drawn N(0,1) draws for the next update, reference order:
        [next_obs sample, obs sample, backup sample (, SimulatedCars: second-step sample, second-step
        backup sample)], each (B, n_u)."""
        self._noise = [torch.as_tensor(e, dtype=torch.float32) for e in eps_list]

    def _sc_pins(self):
        pin = self.__dict__.get("_sc_pin")
        if pin is None:
            pin = self._sc_pin = [torch.zeros(SC.SC_SIZE, dtype=torch.float32).pin_memory() for _ in range(3)]
            self._sc_ev = [torch.cuda.Event() for _ in range(3)]
            self._sc_lag = None
        return pin

    def _scalars(self):
        """Host copy of the device scalars (one 512-byte read through a pinned buffer; waits for the launch stream)."""
        pin = self._sc_pins()
        pin[0].copy_(self.sc, non_blocking=True)
        self._sc_ev[0].record()
        self._sc_ev[0].synchronize()
        return pin[0].numpy().copy()

    @property
    def alpha(self):
        return float(self._scalars()[SC.SC_ALPHA])

    @property
    def backup_alpha(self):
        return float(self._scalars()[SC.SC_BALPHA])

    @property
    def lambda_values(self):
        return [float(x) for x in self._scalars()[SC.SC_LAMBDA:SC.SC_LAMBDA + self.num_constraints]]

    @property
    def backup_lambda_values(self):
        return [float(x) for x in self._scalars()[SC.SC_BLAMBDA:SC.SC_BLAMBDA + self.num_cbfs]]

    @property
    def augmented_term(self):
        return float(self._scalars()[SC.SC_RHO_F64:SC.SC_RHO_F64 + 2].view(np.float64)[0])

    @property
    def backup_augmented_term(self):
        """Pvtol keeps a separate coefficient for the backup controller (P:59, 1033-1034)."""
        return float(self._scalars()[SC.SC_BRHO_F64:SC.SC_BRHO_F64 + 2].view(np.float64)[0])

    def get_control_bounds(self):
        u_min = torch.tensor(self.env.safe_action_space.low).to(self.device)
        u_max = torch.tensor(self.env.safe_action_space.high).to(self.device)
        return u_min, u_max

    # --------------------------------------------------------- action selection
    def _select(self, policy, state, evaluate, warmup):
        if not warmup and np.ndim(state) == 1:
            return policy.act(np.asarray(state, dtype=np.float64), evaluate)      # one observation: the latency path
        state = to_tensor(np.asarray(state, dtype=np.float64), torch.FloatTensor, self.device)
        expand_dim = len(state.shape) == 1
        if expand_dim:
            state = state.unsqueeze(0)
        if warmup:
            action = torch.stack([torch.from_numpy(self.action_space.sample()) for _ in range(state.shape[0])])
        else:
            a, _, mean = policy.sample(state)
            action = mean if evaluate else a
        action = action.detach().cpu().numpy()
        return action[0] if expand_dim else action

    def select_action(self, state, evaluate=False, warmup=False):
        return self._select(self.policy, state, evaluate, warmup)

    def select_action_backup(self, state, evaluate=False, warmup=False):
        if self.backup_policy is None:
            raise AttributeError("this variant has no backup controller")
        return self._select(self.backup_policy, state, evaluate, warmup)

    # ------------------------------------------------------------------ update
    def update_parameters(self, memory, batch_size, updates, dynamics_model, NODE_memory, NODE_model_update_interval,
                          i_episode=None):
        """Same contract as the reference (sac_cbf_clf.py:181-319): one sampled
        minibatch -> 6 floats.  ``dynamics_model`` is accepted for signature
        parity; obs->state runs on the device.  ``i_episode`` is the Pvtol copy's trailing argument (P:181):
        its NODE fit stops after episode 100 (P:205)."""
        fit = updates % NODE_model_update_interval == 0 and self.task.fit_due(i_episode)
        # (sac_cbf_clf.py:205: min(NODE_memory.position, 32768).  ``position`` wraps with the ring: a replay that has just
        #  wrapped reports 0 — the reference would then sample zero rows and fail — and the fit takes the filled size)
        nb = min(NODE_memory.position or len(NODE_memory), 32768) if fit else 0
        fit = fit and nb > 0
        if hasattr(memory, "sample_rows"):           # replay resident in HBM: gather on the device
            ws = self._workspace(batch_size)
            eps_ready = self._noise is None and getattr(memory, "device_rng", False)
            memory.sample_rows(batch_size, out=ws.mb, **({"eps_out": ws.eps} if eps_ready else {}))
            if fit:                                  # same draw order as the reference: minibatch, then NODE rows
                self.fit_node_rows(NODE_memory.sample_rows(nb) if hasattr(NODE_memory, "sample_rows") else
                                   self._rows_from_host(NODE_memory.sample(batch_size=nb)).to(self.device))
            return self.update_on_device(ws, updates, eps_ready=eps_ready)
        batch = memory.sample(batch_size=batch_size)
        node_rows = NODE_memory.sample(batch_size=nb) if fit else None
        return self.update_from_host(batch, updates, node_rows)

    def _rows_from_host(self, batch):
        """Pack the 10-tuple of ``ReplayMemory.sample`` into minibatch-layout rows (one H2D copy)."""
        lay = self.lay
        batch = tuple(batch)
        sig = None
        if lay.sig is not None:        # 11 fields: barrier_signal sits after the constraint
            sig, batch = batch[4], batch[:4] + batch[5:]
        state, action, reward, constraint, lya_in, next_lya_in, nstate, mask = batch[:8]
        n = np.asarray(state).shape[0]
        host = np.zeros((n, lay.LD), dtype=np.float32)
        if sig is not None:
            host[:, lay.sig] = sig
        host[:, lay.obs:lay.obs + lay.obs_dim] = state
        host[:, lay.act:lay.act + lay.act_dim] = np.asarray(action).reshape(n, lay.act_dim)
        host[:, lay.rew], host[:, lay.con] = reward, constraint
        host[:, lay.lya:lay.lya + lay.lya_dim] = lya_in
        host[:, lay.nlya:lay.nlya + lay.lya_dim] = next_lya_in
        host[:, lay.nobs:lay.nobs + lay.obs_dim] = nstate
        host[:, lay.mask] = mask
        if len(batch) > 9 and batch[8] is not None and np.asarray(batch[8]).dtype != object:
            host[:, lay.t], host[:, lay.nt] = batch[8], batch[9]
        return torch.from_numpy(host)

    def update_from_host(self, batch, updates, node_batch=None):
        """batch / node_batch: tuples of numpy arrays in the field order of ``ReplayMemory.sample``
        (node_batch may also be the short form (obs, action, next_obs[, t]))."""
        B = np.asarray(batch[0]).shape[0]
        ws = self._workspace(B)
        ws.mb.copy_(self._rows_from_host(batch), non_blocking=False)
        if node_batch is not None:
            if len(node_batch) <= 4:       # (obs, action, next_obs[, t]) -> full field order
                o, a_, no = node_batch[:3]
                n = np.asarray(o).shape[0]
                t = np.asarray(node_batch[3]).reshape(n) if len(node_batch) == 4 else np.zeros(n)
                zl = np.zeros((n, self.lay.lya_dim))
                node_batch = (o, a_, np.zeros(n), np.zeros(n)) + ((np.zeros(n),) if self.lay.sig is not None else ()) \
                    + (zl, zl, no, np.ones(n), t, t)
            self.fit_node_rows(self._rows_from_host(node_batch).to(self.device))
        return self.update_on_device(ws, updates)

    def _workspace(self, B):
        if B not in self._ws:
            self._ws[B] = _Workspace(B, self.hidden, self.device, self.lay, self.task)
        return self._ws[B]

    # -- NODE fit (model.py:221-260 via sac_cbf_clf.py:205-219) -----------------
    def fit_node_rows(self, rows):
        """One Adam step of the NODE regression on a device tensor of minibatch-layout rows (N, LD)
        (obs, action, next_obs and, where the model takes it, t are read in place)."""
        self._fit(*self.task.fit_inputs(rows))

    def _fit(self, p_obs, obs_ld, action, p_nobs, nobs_ld, N):
        task = self.task
        self._fill.clear()      # (pieces of an update that raised half-way must not run behind this fit's solves)
        if N not in self._fit_ws:
            while len(self._fit_ws) >= 2:       # the fit batch grows with the replay: keep the two latest sizes only
                self._fit_ws.pop(next(iter(self._fit_ws)))      # (its graphs go with it)
            w = task.fit_ws(N)
            w.update(graphs={}, warm=0)
            self._fit_ws[N] = w
        w = self._fit_ws[N]
        w["u"].copy_(action)
        key = (p_obs, obs_ld, p_nobs, nobs_ld, self.solver)
        part1 = lambda: task.fit_part1(w, p_obs, obs_ld, p_nobs, nobs_ld, N)
        if not self._graphs_on() or w["warm"] < 1:
            w["warm"] += 1
            part1()
            self._fit_part2(w, N, task.fit_solver.forward_finish())
            return
        g = w["graphs"]
        self._replay(self._graph(g, ("p1",) + key, part1))
        if self.solver == "dopri5" and not task.fit_solver.first_step_done():
            self._fit_part2(w, N, task.fit_solver.forward_finish())      # rare: finish this one eagerly
            return
        self._replay(self._graph(g, ("p2",) + key,
                                 lambda: self._fit_part2(w, N, task.fit_solver.forward_finish(assume_single_step=True))))

    def _fit_part2(self, w, N, pred):
        s = stream_ptr()
        ns = self.task.n_s
        nblk = (N + 255) // 256
        NG = N * self.world
        _lib.call("nlbac_mse_fwd_bwd", pred.data_ptr(), ns, w["nst"].data_ptr(), ns, N, NG, ns, w["dpred"].data_ptr(),
                  ns, w["part"].data_ptr(), s)
        _lib.call("nlbac_sum_partials", w["part"].data_ptr(), nblk, 1, 1.0 / (NG * ns),
                  self.sc.data_ptr() + 4 * SC.SC_NODE_LOSS, s)
        self.task.fit_solver.backward(w["dpred"], need_du=False, need_params=True)
        # every accepted RK step writes its own gradient slabs; a solve with many steps takes narrower ones
        n_steps = max(1, len(self.task.fit_solver.ctx.get("steps") or [None]))
        used = self.task.fit_solver.accumulate_param_grads(
            self.ar_n, max(1, min(self.n_fit_slabs, self.ar_n.n_slabs // min(n_steps, self.ar_n.n_slabs))))
        self._adam(self.ar_n, 1e-3, used, extra=self.sc[SC.SC_NODE_LOSS:SC.SC_NODE_LOSS + 1])

    def _capture(self, fn):
        """Record the launches of ``fn`` into a hipGraph (all kernel arguments are static device pointers /
        constants; per-update scalars live in device memory).  The graph bakes in the addresses of the solvers'
        buffers and belongs to the solves it recorded: the entry keeps the solvers' ``generation`` (any freed or
        re-laid-out buffer invalidates it) and their solve contexts (restored before a replay, so that what the host
        does around the replay — reading the control block, finishing a solve eagerly — talks about THIS graph's
        solve and not about whichever batch size ran last)."""
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        svs = self.task.solvers
        return types.SimpleNamespace(graph=g, gens=tuple(sv.generation for sv in svs),
                                     ctxs=[(sv, sv.__dict__.get("ctx"), sv.__dict__.get("_cur_n")) for sv in svs])

    def _graph(self, cache, key, fn):
        """The captured graph for ``key``, re-captured if the solvers' buffers have moved since."""
        e = cache.get(key)
        if e is not None and e.gens != tuple(sv.generation for sv in self.task.solvers):
            e = None
        if e is None:
            e = cache[key] = self._capture(fn)
        return e

    def _replay(self, e):
        for sv, ctx, n in e.ctxs:
            if ctx is not None:
                sv.ctx, sv._cur_n = ctx, n
        e.graph.replay()

    # -- the update proper --------------------------------------------------------
    def _plan(self, ws, NP=None):
        """ctypes launch descriptors of one workspace: every pointer is static (arenas, workspace tensors),
        so they are built once (per number of controllers updated: Pvtol trains its backup every 20th update);
        an update is then a plain sequence of C calls."""
        NP = NP or self.task.n_pol
        if ws.plan is None:
            ws.plan = {}
        if NP in ws.plan:
            return ws.plan[NP]
        B, lay = ws.B, self.lay
        mb = ws.mb.data_ptr()
        LD, Do, Da, Dl = lay.LD, lay.obs_dim, lay.act_dim, lay.lya_dim
        col = lambda c: mb + 4 * c
        lc, lnc = self.task.lya_train_cols(lay)       # inputs the Lyapunov critic is regressed on
        p_obs, p_act, p_cen, p_ncen, p_nobs = col(lay.obs), col(lay.act), col(lc), col(lnc), col(lay.nobs)

        def x(io, i, p0, d0, ld0, p1=None, d1=0, ld1=0):
            io[i].x0, io[i].x0_dim, io[i].x0_ld = p0, d0, ld0
            if p1 is not None:
                io[i].x1, io[i].x1_dim, io[i].x1_ld = p1, d1, ld1
        P = types.SimpleNamespace()
        P.p_obs, P.p_rew, P.p_con, P.p_mask, P.LD = p_obs, col(lay.rew), col(lay.con), col(lay.mask), LD
        q1, q2, l, pi = self.h_q1, self.h_q2, self.h_l, self.h_p
        P.NP, NX = NP, len(self.h_extra)
        # A: pi(s')
        P.n_pol, P.io_pol_next = mlp_array([pi.desc]), io_array(1)
        x(P.io_pol_next, 0, p_nobs, Do, LD)
        P.io_pol_next[0].y, P.io_pol_next[0].y_ld = ws.heads_n.data_ptr(), 2 * Da
        # A: targets + critic / Lyapunov forward (6 nets)
        descs = [q1.desc_target, q2.desc_target, l.desc_target, q1.desc, q2.desc, l.desc]
        for h in self.h_extra:
            descs += [h.desc_target, h.desc]
        P.n_six, P.n_six_count = mlp_array(descs), len(descs)
        io = P.io_six = io_array(len(descs))
        for i in range(len(descs)):
            io[i].y, io[i].y_ld = ws.q6[i].data_ptr(), 1
        for k in range(NX):            # extra critic-type nets on (s', a') [target] and (s, a)
            x(io, 6 + 2 * k, p_nobs, Do, LD, ws.na.data_ptr(), Da, Da)
            x(io, 7 + 2 * k, p_obs, Do, LD, p_act, Da, LD)
            io[7 + 2 * k].acts = ws.acts_c[3 + k].data_ptr()
        for i in (0, 1):
            x(io, i, p_nobs, Do, LD, ws.na.data_ptr(), Da, Da)
        x(io, 2, p_ncen, Dl, LD)
        for i in (3, 4):
            x(io, i, p_obs, Do, LD, p_act, Da, LD)
            io[i].acts = ws.acts_c[i - 3].data_ptr()
        x(io, 5, p_cen, Dl, LD)
        io[5].acts = ws.acts_c[2].data_ptr()
        # B: critic / Lyapunov backward
        P.n_crit = mlp_array([h.desc for h in self.h_crit])
        io = P.io_crit = io_array(3 + NX)
        for i in range(3 + NX):
            io[i].dy, io[i].dy_ld = ws.dq3[i].data_ptr(), 1
            io[i].acts, io[i].dz = ws.acts_c[i].data_ptr(), ws.dz_c[i].data_ptr()
            io[i].grad = self.ar_c.grad.data_ptr()
        for i in [0, 1] + list(range(3, 3 + NX)):
            x(io, i, p_obs, Do, LD, p_act, Da, LD)
        x(io, 2, p_cen, Dl, LD)
        # (its data backward leaves the skinny-gradient partial sums for the weight backward: one launch less)
        P.sk_crit = skinny_partials_ws(P.n_crit, (io,), 3 + NX, B, self.device)
        # C: both actors (forward and backward share one descriptor)
        def act_io(io, j, i):            # entry j of an io array describes controller i
            x(io, j, p_obs, Do, LD)
            io[j].y, io[j].y_ld = ws.heads2[i * B:].data_ptr(), 2 * Da
            io[j].acts, io[j].dz = ws.acts_p[i].data_ptr(), ws.dz_p[i].data_ptr()
            io[j].dy, io[j].dy_ld = ws.dheads2[i * B:].data_ptr(), 2 * Da
            io[j].grad = self.pol_arena[i].grad.data_ptr()
        P.n_act = mlp_array([h.desc for h in self.h_pols[:NP]])
        io = P.io_act = io_array(NP)
        for i in range(NP):
            act_io(io, i, i)
        P.n_pol3 = mlp_array([pi.desc] + [h.desc for h in self.h_pols[:NP]])     # pi(s') + the actors on s
        io3 = P.io_pol3 = io_array(1 + NP)
        x(io3, 0, p_nobs, Do, LD)
        io3[0].y, io3[0].y_ld = ws.heads_n.data_ptr(), 2 * Da
        for i in range(NP):
            act_io(io3, 1 + i, i)
        P.act_groups = []                # per Adam group: the nets whose weight gradients land in its arena
        for g in self.actor_groups:
            cnt = min(g.count, NP - g.first)
            if cnt <= 0:
                continue
            gio = io_array(cnt)
            for j in range(cnt):
                act_io(gio, j, g.first + j)
            nets_g = mlp_array([h.desc for h in self.h_pols[g.first:g.first + cnt]])
            # the actors' data backward (P.io_act) leaves this group's skinny-gradient partials for its weight backward
            io_act_g = [P.io_act[g.first + j] for j in range(cnt)]       # (views into the array, not copies)
            sk = skinny_partials_ws(nets_g, (gio, io_act_g), cnt, B, self.device)
            P.act_groups.append((g, cnt, nets_g, gio, sk))
        # C: Q(s, pi) for primary / backup + V(current Lyapunov input)
        extra = self.task.extra_value_nets()
        P.n_q5 = mlp_array([q1.desc, q2.desc] * NP + [l.desc] + [h.desc for h in extra])
        P.n_q5_count = 2 * NP + 1 + len(extra)
        io = P.io_q5 = io_array(P.n_q5_count)
        for i in range(2 * NP):
            half = i // 2                                      # 0 primary, 1 backup
            x(io, i, p_obs, Do, LD, ws.pi2[half * B:].data_ptr(), Da, Da)
            io[i].y, io[i].y_ld = ws.qpi[i % 2, half * B:].data_ptr(), 1
            io[i].acts = ws.acts_q[i].data_ptr()
            io[i].dy, io[i].dy_ld = ws.dq_pi[i % 2, half * B:].data_ptr(), 1
            io[i].dx, io[i].dx_ld = ws.dxq[i % 2, half * B:].data_ptr(), Do + Da
            io[i].dx_first = Do                                # (only dQ / da is consumed)
        self.task.value_now_io(ws, io, 2 * NP)
        self.task.extra_value_io(ws, io, 2 * NP + 1)
        self.task.plan(ws, P)
        self._masks_for_dx_only_nets(ws, P)
        ws.plan[NP] = P
        return P

    def _masks_for_dx_only_nets(self, ws, P):
        """Where the register-resident MLP kernels serve the heads, every forward that saves something for a backward
        also leaves ReLU mask words (``nlbac_mlp_io::masks``, 64 B per row) and every data backward gates with them
        instead of loading the activation rows (24 float4 per lane at the head of each tile's critical path).  Nets that
        are only differentiated w.r.t. their inputs (Q(s, pi), V(p(x')), the barrier on predicted states: dx wanted, no
        dz / weight gradients) then keep nothing else — their activation buffer is dropped from the descriptors: the
        forward's 2 KB-per-row store burst goes.  Works on the finished launch descriptors: an activation buffer that no
        descriptor pairs with a dz buffer is such a net's."""
        if not self.fold_launches or not _lib.load().nlbac_mlp_masks_ok(mlp_array([h.desc for h in self.h_crit + self.h_pols]),
                                                                       len(self.h_crit) + len(self.h_pols)):
            return
        arrays = [v for v in P.__dict__.values() if isinstance(v, C.Array) and getattr(v, "_type_", None) is _lib.MlpIO]
        arrays += [g[3] for g in P.act_groups]
        rows = set()
        for arr in arrays:
            for e in arr:
                if e.acts and (e.dz or e.grad or e.skinny_ws):
                    rows.add(e.acts)
        bufs = ws.__dict__.setdefault("_mask_bufs", {})
        for arr in arrays:
            for e in arr:
                if e.acts:
                    if e.acts not in bufs:
                        bufs[e.acts] = torch.zeros(2, ws.B, 8, dtype=torch.int32, device=self.device)
                    e.masks = bufs[e.acts].data_ptr()
                    if e.acts not in rows:
                        e.acts = None

    def auglag_fused(self, ws, n_cbf, lam_upd):
        """The (fused, ticket, sc) tail of a ``*_constraints_fwd`` call: on one GPU the launch's last workgroup runs the
        augmented-Lagrangian step (``nlbac_auglag``) itself; under data parallelism the sums are all-reduced first."""
        if self.world > 1:
            return None, None, None
        A = _lib.AuglagArgs()
        A.n_cbf, A.n_clf, A.batch_size = n_cbf, 1, float(self.batch_size)
        A.do_lambda_update, A.do_backup_lambda_update = lam_upd, ws.blam_upd
        A.ratio_mode = self.task.ratio_mode
        A.backup_mode = self.task.backup_mode if ws.np_now == 2 else 0
        A.lam_lo, A.lam_hi = 0.01, self.task.lam_hi
        ws._auglag_args = A          # (kept alive until the call has been made)
        return C.byref(A), self._tickets.data_ptr() + 4 * 12, self.sc.data_ptr()

    def auglag(self, ws, n_cbf, lam_upd):
        """required_matrix, ratio, lambda / rho updates and loss coefficients from the constraint partial sums
        (all-reduced first under data parallelism: they enter the loss nonlinearly).  Single GPU: already done by the
        constraints launch (``auglag_fused``)."""
        s, call = stream_ptr(), _lib.call
        NP = ws.np_now
        if self.world == 1:
            ws.p_part_q, ws.n_part_q = ws.part_q.data_ptr(), ws.nblk
            return
        ncol = n_cbf + 1 + (n_cbf if NP == 2 else 0)
        p_part_c, n_part = ws.part_c.data_ptr(), ws.nblk
        ws.p_part_q, ws.n_part_q = ws.part_q.data_ptr(), ws.nblk
        if self.world > 1:
            xs = self._exchange_buf("sums", 64)
            call("nlbac_sum_partials", ws.part_c.data_ptr(), ws.nblk, ncol, 1.0, xs.data_ptr(), s)
            for pp in range(NP):
                call("nlbac_sum_partials", ws.part_q[pp].data_ptr(), ws.nblk, 2, 1.0, xs.data_ptr() + 4 * (32 + 2 * pp), s)
            self.dp.all_reduce_(xs)
            p_part_c, n_part = xs.data_ptr(), 1
            ws.p_part_q, ws.n_part_q = xs.data_ptr() + 4 * 32, 1
        call("nlbac_auglag", p_part_c, n_part, n_cbf, 1, float(self.batch_size), lam_upd, ws.blam_upd,
             self.task.ratio_mode, self.task.backup_mode if NP == 2 else 0, 0.01, self.task.lam_hi,
             self.sc.data_ptr(), s)

    def update_on_device(self, ws, updates, sync=True, eps_ready=False, prefetch=None):
        """Minibatch already in ``ws.mb``; returns the reference's 6 floats.  ``eps_ready``: ``ws.eps`` already holds
        this update's N(0,1) draws (``DeviceReplayMemory.sample_rows(..., eps_out=ws.eps)``).

        ``prefetch``: a callable that draws a minibatch into ``ws.mb`` / ``ws.eps`` with device launches only
        (``lambda: replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)``); the caller then never draws itself.  This update's
        rows are drawn by it unless the previous call already has: behind its last launch (the actors' optimiser step —
        nothing reads ``ws.mb`` after it) every update queues the NEXT update's draw and policy forward before the host
        blocks on the six returned floats, so the update boundary — the host waking up, returning to its loop and coming
        back with the first launches, ~30 us of idle GPU otherwise — is covered by ~30 us of queued work.  The draw sees
        the replay as of the end of this update: a driver that pushes transitions between two updates and wants them
        eligible at once does not pass ``prefetch``."""
        pre = ws.__dict__.get("_prefetched")
        ws._prefetched = None
        if prefetch is not None:
            eps_ready = True
            if pre is None or pre[0] != updates:
                pre = None
                prefetch()
        else:
            pre = None
        ws._pre_now, ws._prefetch_fn, ws._sync = pre, prefetch, sync
        if self._noise is not None:
            assert len(self._noise) == self.task.n_eps, "set_noise needs %d draws" % self.task.n_eps
            order = self.task.eps_order or range(self.task.n_eps)     # device slot -> reference draw index
            for i, j in enumerate(order):
                ws.eps[i].copy_(self._noise[j].to(self.device).reshape(ws.eps[i].shape))
            self._noise = None
        elif not eps_ready:
            ws.eps.normal_()
        soft = (updates % self.target_update_interval == 0)
        lam_upd = 1 if updates % self.Lagrangian_multiplier_update_interval == 0 else 0
        NP = ws.np_now = self.task.n_pol_now(updates)
        ws.updates_now = updates
        assert ws._pre_now is None or ws._pre_now[1] == NP
        ws.blam_upd = self.task.backup_lam_due(updates, self.Lagrangian_multiplier_update_interval)
        # where the update's last launch (the actors' optimiser step) leaves the scalars block for the host: straight
        # in pinned memory, so that no copy launch sits between that step and the host's wait.  Not under hipGraph
        # replay (the address would be baked in) or data parallelism (the step is not the last thing that happens).
        self._mirror = None
        if sync and self.world == 1 and not self._graphs_on() and self.fold_launches:
            pin = self._sc_pins()
            if sync == "lagged":
                k = 1 + (self.__dict__.get("_sc_flip", 0) & 1)
                self._sc_flip = k
            else:
                k = 0
            self._mirror = (k, pin[k])
        if not self._graphs_on() or ws.warm < 1:
            ws.warm += 1
            self._upd_part1(ws, soft)
            self._upd_part2(ws, lam_upd, False)
        else:
            # hipGraph replay: part 1 up to the dopri5 accept decision, one 256-byte read, part 2
            g = ws.graphs
            k1, k2 = ("p1", soft, self.solver, NP), ("p2", lam_upd, ws.blam_upd, self.solver, NP)
            self._replay(self._graph(g, k1, lambda: self._upd_part1(ws, soft)))
            if self.solver == "dopri5" and not self.task.first_step_done():
                self._upd_part2(ws, lam_upd, False)          # rare: finish eagerly
            else:
                self._replay(self._graph(g, k2, lambda: self._upd_part2(ws, lam_upd, True)))
        return self._returns(sync)

    def _returns(self, sync):
        """The reference's 6 floats.  ``sync``: True — of this update (the host waits for it, as the reference's
        ``.item()`` calls do); "lagged" — of the previous ``"lagged"`` call (None on the first), while this update's
        are on their way to pinned memory: the launch stream never drains, for drivers that only log the values;
        False — nothing."""
        if not sync:
            return None
        mirrored = self.__dict__.get("_mirror_done")      # (buffer index the last optimiser step wrote to, or None)
        self._mirror_done = None
        if sync == "lagged":
            if mirrored is not None:
                k = mirrored
            else:
                self._sc_pins()
                k = 1 + (self.__dict__.get("_sc_flip", 0) & 1)
                self._sc_flip = k
                self._sc_pin[k].copy_(self.sc, non_blocking=True)
            prev, self._sc_lag = self._sc_lag, k
            if mirrored is None:
                self._sc_ev[k].record()
            if prev is None:
                return None
            self._sc_ev[prev].synchronize()
            h = self._sc_pin[prev].numpy().copy()
        elif mirrored is not None:
            self._sc_ev[mirrored].synchronize()
            h = self._sc_pin[mirrored].numpy().copy()
        else:
            h = self._scalars()
        alpha_loss = float(h[SC.SC_ALOSS]) if self.automatic_entropy_tuning else 0.0
        return (float(h[SC.SC_QF1]), float(h[SC.SC_QF2]), float(h[SC.SC_LF]), float(h[SC.SC_PL1]),
                alpha_loss, float(h[SC.SC_ALPHA]))

    def _upd_part1(self, ws, soft):
        """Phases A, B and the forward half of C up to the rollout's first host decision point."""
        B, A = ws.B, self.lay.act_dim
        G = B * self.world                      # rows the batch means run over
        s = stream_ptr()
        NP = ws.np_now
        P = self._plan(ws, NP)
        LD = P.LD

        # ---- A. targets (no grad): pi(s'), Q_target(s', a'), L_target(c') ; critic / Lyapunov forward
        if ws.__dict__.get("_pre_now") is None:       # (else: queued behind the previous update's last launch, see prefetch)
            self._policy_forward(ws, P, NP)
        # the rollout of the learned dynamics needs only pi(s) and the NODE: its first attempted step goes in here,
        # so that the critic phase below is queued behind it while the host waits for the accept decision
        self.task.rollout_begin(ws, P)
        # The rest of part 1 does not depend on the rollout.  It is cut into three pieces that the solvers pull in one
        # at a time just before each wait for an accept decision (``before_wait``), so that every attempted step —
        # the second of a two-step solve, the second and third solve of Pvtol's chain — has work queued behind it;
        # whatever is left goes in when the rollout is finished (``drain_fill``, called by the task).
        self._fill = collections.deque((lambda: self._part1_targets(ws, P, B, G, LD, s),
                                        lambda: self._part1_critic_step(ws, P, B, soft),
                                        lambda: self._part1_actor_q(ws, P, B, G, NP, s)))
        # how many pieces the first wait pulls: a rollout of ONE adaptive solve gets them all at once (the host needs
        # ~100 us of queued work to read the decision and come back); chained solves (SimulatedCars 2, Pvtol 3) one per wait
        self._fill_first = self.task.rollout_waits == 1
        if ws.__dict__.get("_pre_now") is not None and ws._pre_now[2]:
            self._fill.popleft()                # (targets + critic data backward: queued with the prefetch)
            # (the first wait still queues both remaining pieces: holding the Q(s, pi) forward back for the launch it
            #  could share with V(p(x')) left the stream dry for ~20 us while the host got from the accept decision to
            #  that launch)
        if self.solver != "dopri5" or torch.cuda.is_current_stream_capturing():
            self.drain_fill()                   # no waits on this path (fixed-step solver / graph capture)

    def _policy_forward(self, ws, P, NP):
        """pi(s') and the actors on s.  (The actors' forward on s does not depend on the critic step: it shares pi(s')'s
        launch, and all (1+NP)*B samples are drawn by one launch - eps[0 .. NP] are contiguous.)"""
        B, A = ws.B, self.lay.act_dim
        s, call = stream_ptr(), _lib.call
        pol = self.policy
        p_scale, p_bias = pol.action_scale.data_ptr(), pol.action_bias.data_ptr()
        if self.fold_launches:      # (the samples are drawn by the policy launch itself: nlbac_gauss_head)
            gh = P.__dict__.get("head_pol3")
            if gh is None:
                gh = P.head_pol3 = _lib.GaussHead()
                gh.eps, gh.scale, gh.bias, gh.n_u = ws.eps.data_ptr(), p_scale, p_bias, A
                gh.action, gh.action_ld, gh.logp = ws.act3.data_ptr(), A, ws.logp3.data_ptr()
            call("nlbac_mlp_fwd_gauss", P.n_pol3, P.io_pol3, 1 + NP, B, C.byref(gh), s)
        else:
            call("nlbac_mlp_fwd", P.n_pol3, P.io_pol3, 1 + NP, B, s)
            call("nlbac_gauss_sample_fwd", ws.heads3.data_ptr(), 2 * A, ws.eps.data_ptr(), p_scale, p_bias, A, (1 + NP) * B,
                 ws.act3.data_ptr(), A, ws.logp3.data_ptr(), s)

    def _prefetch_next(self, ws, updates):
        """Behind this update's last launch: the next update's minibatch draw and first launches (update_on_device)."""
        fn = ws.__dict__.get("_prefetch_fn")
        if fn is None or self._graphs_on() or self.world != 1 or self._noise is not None:
            return
        if ws._sync and self.__dict__.get("_mirror") is None:
            return      # (the returned floats would be copied BEHIND the queued launches, whose dy head rewrites the losses)
        fn()
        NP = self.task.n_pol_now(updates + 1)
        ws._prefetched = (updates + 1, NP, self._prefetch_launches(ws, NP))

    def update_prefetch(self, ws, updates, prefetch):
        """Draw update ``updates``'s minibatch (``prefetch``, see update_on_device) and queue its policy forward now — what
        every update does for its successor; for callers that need the draw to come before something else they queue."""
        if self._graphs_on() or self.world != 1:
            return
        prefetch()
        NP = self.task.n_pol_now(updates)
        ws._prefetched = (updates, NP, self._prefetch_launches(ws, NP))

    def _prefetch_launches(self, ws, NP):
        """What of an update needs nothing but its minibatch and the parameters as the previous update left them: the
        policy forward, then the target / critic forward and the critics' data backward (~95 us of launches: more than
        the host needs to come round to the next update)."""
        P = self._plan(ws, NP)
        self._policy_forward(ws, P, NP)
        # (a rollout of three chained solves keeps the targets piece for its waits: each solve's accept decision is a
        #  host round trip, and a piece of independent work queued behind every one of them is worth more than a longer
        #  head start — the policy forward alone covers the update boundary at those batch sizes.  Returns whether the
        #  targets piece went out.)
        if self.task.rollout_waits >= 3:
            return False
        self._part1_targets(ws, P, ws.B, ws.B * self.world, P.LD, stream_ptr())
        return True

    def _fill_one(self):
        """Called by a solver just before it waits for an accept decision: queue the next piece(s) of part 1 behind the
        attempted step.  The first wait of an update gets two (the host needs ~100 us of queued work to read the
        decision and launch what follows without the stream running dry), later ones one each."""
        k, self._fill_first = (2 if self.__dict__.get("_fill_first", True) else 1), False
        while k and self._fill:
            self._fill.popleft()()
            k -= 1

    def drain_fill(self):
        while self._fill:
            self._fill.popleft()()

    def _part1_targets(self, ws, P, B, G, LD, s):
        sc, call = self.sc.data_ptr(), _lib.call
        one = self.world == 1
        call("nlbac_mlp_fwd", P.n_six, P.io_six, P.n_six_count, B, s)
        q = ws.q6
        if one and len(self.h_extra) <= 1 and self.fold_launches:
            # single GPU, no extra critic: targets, dL/dq and the three losses are produced by the critics' data backward
            # itself (nlbac_dy_head kind 2) — no launch between the six-net forward and the backward
            H = P.__dict__.get("head_td")
            if H is None:
                H = P.head_td = _lib.DyHead()
                H.kind, H.B_norm = 2, G
                H.q1t, H.q2t, H.lt, H.nlogp = q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), ws.nlogp.data_ptr()
                H.reward, H.constraint, H.mask, H.rcm_ld = P.p_rew, P.p_con, P.p_mask, LD
                H.alpha, H.gamma = sc + 4 * SC.SC_ALPHA, self.gamma
                for k in range(3):
                    H.q[k], H.dq[k] = q[3 + k].data_ptr(), ws.dq3[k].data_ptr()
                H.next_q, H.next_l = ws.next_q.data_ptr(), ws.next_l.data_ptr()
                H.partials, H.ticket = ws.part_td32.data_ptr(), ws.tickets_td.data_ptr()
                H.mul, H.out = 1.0 / G, sc + 4 * SC.SC_QF1
                if self._sums_defer():
                    # no election at the end of this launch: its tiles leave their squared-error sums, a workgroup of
                    # the actors' data backward (the last MLP launch of the update) adds them up (nlbac_head_sums)
                    H.sums_defer, H.sums_tiles = 1, ws.sums_tiles.data_ptr()
                if self.h_extra:        # BarrierNet TD step (NU/sac_cbf_clf.py:224-233): the launch's 4th net
                    H.xt, H.xq, H.dxq = q[6].data_ptr(), q[7].data_ptr(), ws.dq3[3].data_ptr()
                    H.xsig, H.xsig_ld = ws.mb.data_ptr() + 4 * self.lay.sig, LD
                    H.out_x = sc + 4 * SC.SC_XLOSS
            call("nlbac_mlp_bwd_data_head", P.n_crit, P.io_crit, len(self.h_crit), B, C.byref(H), s)
            return
        call("nlbac_td_targets", q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), ws.nlogp.data_ptr(),
             P.p_rew, P.p_con, P.p_mask, LD, q[3].data_ptr(), q[4].data_ptr(), q[5].data_ptr(),
             sc + 4 * SC.SC_ALPHA, self.gamma, B, G, ws.dq3[0].data_ptr(), ws.dq3[1].data_ptr(), ws.dq3[2].data_ptr(),
             ws.next_q.data_ptr(), ws.next_l.data_ptr(), ws.part_td.data_ptr(),
             *((self._tickets.data_ptr(), 1.0 / G, sc + 4 * SC.SC_QF1) if one else (None, 0.0, None)), s)
        if not one:       # (single GPU: the launch's last workgroup has summed the three losses itself)
            call("nlbac_sum_partials", ws.part_td.data_ptr(), ws.nblk, 3, 1.0 / G, sc + 4 * SC.SC_QF1, s)
        for k in range(len(self.h_extra)):      # barrier TD step (NU/sac_cbf_clf.py:224-233)
            call("nlbac_td_value", q[6 + 2 * k].data_ptr(), ws.mb.data_ptr() + 4 * self.lay.sig, LD, P.p_mask, LD,
                 q[7 + 2 * k].data_ptr(), self.gamma, B, G, ws.dq3[3 + k].data_ptr(), None, ws.part_tdx[k].data_ptr(),
                 *((self._tickets.data_ptr() + 4 * (1 + k), 1.0 / G, sc + 4 * SC.SC_XLOSS) if one else (None, 0.0, None)), s)
            if not one:
                call("nlbac_sum_partials", ws.part_tdx[k].data_ptr(), ws.nblk, 1, 1.0 / G, sc + 4 * SC.SC_XLOSS, s)

        # ---- B. critic / Lyapunov backward + Adam (+ Polyak targets) ---------------
        call("nlbac_mlp_bwd_data", P.n_crit, P.io_crit, len(self.h_crit), B, s)

    def _part1_critic_step(self, ws, P, B, soft):
        a = self.ar_c
        bwd_weights(P.n_crit, P.io_crit, len(self.h_crit), B, a.n_slabs, a.n, self.device, ws=P.sk_crit)
        self._adam(a, self.critic_lyapunov_lr, a.n_slabs, extra=self.sc[SC.SC_QF1:SC.SC_QF1 + 3],
                   target=a.target.data_ptr(), tau=self.tau if soft else -1.0)

    def _part1_actor_q(self, ws, P, B, G, NP, s):
        # ---- C. actors: Q(s, pi) with the stepped critics (the rollout was started in phase A) -----
        sc, call = self.sc.data_ptr(), _lib.call
        call("nlbac_mlp_fwd", P.n_q5, P.io_q5, P.n_q5_count, B, s)
        if self.world == 1 and self.fold_launches:
            return               # (the branch terms and their sums come out of the Q(s, pi) data backward: _actor_q_head)
        fused = None
        if self.world == 1:      # policy_loss_1 / alpha losses / d log_alpha by the launch's last workgroup (nlbac_actor_scalars)
            fused = P.__dict__.get("actor_scalars")
            if fused is None:
                fused = P.actor_scalars = _lib.ActorScalarArgs()
                fused.target_entropy, fused.sc = self.target_entropy, sc
                for g in self.actor_groups:
                    for k in range(min(g.count, NP - g.first)):
                        off = g.la_off + k * g.la_stride
                        fused.log_alpha[g.first + k] = g.arena.theta.data_ptr() + 4 * off
                        fused.g_log_alpha[g.first + k] = g.arena.grad.data_ptr() + 4 * off
        call("nlbac_actor_q_terms", ws.qpi[0].data_ptr(), ws.qpi[1].data_ptr(), ws.logp2.data_ptr(),
             sc + 4 * SC.SC_ALPHA, B, G, NP, ws.dq_pi[0].data_ptr(), ws.dq_pi[1].data_ptr(), ws.part_q.data_ptr(),
             C.byref(fused) if fused is not None else None, self._tickets.data_ptr() + 4 * 8 if fused is not None else None, s)

    def _actor_q_head(self, ws, P, NP, G):
        H = P.__dict__.get("head_actor_q")
        if H is None:
            sc = self.sc.data_ptr()
            H = P.head_actor_q = _lib.DyHead()
            H.kind, H.B_norm, H.n_prob = 3, G, NP
            H.qa, H.qb, H.logp = ws.qpi[0].data_ptr(), ws.qpi[1].data_ptr(), ws.logp2.data_ptr()
            H.dqa, H.dqb = ws.dq_pi[0].data_ptr(), ws.dq_pi[1].data_ptr()
            H.alpha = sc + 4 * SC.SC_ALPHA
            H.actor.target_entropy, H.actor.sc = self.target_entropy, sc
            for g in self.actor_groups:
                for k in range(min(g.count, NP - g.first)):
                    off = g.la_off + k * g.la_stride
                    H.actor.log_alpha[g.first + k] = g.arena.theta.data_ptr() + 4 * off
                    H.actor.g_log_alpha[g.first + k] = g.arena.grad.data_ptr() + 4 * off
            H.partials, H.ticket = ws.part_q32.data_ptr(), ws.tickets_q.data_ptr()
            if self._sums_defer():
                H.sums_defer, H.sums_tiles = 1, ws.sums_tiles.data_ptr() + 4      # (as the td head's, see _part1_targets)
        return H

    def _sums_defer(self):
        """The batch sums of the td / actor-q dy heads are finished by a workgroup of the actors' data backward instead of
        by an election at the end of their own launches (nlbac_dy_head::sums_defer / finish): single GPU with the launch
        folds (the heads exist and the actors' backward follows them in every update).  ``sums_defer = False``
        (NLBAC_SUMS_DEFER=0): the elections."""
        on = self.__dict__.get("sums_defer")
        if on is None:
            on = self.sums_defer = os.environ.get("NLBAC_SUMS_DEFER", "1") != "0"
        return bool(on and self.world == 1 and self.fold_launches)

    def _upd_part2(self, ws, lam_upd, assume_single):
        """Constraints, augmented-Lagrangian scalars, the whole actor backward and the actor Adam step."""
        B, A, Do = ws.B, self.lay.act_dim, self.lay.obs_dim
        G = B * self.world
        s = stream_ptr()
        NP = ws.np_now
        P = self._plan(ws, NP)
        sc = self.sc.data_ptr()
        call = _lib.call
        p_scale = self.policy.action_scale.data_ptr()
        eps2 = ws.eps[1:1 + NP]
        ws.q5_bwd_done = False       # (a task may run the Q(s, pi) data backward inside one of its own launches)
        du2, du_ld = self.task.loss_and_backward(ws, P, lam_upd, assume_single)

        # the Q(s, pi) nets (dx only): single GPU — d min(Q1, Q2), policy_loss_1, the alpha losses and d log_alpha are
        # produced by this launch (nlbac_dy_head kind 3); data parallel — nlbac_actor_q_terms ran in part 1
        if ws.q5_bwd_done:
            pass
        elif self.world == 1 and self.fold_launches:
            call("nlbac_mlp_bwd_data_head", P.n_q5, P.io_q5, 2 * NP, B, C.byref(self._actor_q_head(ws, P, NP, G)), s)
        else:
            call("nlbac_mlp_bwd_data", P.n_q5, P.io_q5, 2 * NP, B, s)
        # the actors: d heads from d action (two Q nets + the rollout) and d logp, inside their data backward (kind 1)
        D = Do + A
        if not self.fold_launches:
            call("nlbac_gauss_sample_bwd", ws.heads2.data_ptr(), 2 * A, eps2.data_ptr(), p_scale, A,
                 NP * B, B, ws.dxq[0].data_ptr() + 4 * Do, D, ws.dxq[1].data_ptr() + 4 * Do, D, du2.data_ptr(), du_ld,
                 sc + 4 * SC.SC_ALPHA, 1.0 / G, ws.dheads2.data_ptr(), 2 * A, s)
            call("nlbac_mlp_bwd_data", P.n_act, P.io_act, NP, B, s)
        H = P.__dict__.get("head_gauss") if self.fold_launches else False
        if H is None:
            H = P.head_gauss = _lib.DyHead()
            H.kind, H.B_norm = 1, G
            H.heads, H.heads_ld, H.eps, H.scale, H.n_u = ws.heads2.data_ptr(), 2 * A, eps2.data_ptr(), p_scale, A
            H.da[0], H.da_ld[0] = ws.dxq[0].data_ptr() + 4 * Do, D
            H.da[1], H.da_ld[1] = ws.dxq[1].data_ptr() + 4 * Do, D
            H.alpha, H.dlogp_mul = sc + 4 * SC.SC_ALPHA, 1.0 / G
            H.dheads, H.dheads_ld = ws.dheads2.data_ptr(), 2 * A
            # the sums the td head and the actor-q head of this update left as tile partials (sums_defer): two workgroups
            # of this launch finish them — before the Adam step that reads d log_alpha and mirrors the losses
            J = 0
            for src in (P.__dict__.get("head_td"), P.__dict__.get("head_actor_q")):
                if src is None or not src.sums_defer:
                    continue
                F = H.finish[J]
                F.kind, F.partials, F.n_tiles = src.kind, src.partials, src.sums_tiles
                if src.kind == 2:
                    F.n_nets, F.mul, F.out, F.out_x = len(self.h_crit), src.mul, src.out, src.out_x
                else:
                    F.n_nets, F.B_norm = src.n_prob, src.B_norm
                    C.memmove(C.byref(F.actor), C.byref(src.actor), C.sizeof(_lib.ActorScalarArgs))
                J += 1
        if H is not False:
            # (per call: the step's lambda-update flags change from update to update) the augmented-Lagrangian step a
            # constraint head deferred (tasks.py: cf_job) is committed by this launch
            job = P.__dict__.get("cf_job")
            F = H.finish[2]
            F.kind = 4 if job else 0
            if job:
                F.partials, F.sc = job[4], job[3]        # (the stepped block the constraint backward staged)
        if H is not False:
            H.da[2], H.da_ld[2] = du2.data_ptr(), du_ld
            call("nlbac_mlp_bwd_data_head", P.n_act, P.io_act, NP, B, C.byref(H), s)
        tune = self.automatic_entropy_tuning
        p_part_q, n_part = ws.p_part_q, ws.n_part_q
        for g, cnt, nets, gio, sk_ws in P.act_groups:
            a = g.arena
            bwd_weights(nets, gio, cnt, B, a.n_slabs, a.n, self.device, ws=sk_ws)
            la = a.theta.data_ptr() + 4 * g.la_off

            def alpha_grads(p_grad, g=g, cnt=cnt, la=la):
                # policy_loss_1 / alpha losses from the (global) partial sums; d log_alpha goes straight into
                # the gradient the Adam step reads (it is already a global mean: it must not be all-reduced).
                # Single GPU: nlbac_actor_q_terms has done this in its own launch.
                if self.world > 1:
                    call("nlbac_actor_scalars", p_part_q, n_part, G, g.first, cnt, self.target_entropy, la, g.la_stride,
                         p_grad + 4 * g.la_off, sc, s)
                if not tune:
                    z = torch.zeros(1, device=self.device)
                    for off in [g.la_off + k * g.la_stride for k in range(cnt)]:
                        call("nlbac_axpby", 0.0, z.data_ptr(), 0.0, None, 1, p_grad + 4 * off, s)
            # (alpha = exp(log_alpha) is refreshed by the thread of the Adam step that moves log_alpha)
            refresh = ([g.la_off + k * g.la_stride for k in range(cnt)],
                       [sc + 4 * (SC.SC_ALPHA + g.first + k) for k in range(cnt)]) if tune else None
            last = g is P.act_groups[-1][0]
            mir = self.__dict__.get("_mirror") if last else None
            side = mir is not None and HOST_COPY == "side"
            self._adam(a, self.lr, a.n_slabs, before_step=alpha_grads, alpha=refresh, mirror=mir[1] if (mir and not side) else None)
            if side:
                # the scalars block reaches the host by a copy on a side stream: a kernel that writes host memory holds
                # the launch stream until the write has crossed PCIe (~5 us before the prefetched launches could start)
                st = self.__dict__.get("_sc_side")
                if st is None:
                    st = self._sc_side = (torch.cuda.Stream(device=self.device), torch.cuda.Event())
                st[1].record()
                st[0].wait_event(st[1])
                with torch.cuda.stream(st[0]):
                    mir[1].copy_(self.sc, non_blocking=True)
                    self._sc_ev[mir[0]].record()
                self._mirror_done = mir[0]
            elif mir:
                self._mirror_done = mir[0]
                self._sc_ev[mir[0]].record()     # (here, not in _returns: what _prefetch_next queues is not waited for)
        self._prefetch_next(ws, ws.updates_now)

    # ------------------------------------------------------------ checkpoints
    def save_model(self, output):
        print('Saving models in {}'.format(output))
        torch.save(self.policy.state_dict(), '{}/actor.pkl'.format(output))
        torch.save(self.critic.state_dict(), '{}/critic.pkl'.format(output))
        torch.save(self.lyapunovNet.state_dict(), '{}/lyapunov.pkl'.format(output))
        if self.BarrierNet is not None:
            torch.save(self.BarrierNet.state_dict(), '{}/barrier.pkl'.format(output))
        torch.save(self.neural_ode_model.state_dict(), '{}/node_model.pkl'.format(output))

    def load_weights(self, output):
        if output is None:
            return
        print('Loading models from {}'.format(output))
        dev = torch.device(self.device)
        self.policy.load_state_dict(torch.load('{}/actor.pkl'.format(output), map_location=dev, weights_only=True))
        self.critic.load_state_dict(torch.load('{}/critic.pkl'.format(output), map_location=dev, weights_only=True))
        self.lyapunovNet.load_state_dict(torch.load('{}/lyapunov.pkl'.format(output), map_location=dev, weights_only=True))
        if self.BarrierNet is not None:
            self.BarrierNet.load_state_dict(torch.load('{}/barrier.pkl'.format(output), map_location=dev, weights_only=True))
        self.repack_all()

    def load_model(self, actor_path, critic_path, lyapunov_path, barrier_path=None):
        if actor_path is not None:
            self.policy.load_state_dict(torch.load(actor_path, weights_only=True))
        if critic_path is not None:
            self.critic.load_state_dict(torch.load(critic_path, weights_only=True))
        if lyapunov_path is not None:
            self.lyapunovNet.load_state_dict(torch.load(lyapunov_path, weights_only=True))
        if barrier_path is not None and self.BarrierNet is not None:
            self.BarrierNet.load_state_dict(torch.load(barrier_path, weights_only=True))
        self.repack_all()


    # Full training state (row f4: what the reference's save_model omits — targets, optimiser moments and step counts,
    # multipliers, augmented terms, temperatures, the NODE): resume continues bit for bit.
    def save_checkpoint(self, path):
        names = ("ar_c", "ar_a", "ar_b", "ar_n")
        state = {"solver": self.solver, "sc": self.sc.cpu()}
        for n in names:
            a = getattr(self, n, None)
            if a is not None:
                state[n] = {k: getattr(a, k).cpu() for k in ("theta", "m", "v", "state")}
                if a.target is not None:
                    state[n]["target"] = a.target.cpu()
        torch.save(state, path)

    def load_checkpoint(self, path):
        state = torch.load(path, map_location="cpu", weights_only=True)
        for n, a_state in state.items():
            if n in ("solver", "sc"):
                continue
            a = getattr(self, n)
            for k, v in a_state.items():
                getattr(a, k).copy_(v)
        self.sc.copy_(state["sc"])
        self.solver = state["solver"]
        self.repack_all()


class _TargetView:
    """state_dict-style access to a target network stored in ``arena.target``."""

    def __init__(self, module, arena):
        self.module, self.arena = module, arena

    def state_dict(self):
        out = {}
        for k, p in self.module.named_parameters():
            off = self.arena.offset_of[id(p)]
            out[k] = self.arena.target[off:off + p.numel()].view(p.shape)
        return out

    def load_state_dict(self, sd):
        cur = self.state_dict()
        with torch.no_grad():
            for k, v in sd.items():
                cur[k].copy_(v)
