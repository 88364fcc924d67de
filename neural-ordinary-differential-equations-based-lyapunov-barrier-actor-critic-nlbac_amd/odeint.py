"""Batched IVP solve of the learned control-affine dynamics on ``t = [0, dt]``
and its exact (discretise-then-differentiate) backward — the device
replacement of the ``torchdiffeq.odeint`` call sites
``U/sac_cbf_clf/sac_cbf_clf.py:453,577`` and ``U/sac_cbf_clf/model.py:252``.

Solvers (same semantics as ``oracle/nlbac_oracle.odeint``):
  * ``euler``  one explicit Euler step over [0, dt] — the reference's setting
  * ``rk4``    one 3/8-rule step
  * ``dopri5`` Dormand–Prince 5(4), FSAL, one shared adaptive step per problem
               (RMS error norm over the whole (rows, n_s+n_u) tensor), result =
               4th-order interpolant at dt.  Step sizes carry no gradient.

Rows are ``P`` problems x ``rpp`` rows (e.g. primary and backup controller
actions on the same states).  Every stage costs one batched f_net/g_net
launch (``nlbac_mlp_fwd``) plus two per-row algebra kernels; the step-size
controller runs on the device and the host reads one 128-byte control block
per attempted step.
"""
import os
import ctypes as C

import torch

from . import _lib
from ._lib import fptr
from .arena import bwd_weights, io_array, mlp_array, stream_ptr

DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
DP_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
            -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.]

TABLEAU = {
    "euler": dict(beta=[], c_sol=[1.0]),
    "rk4": dict(beta=[[1 / 3], [-1 / 3, 1.0], [1.0, -1.0, 1.0]], c_sol=[1 / 8, 3 / 8, 3 / 8, 1 / 8]),
    "dopri5": dict(beta=DP_BETA, c_sol=None),
    # the initial-step probe of dopri5: f(y0 + h0 f0) as "stage 1" of a two-stage table (f0 = stage 0 is in place)
    "probe": dict(beta=[[1.0]], c_sol=None),
}


# How small results reach the host (the dopri5 control block here, the update's scalars block in sac_cbf_clf.py):
# "kernel" — the producing kernel writes pinned host memory itself; "side" — an async copy on a side stream behind an event.
HOST_COPY = os.environ.get("NLBAC_HOST_COPY", "kernel")
# How the host learns that a controller launch has left its decision in the pinned control block: by polling the block's
# stamp (a sequence lock the launch writes, nlbac_rk_chain::ctl_seq) or ("0") by an event behind the launch.
CTL_POLL = os.environ.get("NLBAC_CTL_POLL", "1") != "0"
_SEQ = [0]
NORM_DEFER_ATTEMPT = os.environ.get("NLBAC_NORM_DEFER_ATTEMPT", "1") != "0"

class _Carver:
    """Hands out the buffers of ONE step slot: consecutive 16-byte-aligned pieces of a flat float32 slice.  Every slot
    of a pool is carved by the same sequence of requests, so a buffer sits at the same offset in every slot — which is
    what lets the device-driven dopri5 chain address "the same buffer, k slots further" by pointer arithmetic
    (``nlbac_rk_chain.slot_floats``).  ``flat is None``: dry run, only counts."""

    def __init__(self, flat, device):
        self.flat, self.device, self.k = flat, device, 0

    def zeros(self, *shape, dtype=torch.float32):
        numel = 1
        for d in shape:
            numel *= d
        take = (numel + 3) & ~3
        off, self.k = self.k, self.k + take
        if self.flat is None:
            return torch.empty(0, dtype=dtype)
        assert self.k <= self.flat.numel(), "step slot too small"
        t = self.flat[off:off + numel]
        if dtype != torch.float32:
            t = t.view(dtype)
        return t.view(*shape)


class _SlotPool:
    """Step slots of one capacity bucket: chunks of ``slots_per_chunk`` slots, each chunk ONE allocation
    [slots][slot_floats].  The device-driven chain works inside chunk 0 (contiguous, ``n_slots`` = its size); the
    host-driven path just asks for the next slot and may spill into further chunks."""

    def __init__(self, solver, cap, S, n_slots, dry_only=False):
        self.solver, self.cap, self.S, self.n_slots = solver, cap, S, n_slots
        dry = _Carver(None, solver.device)
        solver.STEP_WS(solver, cap, S, dry).bwd(solver)
        self.slot_floats = (dry.k + 63) & ~63
        if dry_only:
            return
        self.chunks = [torch.zeros(n_slots, self.slot_floats, dtype=torch.float32, device=solver.device)]
        self.views = {}                  # (n, idx) -> step workspace

    def ws(self, n, idx):
        assert n <= self.cap
        w = self.views.get((n, idx))
        if w is None:
            c, i = divmod(idx, self.n_slots)
            while c >= len(self.chunks):
                self.chunks.append(torch.zeros(self.n_slots, self.slot_floats, dtype=torch.float32,
                                               device=self.solver.device))
            for k in [k for k in self.views if k[0] != n]:      # views laid out for another row count go
                del self.views[k]
            w = self.views[(n, idx)] = self.solver.STEP_WS(self.solver, n, self.S, _Carver(self.chunks[c][i], self.solver.device))
            w.slot, w.pool = idx, self
        return w


class _StepWS:
    """Device buffers of one RK step for n rows (stage-major)."""
    # what a per-problem solver takes over from a joint first attempt: (buffer, leading blocks per row range)
    ADOPT = ("K", "Y", "gout", "err", "acts_f", "acts_g")

    def __init__(self, solver, n, S, store):
        dev, ns, nu = solver.device, solver.n_s, solver.n_u
        f, g = solver.f, solver.g
        self.n, self.S = n, S
        self._store = store
        z = self._store.zeros
        self.K = z(S, n, ns)
        self.Y = z(S, n, ns)
        self.fout = z(n, ns)
        self.gout = z(S, n, ns * nu)
        # a rollout that is only differentiated w.r.t. its inputs keeps bit-packed ReLU masks (one uint32 per 32
        # hidden units) instead of the activations: 1/32 of the HBM traffic of the fused step kernels
        self.bits = bool(solver.fused and not solver.keep_acts)
        if self.bits:
            zi = lambda *s: self._store.zeros(*s, dtype=torch.int32)
            # words per row and layer: one per 32 hidden units, or (register-resident kernels) one per lane quarter
            lib = _lib.load()
            self.wf = lib.nlbac_node_rk_mask_words(C.byref(f.desc), C.byref(g.desc), 0)
            self.wg = lib.nlbac_node_rk_mask_words(C.byref(f.desc), C.byref(g.desc), 1)
            self.acts_f = zi(f.n_layers - 1, S * n, self.wf)
            self.acts_g = zi(g.n_layers - 1, S * n, self.wg)
        else:
            self.wf, self.wg = f.hid, g.hid
            self.acts_f = z(f.n_layers - 1, S * n, f.hid)
            self.acts_g = z(g.n_layers - 1, S * n, g.hid)
        self.y1 = z(n, ns)
        self.err = z(n, ns)
        self._bwd = None
        self.io_fwd, self.io_bwd = {}, {}

    def bwd(self, solver):
        if self._bwd is None:
            dev, ns, nu, n, S = solver.device, solver.n_s, solver.n_u, self.n, self.S
            f, g = solver.f, solver.g
            z = self._store.zeros
            self.dK = z(S, n, ns)
            keep = solver.keep_acts or not solver.fused      # weight gradients / the stage-by-stage path need dz
            self.dG = z(S, n, ns * nu) if keep else None
            self.dz_f = z(f.n_layers - 1, S * n, f.hid) if keep else None
            self.dz_g = z(g.n_layers - 1, S * n, g.hid) if keep else None
            self.dXf = z(n, f.in_dim)
            self.dXg = z(n, g.in_dim)
            self.dy0 = z(n, ns)
            self.dy1 = z(n, ns)
            self._bwd = True
        return self


class AffineNodeSolver:
    """odeint for ``dx/dt = f(x) + g(x) u`` with u constant over the step."""
    STEP_WS = _StepWS

    def __init__(self, node, device):
        self.node, self.f, self.g = node, node.f, node.g
        self.n_s, self.n_u = node.n_s, node.n_u
        self.device = torch.device(device)
        self._ws = {}          # (n, S, idx) -> _StepWS
        self._scratch = {}
        self.nfe = 0
        self._net_arr = None
        self._coefs = {}
        # one nlbac_node_rk_fwd / _bwd launch per RK step instead of 3 launches per stage — when the fused backward's
        # LDS carve fits (4 tiles + both nets' first / last layers: up to 232-wide nets); wider ones run stage by stage
        self.fused = self._fused_fits(node.f, node.g)
        self.keep_acts = True  # False: backward never asks for weight gradients -> ReLU bit masks suffice
        self._children = {}    # per-problem solvers for batches whose problems diverge (dopri5)
        self.stats = dict(solves=0, single_step=0, multi_attempt=0, split=0)
        self.comm = None       # nlbac_amd.parallel.DataParallel: global dopri5 error norms
        # > 1: every problem of a solve is cut into this many contiguous row groups, each an adaptive solve of its own
        # (own error norm, step sizes and accept decisions) — what a sample-sharded run with per-shard step control
        # (SAC_CBF_CLF.enable_data_parallel(step_control="shard")) computes on that many ranks, on one device
        self.row_groups = 1
        self.adjoint = False   # True: ``backward`` is the continuous adjoint (odeint_adjoint); the forward keeps nothing
        self.generation = 0    # bumped whenever device buffers are freed or re-laid-out (owners of hipGraphs watch it)
        self.device_loop = True   # dopri5 attempts as a device-driven chain (no host decision per attempt)

    @staticmethod
    def _fused_fits(f, g):
        pad = lambda x, m: (x + m - 1) // m * m
        ld = pad(max(f.hid, g.hid), 32) + 4
        sw = pad((f.out_dim + f.in_dim) * f.hid, 4) + pad((g.out_dim + g.in_dim) * g.hid, 4)
        lds = 4 * (4 * 32 * ld + 8 * 32 * 8 + 32 * (4 + 1 + 8 + 4 + 2 * 8 + 2 * 16) + sw)      # nlbac_node_rk_bwd's carve
        return lds <= 160 * 1024 - 64

    # -- workspace -----------------------------------------------------------
    MAX_SIZES = 2      # distinct row counts whose buffers are kept (the NODE fit's batch grows with the replay)

    def _touch(self, n):
        """Start of a solve on n rows: make n the current size and drop the scratch of the least recently used sizes
        beyond ``MAX_SIZES`` — a training run feeds the NODE fit min(replay size, 32768) rows, a new count at every fit
        while the replay fills.  (Step slots live in per-capacity pools, see ``_pool``.)"""
        order = self.__dict__.setdefault("_n_order", [])
        if n in order:
            order.remove(n)
        order.append(n)
        while len(order) > self.MAX_SIZES:
            old = order.pop(0)
            self._scratch.pop(old, None)
            self.generation += 1
        self._cur_n = n

    @staticmethod
    def _bucket(n):
        return n if n <= 4096 else -(-n // 4096) * 4096

    DEFAULT_SLOTS = 4      # step slots of a pool's first chunk = steps the device-driven chain can accept without a restart

    def _pool(self, n, S, min_slots=1):
        """The slot pool serving n rows / S stages (one per capacity bucket and solver mode; the two most recently
        used buckets are kept).  ``generation`` counts every event that frees or re-lays-out device buffers — captured
        hipGraphs bake their addresses in and are dropped by their owners when it moves."""
        pools = self.__dict__.setdefault("_pools", {})
        key = (self._bucket(n), S, self.fused, self.keep_acts)
        pool = pools.get(key)
        dropped = False
        if pool is not None and pool.n_slots < min_slots:
            del pools[key]
            pool, dropped = None, True
        if pool is None:
            buckets = list(dict.fromkeys(k[0] for k in pools))              # in order of first use
            if key[0] not in buckets and len(buckets) >= self.MAX_SIZES:
                for k in [k for k in pools if k[0] == buckets[0]]:         # the oldest bucket goes, whole
                    del pools[k]
                dropped = True
            if dropped:
                self.generation += 1
                # pools are GB-sized and each regrowth asks for a new size: hand the freed blocks back to the driver,
                # or the caching allocator keeps every size it has ever seen (279 GiB reserved for 39 GiB in use in a
                # long dopri5 training run before this)
                if not torch.cuda.is_current_stream_capturing():
                    torch.cuda.empty_cache()
            n_slots = max(min_slots, self.DEFAULT_SLOTS if S == 7 else 1)
            pool = _SlotPool(self, key[0], S, n_slots, dry_only=True)
            if pool.slot_floats * 4 * n_slots > self.MAX_POOL_BYTES:
                raise _lib.NlbacError(
                    "dopri5: %d accepted steps of %d rows need %.0f GiB of step slots (limit %.0f GiB): the field has "
                    "become stiff for back-propagation through the steps — use the adjoint (agent.adjoint = True / "
                    "odeint_adjoint), whose memory does not grow with the step count"
                    % (n_slots, key[0], pool.slot_floats * 4 * n_slots / 2 ** 30, self.MAX_POOL_BYTES / 2 ** 30))
            pool = pools[key] = _SlotPool(self, key[0], S, n_slots)
        return pool

    MAX_POOL_BYTES = 128 * 2 ** 30

    def _step_ws(self, n, S, idx):
        pool = self._pool(n, S)
        had = (n, idx) in pool.views
        ws = pool.ws(n, idx)
        if not had and any(k[0] != n for k in pool.views):
            self.generation += 1
        return ws

    def reserve(self, n, P, method, steps=2):
        """Allocate the buffers of a solve on n rows / P problems ahead of time (the slot pool and, for a host-driven
        multi-problem dopri5 solve, the per-problem fallback solvers), so that the first multi-step or diverging solve
        of a run does not pay tens of milliseconds of allocation in the middle of training."""
        self._touch(n)
        P *= self.row_groups
        if method != "dopri5":
            S = len(TABLEAU[method]["c_sol"])
            self._step_ws(n, S, 0).bwd(self)
            return
        for idx in range(steps):
            self._step_ws(n, 7, idx).bwd(self)
        self._ctl_io(P)
        if P > 1 and not self._chain_ok(P, n // P):
            for p in range(P):
                if p not in self._children:
                    self._children[p] = type(self)(self.node, self.device)
                k = self._children[p]
                k.comm, k.fused, k.keep_acts = self.comm, self.fused, self.keep_acts
                # the per-problem solvers run one after the other inside this solver's solve: they share its read-back
                # stream and pinned blocks (a pinned allocation costs milliseconds)
                self._ctl_io(1)
                k._side, k._ev_ctl, k._ctl_pin = self._side, self._ev_ctl, self._ctl_pin
                k.before_wait = self.__dict__.get("before_wait")
                k.reserve(n // P, 1, method, steps)

    def _out_buf(self, n, fallback=None):
        """Where the solve's result goes: the caller's tensor (``out_into``, when it has the solve's shape — the next
        launches read it there, no copy) or a solver-owned buffer."""
        t = self.__dict__.get("out_into")
        if t is not None and tuple(t.shape) == (n, self.n_s) and t.is_contiguous():
            return t
        return fallback if fallback is not None else self._buf("dopri_out", n, self.n_s)

    def _buf(self, name, *shape, dtype=torch.float32):
        """Named scratch buffer of the current solve size (dropped with that size's workspaces, see ``_touch``)."""
        pool = self._scratch.setdefault(self.__dict__.get("_cur_n", 0), {})
        key = (name, shape, dtype)
        if key not in pool:
            pool[key] = torch.zeros(*shape, dtype=dtype, device=self.device)
        return pool[key]

    # -- one field evaluation k = f(x) + g(x) u -----------------------------------
    def _nets(self):
        if self._net_arr is None:
            self._net_arr = mlp_array([self.f.desc, self.g.desc])
        return self._net_arr

    def _eval_io(self, x, g_out, acts_f=None, acts_g=None, ls_f=0, ls_g=0):
        io = io_array(2)
        ns, nu = self.n_s, self.n_u
        fout = self._buf("fout", x.shape[0], ns)
        for i, (y, ld, acts, ls) in enumerate(((fout, ns, acts_f, ls_f), (g_out, ns * nu, acts_g, ls_g))):
            io[i].x0, io[i].x0_dim, io[i].x0_ld = x.data_ptr(), ns, ns
            io[i].y, io[i].y_ld = y.data_ptr(), ld
            if acts is not None:
                io[i].acts, io[i].acts_ls = acts.data_ptr(), ls
        return io, fout

    def _eval(self, x, u, n, k_out, g_out, io=None):
        if io is None:
            io = self._eval_io(x, g_out)
        io, fout = io
        s = stream_ptr()
        _lib.call("nlbac_mlp_fwd", self._nets(), io, 2, n, s)
        _lib.call("nlbac_affine_combine_fwd", fout.data_ptr(), g_out.data_ptr(), u.data_ptr(), self.n_s, self.n_u, n,
                  k_out.data_ptr(), s)
        self.nfe += 1

    def _probe_eval(self, ytmp, u, n, ktmp, gtmp):
        """field evaluation outside the step workspaces (dopri5 initial-step probe), nothing saved"""
        pool = self._scratch.setdefault(self.__dict__.get("_cur_n", 0), {})
        tio = pool.get(("tmp_io", n))
        if tio is None:
            tio = pool[("tmp_io", n)] = self._eval_io(ytmp, gtmp)
        self._eval(ytmp, u, n, ktmp, gtmp, tio)

    def _stage_eval(self, ws, st, u):
        n, S = ws.n, ws.S
        io = ws.io_fwd.get(st)
        if io is None:       # pointers are static per (workspace, stage): build the descriptors once
            f, g = self.f, self.g
            io = ws.io_fwd[st] = self._eval_io(ws.Y[st], ws.gout[st], ws.acts_f[:, st * n:], ws.acts_g[:, st * n:],
                                               S * n * f.hid, S * n * g.hid)
        self._eval(ws.Y[st], u, n, ws.K[st], ws.gout[st], io)

    def _beta(self, method):
        key = ("beta", method)
        b = self._coefs.get(key)
        if b is None:
            rows = TABLEAU[method]["beta"]
            S = len(rows) + 1
            flat = [0.0] * (S * S)
            for i, r in enumerate(rows):
                for j, v in enumerate(r):
                    flat[(i + 1) * S + j] = v
            b = self._coefs[key] = (fptr(*flat), S)
        return b

    def _rk_fused(self, ws, y0, u, P, rpp, method, st0, st1, h_host=None, h_dev=None, c_out=None, out=None,
                  c_err=None, err=None, save_acts=True, chain=None):
        """One launch for stages [st0, st1) of ``method`` on the step workspace ``ws`` (nlbac_node_rk_fwd)."""
        beta, S = self._beta(method)
        f, g = self.f, self.g
        n = P * rpp
        save_acts = save_acts and not self.adjoint       # (the adjoint re-computes every stage it differentiates)
        im = None
        if st0 == 0 and self.ctx.pop("in_map_pending", None):
            # first launch of the solve: it forms the initial state itself (set_in_map) and leaves it in y0
            m, im = self._in_map, _lib.InMap()
            im.kind, im.obs, im.obs_ld, im.l = m["kind"], m["obs"].data_ptr(), m["obs_ld"], m["l"]
            im.ps = m["ps"].data_ptr() if m["ps"] is not None else None
        _lib.call("nlbac_node_rk_fwd", C.byref(f.desc), C.byref(g.desc), y0.data_ptr(), u.data_ptr(), P, rpp,
                  st0, st1, S, beta, c_out, len(c_out) if c_out is not None else 0,
                  c_err, len(c_err) if c_err is not None else 0,
                  fptr(*h_host) if h_host is not None else None, h_dev, _lib.DOPRI_CTL if h_dev else 0,
                  ws.K.data_ptr(), ws.Y.data_ptr(), ws.gout.data_ptr(),
                  ws.acts_f.data_ptr() if save_acts else None, ws.S * n * ws.wf,
                  ws.acts_g.data_ptr() if save_acts else None, ws.S * n * ws.wg, 1 if ws.bits else 0,
                  out.data_ptr() if out is not None else None, err.data_ptr() if err is not None else None,
                  C.byref(chain) if chain is not None else None, C.byref(im) if im is not None else None, stream_ptr())
        self.nfe += st1 - st0

    def _rk_fused_bwd(self, ws, u, P, rpp, method, first_eval, need_dy0, need_params, h_host, h_dev, h_stride, top_up,
                      du, last, chain=None, back_idx=0):
        """One launch for the backward of every evaluated stage of the step in ``ws`` (nlbac_node_rk_bwd)."""
        beta_arr, _ = self._beta(method)
        f, g, S = self.f, self.g, ws.S
        _lib.call("nlbac_node_rk_bwd", C.byref(f.desc), C.byref(g.desc), u.data_ptr(), ws.gout.data_ptr(), P, rpp,
                  S, 0 if first_eval else 1, S, 1 if need_dy0 else 0, beta_arr, h_host, h_dev, h_stride,
                  ws.acts_f.data_ptr(), S * ws.n * ws.wf, ws.acts_g.data_ptr(), S * ws.n * ws.wg,
                  1 if ws.bits else 0, ws.dz_f.data_ptr() if need_params else None,
                  ws.dz_g.data_ptr() if need_params else None, ws.dG.data_ptr() if need_params else None,
                  ws.dK.data_ptr(), top_up.data_ptr() if top_up is not None else None, ws.dy0.data_ptr(), 1,
                  du.data_ptr() if du is not None else None, 0 if last else 1,
                  C.byref(chain) if chain is not None else None, back_idx, stream_ptr())

    def _combine(self, y0, K, n_k, coef, h, P, rpp, out):
        _lib.call("nlbac_rk_combine", y0.data_ptr() if y0 is not None else None, K.data_ptr(), n_k,
                  fptr(*coef), fptr(*h), None, 0, P, rpp, self.n_s, out.data_ptr(), stream_ptr())

    # -- forward ---------------------------------------------------------------
    def forward(self, y0, u, P, rpp, method, dt, atol=1e-7, rtol=1e-5):
        """y0: (P*rpp, n_s), u: (P*rpp, n_u) contiguous device tensors.
        Returns x(dt) (P*rpp, n_s) (a solver-owned buffer, valid until the next call)."""
        self.forward_begin(y0, u, P, rpp, method, dt, atol, rtol)
        return self.forward_finish()

    def forward_begin(self, y0, u, P, rpp, method, dt, atol=1e-7, rtol=1e-5):
        """Enqueue everything up to the first point where the host has to look at a result (dopri5: the
        accept/reject decision of the first attempted step); euler/rk4 run to completion.  No host sync."""
        n = P * rpp
        assert y0.shape == (n, self.n_s) and u.shape == (n, self.n_u)
        if self.row_groups > 1:
            assert rpp % self.row_groups == 0, "row_groups must divide the rows of a problem"
            P, rpp = P * self.row_groups, rpp // self.row_groups
        self._touch(n)
        self.stats["solves"] += 1
        self.ctx = dict(method=method, P=P, rpp=rpp, n=n, u=u, y0=y0, steps=[], t_end=float(dt), atol=atol,
                        rtol=rtol)
        if self.__dict__.pop("_in_map_armed", False):
            self.ctx["in_map_pending"] = True          # (consumed by the solve's first launch, _rk_fused)
        if method in ("euler", "rk4"):
            tab = TABLEAU[method]
            S = len(tab["c_sol"])
            ws = self._step_ws(n, S, 0)
            h = [float(dt)] * P
            if self.fused:
                out = self._out_buf(n, ws.y1)
                self._rk_fused(ws, y0, u, P, rpp, method, 0, S, h_host=h, c_out=fptr(*tab["c_sol"]), out=out)
                self.ctx["steps"].append(dict(ws=ws, h=h, first=True))
                self.ctx["out"] = out
                return
            ws.Y[0].copy_(y0)
            for st in range(S):
                self._stage_eval(ws, st, u)
                if st + 1 < S:
                    self._combine(y0, ws.K, st + 1, tab["beta"][st], h, P, rpp, ws.Y[st + 1])
            self._combine(y0, ws.K, S, tab["c_sol"], h, P, rpp, ws.y1)
            self.ctx["steps"].append(dict(ws=ws, h=h, first=True))
            self.ctx["out"] = ws.y1
        elif method == "dopri5":
            self._dopri_begin(y0, u, P, rpp)
        else:
            raise ValueError("unknown solver %r" % (method,))

    def forward_finish(self, assume_single_step=False):
        """Complete the solve and return x(dt).  dopri5: reads the 128-byte/problem control block (one host
        sync) and continues with further attempts if the first step was rejected or stopped short of dt.
        ``assume_single_step``: skip the read (the caller has already checked ``first_step_done``) — this
        variant enqueues only device work with device-resident step size, so it can be hipGraph-captured."""
        if self.ctx["method"] != "dopri5":
            return self.ctx["out"]
        if self.ctx.get("chain"):
            return self._dopri_finish_chain(assume_done=assume_single_step)
        if assume_single_step:
            return self._dopri_accept_first(None)
        return self._dopri_continue()

    def first_step_done(self):
        """dopri5, after forward_begin: True iff every problem accepted its first step and reached dt
        (the overwhelmingly common case at dt=0.02).  One small D2H read."""
        P = self.ctx["P"]
        c = self._ctl_read(P)
        self.ctx["ctl_host"] = c
        if self.ctx.get("chain"):       # device-driven chain: every problem finished within the attempts enqueued
            cl = c.tolist()             # (plain floats: element-wise reads of a tensor cost microseconds each, on the host's
            #                              way from the accept decision to the launches that wait for it)
            ok = all(cl[p][4] > 0 and not cl[p][13] > 0 for p in range(P))
            if not ok:                  # a captured chain that is too short: later captures enqueue more attempts
                self._chain_len = int(max(float(c[p, 10]) for p in range(P))) + 1
                self.generation += 1
            return ok
        return all(bool(c[p, 3] > 0) and bool(c[p, 4] > 0) for p in range(P))

    def _ctl(self, P):
        return self._buf("ctl", P, _lib.DOPRI_CTL, dtype=torch.float64)

    def _ctl_post(self, P, src=None):
        """After an attempted step: send the control block to pinned host memory on a side stream, so that the
        host can read the accept decision as soon as the controller has run — without draining the launch
        stream, on which the caller may have queued independent work behind the attempt (the agent queues its
        whole critic phase there).  Not inside a hipGraph capture (the replay path reads with ``_ctl(P).cpu()``)."""
        if torch.cuda.is_current_stream_capturing():
            return
        side, pin = self._ctl_io(P)
        self.ctx["ctl_seq"] = None        # (a copy, read behind its event: nothing to poll)
        ev_a, ev_b = self._ev_ctl
        ev_a.record()
        side.wait_event(ev_a)
        with torch.cuda.stream(side):
            pin.copy_(self._ctl(P) if src is None else src, non_blocking=True)
            ev_b.record()
        self.ctx["ctl_pending"] = P

    def _ctl_posted(self, P):
        """Device-driven chain: the controller launches have written the host's copy themselves (``ctl_host``); mark the
        point on the launch stream behind which it is complete."""
        if torch.cuda.is_current_stream_capturing():
            return
        self._ctl_io(P)
        if self.ctx.get("ctl_seq") is None:
            self._ev_ctl[1].record()     # (stamped blocks are polled: no event, no marker on the launch stream)
        self.ctx["ctl_pending"] = P

    def _seq_next(self, first_of_solve=False):
        """Stamp for the next controller launch that writes the host's copy (``CTL_POLL``); ctx["ctl_seq"] = (stamp of the
        solve's first such launch, stamp of its latest).  None when the block is read behind an event."""
        if not CTL_POLL or torch.cuda.is_current_stream_capturing() or HOST_COPY == "side":
            self.ctx["ctl_seq"] = None
            return 0.0
        _SEQ[0] += 1                 # (one counter per process: solvers share pinned blocks, see _solve_split)
        rng = self.ctx.get("ctl_seq")
        self.ctx["ctl_seq"] = (_SEQ[0] if (first_of_solve or rng is None) else rng[0], _SEQ[0])
        return float(_SEQ[0])

    def _ctl_poll(self, P, first, last, patience=0.05):
        """Wait for the stamped control blocks of the launch with stamp ``last``: a problem's block is complete when it
        carries that stamp — or an earlier one of the same solve with the done flag (launches skip finished problems).
        Sequence-lock read: stamp, block, stamp.  After ``patience`` seconds of spinning the launch stream is drained
        (everything queued has then run) and the block must be there."""
        import time
        arr = self._ctl_pin[P].numpy()          # (the same memory)
        t0 = drained = None
        n = 0
        stamps = arr[:, 15]
        while True:
            s1 = stamps.tolist()                       # (stamp, block, stamp: the sequence lock's read side)
            if all(x == last or first <= x < last for x in s1):
                c = arr.copy()
                if all(c[p, 15] == s1[p] and (s1[p] == last or c[p, 4] > 0) for p in range(P)):
                    return torch.from_numpy(c)
            n += 1
            if n & 63 == 0:
                now = time.perf_counter()
                if t0 is None:
                    t0 = now
                elif drained:
                    raise _lib.NlbacError("control block %r never reached stamp %d (solve from %d)" % (s1, last, first))
                elif now - t0 > patience:
                    torch.cuda.current_stream().synchronize()
                    drained = True
                    self.stats["poll_drained"] = self.stats.get("poll_drained", 0) + 1

    def _ctl_io(self, P):
        """(side stream, pinned block for P problems) of the control-block read-back, created on first use."""
        if self.__dict__.get("_side") is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._ev_ctl = (torch.cuda.Event(), torch.cuda.Event())
            self._ctl_pin = {}
        pin = self._ctl_pin.get(P)
        if pin is None:
            pin = self._ctl_pin[P] = torch.zeros(P, _lib.DOPRI_CTL, dtype=torch.float64).pin_memory()
        return self._side, pin

    def _ctl_read(self, P):
        """Host copy of the control block of the last attempted step."""
        if self.ctx.pop("ctl_pending", None) == P:
            hook = self.__dict__.get("before_wait")
            if hook is not None:
                hook()                   # (the owner queues independent work behind the attempt before the host blocks)
            rng = self.ctx.get("ctl_seq")
            if rng is not None:
                return self._ctl_poll(P, *rng)
            self._ev_ctl[1].synchronize()
            return self._ctl_pin[P].clone()
        return self._ctl(P).cpu()

    def _norm_control(self, a, b, y0, y1, u, mode, P, rpp, slot_ctl=None, slot_floats=0, chain=None):
        """Scaled RMS norm(s) of mode 0/1/2 (include/nlbac_hip.h) over each problem's rows, then the step-size
        controller: one launch on a single GPU, norm -> all-reduce -> controller under data parallelism."""
        ctx = self.ctx
        ns, nu, s = self.n_s, self.n_u, stream_ptr()
        nblk = (rpp + 255) // 256
        part = self._buf("part", P, nblk, 2)
        ctl = self._ctl(P)
        dp = lambda t: t.data_ptr() if t is not None else None
        if self.comm is not None and self.comm.world > 1:
            _lib.call("nlbac_dopri_norm_partials", dp(a), dp(b), dp(y0), dp(y1), dp(u), mode, ctx["rtol"], ctx["atol"],
                      ns, nu, rpp, P, part.data_ptr(), slot_ctl, slot_floats, s)
            self._control(part, nblk, mode, P, rpp, ctx["t_end"], ctl, chain)
            return
        tickets = self._buf("tickets", P, dtype=torch.int32)
        _lib.call("nlbac_dopri_norm_control", dp(a), dp(b), dp(y0), dp(y1), dp(u), mode, ctx["rtol"], ctx["atol"],
                  ns, nu, rpp, P, ctx["t_end"], part.data_ptr(), tickets.data_ptr(), ctl.data_ptr(),
                  C.byref(chain) if chain is not None else None, s)

    def _control(self, part, nblk, mode, P, rpp, t_end, ctl, chain=None):
        """Step-size controller; under data parallelism the squared-norm sums are all-reduced first so every
        rank takes the decision the single-device run over the global batch would take."""
        ns, nu, s = self.n_s, self.n_u, stream_ptr()
        if self.comm is not None and self.comm.world > 1:
            sums = self._buf("psum", P, 1, 2)
            for p in range(P):
                _lib.call("nlbac_sum_partials", part[p].data_ptr(), nblk, 2, 1.0, sums[p].data_ptr(), s)
            self.comm.all_reduce_(sums)
            tail = (chain.n_slots, chain.hslots, chain.alog, chain.alog_cap) if chain is not None else (0, None, None, 0)
            _lib.call("nlbac_dopri_control", sums.data_ptr(), 1, mode, ns, nu, rpp * self.comm.world, P, t_end,
                      ctl.data_ptr(), *tail, s)
        else:
            _lib.call("nlbac_dopri_control", part.data_ptr(), nblk, mode, ns, nu, rpp, P, t_end, ctl.data_ptr(), 0, None,
                      None, 0, s)

    def _dopri_attempt(self, ws, cur_y0, u, P, rpp):
        """Stages 1..6 of one attempted step, error estimate, norm and controller (all on the device)."""
        ns, nu, S = self.n_s, self.n_u, 7
        s = stream_ptr()
        ctx = self.ctx
        nblk = (rpp + 255) // 256
        part = self._buf("part", P, nblk, 2)
        ctl = self._ctl(P)
        h_dev = ctl.data_ptr()                    # C_H
        if self.fused:
            self._rk_fused(ws, cur_y0, u, P, rpp, "dopri5", 1, S, h_dev=h_dev, c_err=self._coef("err"), err=ws.err)
        else:
            for st in range(1, S):
                _lib.call("nlbac_rk_combine", cur_y0.data_ptr(), ws.K.data_ptr(), st, self._coef(("b", st)), None,
                          h_dev, _lib.DOPRI_CTL, P, rpp, ns, ws.Y[st].data_ptr(), s)
                self._stage_eval(ws, st, u)
            _lib.call("nlbac_rk_combine", None, ws.K.data_ptr(), S, self._coef("err"), None, h_dev, _lib.DOPRI_CTL,
                      P, rpp, ns, ws.err.data_ptr(), s)
        self._norm_control(ws.err, None, cur_y0, ws.Y[6], None, 2, P, rpp)
        self._ctl_post(P)

    def _coef(self, key):
        c = self._coefs.get(key)
        if c is None:
            vals = DP_C_ERR if key == "err" else ([1.0] if key == "one" else (DP_BETA[5] + [0.0] if key == "sol" else
                                                                              DP_BETA[key[1] - 1]))
            c = self._coefs[key] = fptr(*vals)
        return c

    def _dopri_begin(self, y0, u, P, rpp):
        if self._chain_ok(P, rpp):
            return self._dopri_begin_chain(y0, u, P, rpp)
        n, ns, nu, S = P * rpp, self.n_s, self.n_u, 7
        ctx = self.ctx
        s = stream_ptr()
        nblk = (rpp + 255) // 256
        part = self._buf("part", P, nblk, 2)
        ctl = self._ctl(P)
        ws = self._step_ws(n, S, 0)
        rtol, atol, t_end = ctx["rtol"], ctx["atol"], ctx["t_end"]
        # f0 and the initial step size (Hairer's rule)
        if self.fused:
            self._rk_fused(ws, y0, u, P, rpp, "dopri5", 0, 1, h_dev=ctl.data_ptr())
        else:
            ws.Y[0].copy_(y0)
            self._stage_eval(ws, 0, u)
        self._norm_control(ws.K[0], None, y0, None, u, 0, P, rpp)
        h0_dev = ctl.data_ptr() + 6 * 8           # C_H0
        if self.fused:
            # probe f(y0 + h0 f0): the fused kernel forms the stage input itself (same arithmetic as nlbac_rk_combine);
            # K[1] / Y[1] / gout[1] of the step workspace are scratch until the real stage 1 overwrites them
            self._rk_fused(ws, y0, u, P, rpp, "probe", 1, 2, h_dev=h0_dev, save_acts=False)
            ktmp = ws.K[1]
        else:
            ytmp, ktmp, gtmp = self._buf("ytmp", n, ns), self._buf("ktmp", n, ns), self._buf("gtmp", n, ns * nu)
            _lib.call("nlbac_rk_combine", y0.data_ptr(), ws.K.data_ptr(), 1, self._coef("one"), None, h0_dev,
                      _lib.DOPRI_CTL, P, rpp, ns, ytmp.data_ptr(), s)
            self._probe_eval(ytmp, u, n, ktmp, gtmp)
        self._norm_control(ktmp, ws.K[0], y0, None, None, 1, P, rpp)
        self._dopri_attempt(ws, y0, u, P, rpp)

    # -- dopri5 as a device-driven chain ------------------------------------------------------------------------
    # An attempted step is ONE launch: nlbac_node_rk_fwd with an nlbac_rk_chain description evaluates stages 1-6 in
    # the step slot the control block names, forms the error norm and runs the controller in its own epilogue.  The
    # host enqueues a fixed number of attempts (kernels skip problems that have finished) and looks at the control
    # block once per chain — not once per attempt; problems of one batch advance independently, so there is no
    # per-problem fallback on this path.  The backward walks the slots the same way (``back_idx``).
    ALOG_CAP = 64
    FUSED_NORM_MODES = (0, 1)

    # -- input map (nlbac_in_map) --------------------------------------------------------------------------------
    def set_in_map(self, kind, obs, obs_ld, l, ps=None):
        """The NEXT ``forward_begin``'s ``y0`` is an output: the solve's first launch forms the initial state from the
        owner's observation rows (kind 1: the Unicycle tasks' state map, ``ps`` receives the state's look-ahead point)
        and leaves it there.  Fused control-affine solver only (``fused``)."""
        assert self.fused, "set_in_map: the stage-by-stage path reads y0"
        self._in_map = dict(kind=kind, obs=obs, obs_ld=int(obs_ld), l=float(l), ps=ps)
        self._in_map_armed = True

    # -- output map (nlbac_out_map) ------------------------------------------------------------------------------
    def set_out_map(self, kind, l, p, dp=None, dp2=None):
        """The owner's per-row map of the solve's output (kind 1: planar look-ahead point, ``p`` (n, 2) receives it,
        ``dp`` / ``dp2`` hold its gradient at backward time).  Evaluated inside the dopri5 interpolation launches of the
        device-driven chain; ``out_mapped`` / ``backward(None)`` tell / let the owner skip its own launches.  Anywhere
        else (fixed-step methods, host-driven steps, the adjoint) the map is not applied and the owner launches it."""
        self._out_map = dict(kind=kind, l=float(l), p=p, dp=dp, dp2=dp2)

    @property
    def out_mapped(self):
        return bool(self.ctx.get("out_mapped"))

    def _out_map_fwd(self, n):
        m = self.__dict__.get("_out_map")
        if m is None or m["p"].shape[0] != n or self.adjoint:
            return None
        om = _lib.OutMap()
        om.kind, om.l, om.p = m["kind"], m["l"], m["p"].data_ptr()
        return om

    def _out_map_bwd(self, n):
        m = self.__dict__.get("_out_map")
        if m is None or m["dp"] is None or not self.ctx.get("out_mapped"):
            return None
        om = _lib.OutMap()
        om.kind, om.l, om.dp = m["kind"], m["l"], m["dp"].data_ptr()
        om.dp2 = m["dp2"].data_ptr() if m["dp2"] is not None else None
        om.x = self.ctx["out"].data_ptr()
        return om

    def _interp_nets(self):
        return C.byref(self.f.desc), C.byref(self.g.desc)

    def _interp_fold(self):
        """The interpolation at t_end is evaluated by the attempt launches themselves and its backward by the last step's
        backward launch (nlbac_rk_chain::interp_*; the register-resident kernels): no nlbac_dopri_interp_fwd / _bwd
        launches.  ``interp_fold = False`` (NLBAC_INTERP_FOLD=0) keeps the two launches — the cross-check."""
        on = self._interp_fold_on()
        ok = self.__dict__.get("_interp_ok")
        if ok is None:
            ok = self._interp_ok = bool(_lib.load().nlbac_rk_interp_ok(*self._interp_nets()))
        return on and ok and self.fused

    def _chain_ok(self, P, rpp):
        return bool(self.device_loop and self.fused and (P == 1 or rpp % _lib.MLP_TILE == 0))

    def _chain(self, ws0, pool, P, rpp, norm_mode, read_ctl):
        ctx = self.ctx
        ctl = self._ctl(P)
        nblk = (rpp + _lib.MLP_TILE - 1) // _lib.MLP_TILE
        hs = self._buf("hslots%d" % pool.n_slots, P, pool.n_slots, dtype=torch.float64)
        c = _lib.RkChain()
        c.ctl = ctl.data_ptr() if read_ctl else None
        c.slot_floats, c.n_slots = pool.slot_floats, pool.n_slots
        c.rtol, c.atol, c.t_end = ctx["rtol"], ctx["atol"], ctx["t_end"]
        c.ctl_w, c.hslots = ctl.data_ptr(), hs.data_ptr()
        c.alog, c.alog_cap = self._buf("alog", P, self.ALOG_CAP, 3, dtype=torch.float64).data_ptr(), self.ALOG_CAP
        # the controller leaves the host's copy of the control block in pinned memory itself (see _ctl_posted)
        # ... unless host_copy == "side": a kernel that writes host memory holds the stream until the write has crossed PCIe
        # (~5 us before the next launch may start); a copy on the side stream does not
        c.ctl_host = None if (torch.cuda.is_current_stream_capturing() or HOST_COPY == "side") else self._ctl_io(P)[1].data_ptr()
        # Where the norm + controller run.  Fused into the RK launch's epilogue (last workgroup of a problem) for the two
        # one-stage launches of the initial-step selection: same GPU time as a launch of their own (26.7 us against
        # 18 + 9), one launch less each.  NOT for an attempted step: the epilogue's device-scope atomics queue behind the
        # six stages' stores (+18 us against a 9 us launch, MI355X) — it keeps the separate, slot-aware launch.  Data
        # parallel: always separate (the sums are all-reduced between the norm and the controller).
        if (self.comm is not None and self.comm.world > 1) or norm_mode not in self.FUSED_NORM_MODES:
            c.norm_mode = -1
        else:
            c.norm_mode = norm_mode
            c.partials = self._buf("cpart", P, nblk, 2).data_ptr()
            c.tickets = self._buf("ctickets", P, dtype=torch.int32).data_ptr()
        return c

    def _chain_control(self, ws0, pool, chain, y0, u, mode, P, rpp):
        """the scaled norm + step controller as launches of their own (slot-aware), where the RK launch did not run them
        in its epilogue (see ``_chain``)"""
        if chain.norm_mode >= 0 and not (mode == 2 and chain.norm_defer):
            return
        if chain.norm_mode >= 0:
            # an attempt whose RK launch left its tiles' partial sums (norm_defer): one small workgroup per problem
            if chain.ctl_host:
                chain.ctl_seq = self._seq_next()
            _lib.call("nlbac_dopri_control_tiles", C.byref(chain), self.n_s, self.n_u, rpp, P, stream_ptr())
            return
        ctl = self._ctl(P)
        if mode == 0:
            self._norm_control(ws0.K[0], None, y0, None, u, 0, P, rpp)
        elif mode == 1:
            self._norm_control(ws0.K[1], ws0.K[0], y0, None, None, 1, P, rpp)
        else:
            if chain.ctl_host and not (self.comm is not None and self.comm.world > 1):
                chain.ctl_seq = self._seq_next()
            self._norm_control(ws0.err, None, y0, ws0.Y[6], None, 2, P, rpp, slot_ctl=ctl.data_ptr(),
                               slot_floats=pool.slot_floats, chain=chain)

    def _dopri_begin_chain(self, y0, u, P, rpp, min_slots=1):
        n, S = P * rpp, 7
        ctx = self.ctx
        ctx["ctl_seq"] = None         # (stamps of this solve's controller launches start here: _seq_next)
        pool = self._pool(n, S, min_slots)
        ws0 = pool.ws(n, 0)
        ctl = self._ctl(P)
        cp = ctl.data_ptr()
        ch = [self._chain(ws0, pool, P, rpp, m, read_ctl=(m == 2)) for m in (0, 1, 2)]
        ip, om = self._interp_fold(), self._out_map_fwd(n)
        if ip and om is not None and isinstance(self, ConcatNodeSolver):
            ip = False                # (the single-net kernels evaluate no out-map)
        if ip:
            ch[2].interp_out = self._out_buf(n).data_ptr()
            if om is not None:
                ch[2].interp_kind, ch[2].interp_l, ch[2].interp_p = om.kind, om.l, om.p
        ctx["chain"] = dict(pool=pool, ws0=ws0, ch=ch[2], attempts=0, y0=y0, ip=ip, ip_om=om is not None)
        k = max(1, int(self.__dict__.get("_chain_len", 1)))
        if self._begin_persistent(ws0, ch, y0, u, P, rpp):
            # f0, the probe and the first attempted step were ONE launch (nlbac_node_rk_fwd_begin); the attempt's norm +
            # controller and any further attempts follow as usual
            self._chain_attempts(k, first_rk_done=True)
            return
        if self._norm_defer_ok(ch):
            # The norms of f0 and of the probe without their elections (nlbac_rk_chain::norm_defer / norm_pre): each
            # launch leaves its tiles' partial sums, the NEXT launch's workgroups sum them and run the controller
            # themselves under their prologue's loads.
            nblk = (rpp + _lib.MLP_TILE - 1) // _lib.MLP_TILE
            part0, part1 = self._buf("cpart", P, nblk, 2).data_ptr(), self._buf("cpart1", P, nblk, 2).data_ptr()
            ch[0].norm_defer, ch[0].partials = 1, part0
            ch[1].norm_pre, ch[1].partials_pre, ch[1].norm_defer, ch[1].partials = 1, part0, 1, part1
            if self.__dict__.get("norm_defer_attempt", NORM_DEFER_ATTEMPT):
                # ... and the attempts': the RK launch leaves the error norm's tile partials, the controller launch is one
                # 64-thread workgroup per problem (nlbac_dopri_control_tiles) instead of a pass over the error rows
                ch[2].norm_mode, ch[2].norm_defer = 2, 1
                ch[2].partials = self._buf("cpart2", P, nblk, 2).data_ptr()
            first = _lib.RkChain.from_buffer_copy(ch[2])
            first.norm_pre, first.partials_pre = 2, part1
            ctx["chain"]["ch_first"] = first        # (the first attempted step only: later attempts get their step size from the controller launch)
        # f0 + Hairer's first guess, the probe f(y0 + h0 f0) + the initial step: one launch each (norms fused)
        self._rk_fused(ws0, y0, u, P, rpp, "dopri5", 0, 1, h_dev=cp, chain=ch[0])
        self._chain_control(ws0, pool, ch[0], y0, u, 0, P, rpp)
        self._rk_fused(ws0, y0, u, P, rpp, "probe", 1, 2, h_dev=cp + 8 * 6, save_acts=False, chain=ch[1])
        self._chain_control(ws0, pool, ch[1], y0, u, 1, P, rpp)
        self._chain_attempts(k)

    def _norm_defer_ok(self, ch):
        """The election-free form of the two fused norms that open a dopri5 solve: where those norms are fused at all (one
        GPU) and the register-resident kernels serve the nets.  ``norm_defer = False`` (NLBAC_NORM_DEFER=0): the fused
        norms with their elections — the cross-check."""
        on = self.__dict__.get("norm_defer")
        if on is None:
            on = self.norm_defer = os.environ.get("NLBAC_NORM_DEFER", "1") != "0"
        if not on or ch[0].norm_mode != 0 or ch[1].norm_mode != 1 or self._interp_nets()[1] is None:
            return False
        ok = self.__dict__.get("_interp_ok")
        if ok is None:
            ok = self._interp_ok = bool(_lib.load().nlbac_rk_interp_ok(*self._interp_nets()))
        return bool(ok and self.fused)

    def _begin_persistent(self, ws0, ch, y0, u, P, rpp):
        """f0 + first guess, probe + initial step and the first attempted step as ONE persistent launch
        (nlbac_node_rk_fwd_begin) where the kernels and the sizes allow it: single GPU (the first two norms are fused:
        no all-reduce in between), mask words (rollouts), every workgroup resident at once (<= 8192 rows).
        OFF unless ``persistent = True`` / NLBAC_NODE_PERSIST=1: measured on MI355X it does not win — 108 us against
        23.6 + 20.7 + 60.9 = 105 us for the three launches at 8192 rows (profiles/r04: what the one-stage launches cost
        beyond their stage is not dispatch but the norm's election — partial sums, ticket, the controller, the release —
        ~10 us of serial round trips each, and the persistent launch has the same two)."""
        if not (self.fused and (self.comm is None or self.comm.world == 1) and ws0.bits and not self.adjoint):
            return False
        on = self.__dict__.get("persistent")
        if on is None:
            on = os.environ.get("NLBAC_NODE_PERSIST", "0") == "1"
        if not on:
            return False
        key = (P, rpp)
        ok = self.__dict__.setdefault("_pers_ok", {}).get(key)
        if ok is None:
            ok = self._pers_ok[key] = bool(_lib.load().nlbac_node_rk_fwd_begin_ok(C.byref(self.f.desc), C.byref(self.g.desc), P, rpp))
        if not ok or ch[0].norm_mode != 0 or ch[1].norm_mode != 1:       # (the first two norms must be the fused ones)
            return False
        n = P * rpp
        im = None
        if self.ctx.pop("in_map_pending", None):
            m, im = self._in_map, _lib.InMap()
            im.kind, im.obs, im.obs_ld, im.l = m["kind"], m["obs"].data_ptr(), m["obs_ld"], m["l"]
            im.ps = m["ps"].data_ptr() if m["ps"] is not None else None
        c = _lib.RkChain.from_buffer_copy(ch[2])
        c.partials, c.tickets = ch[0].partials, ch[0].tickets       # (the fused norms of the first two phases)
        gen = self._buf("pers_gen", 8, dtype=torch.int32)
        self._pers_id = self.__dict__.get("_pers_id", 0) + 1
        beta, S = self._beta("dopri5")
        cerr = self._coef("err")
        _lib.call("nlbac_node_rk_fwd_begin", C.byref(self.f.desc), C.byref(self.g.desc), y0.data_ptr(), u.data_ptr(), P, rpp,
                  beta, cerr, len(cerr), ws0.K.data_ptr(), ws0.Y.data_ptr(), ws0.gout.data_ptr(),
                  ws0.acts_f.data_ptr(), ws0.S * n * ws0.wf, ws0.acts_g.data_ptr(), ws0.S * n * ws0.wg, ws0.err.data_ptr(),
                  C.byref(c), C.byref(im) if im is not None else None, gen.data_ptr(), 2 * self._pers_id, stream_ptr())
        self.nfe += 8
        return True

    def _chain_attempts(self, k, first_rk_done=False):
        ctx = self.ctx
        st = ctx["chain"]
        P, rpp, u = ctx["P"], ctx["rpp"], ctx["u"]
        ws0, pool, ch = st["ws0"], st["pool"], st["ch"]
        cp = self._ctl(P).data_ptr()
        for i in range(k):
            if not (first_rk_done and i == 0):
                self._rk_fused(ws0, st["y0"], u, P, rpp, "dopri5", 1, 7, h_dev=cp, c_err=self._coef("err"), err=ws0.err,
                               chain=st.pop("ch_first", None) or ch)
            self._chain_control(ws0, pool, ch, st["y0"], u, 2, P, rpp)
        st["attempts"] += k
        if (self.comm is not None and self.comm.world > 1) or HOST_COPY == "side" or (ch.norm_mode == 2 and not ch.norm_defer):
            self._ctl_post(P)        # (the all-reduced controller is nlbac_dopri_control: it leaves no host copy; nor does
                                     #  the RK launch's own epilogue, FUSED_NORM_MODES with 2)
        else:
            self._ctl_posted(P)

    def _dopri_finish_chain(self, assume_done=False):
        ctx = self.ctx
        st = ctx["chain"]
        P, rpp, n, ns = ctx["P"], ctx["rpp"], ctx["n"], self.n_s
        pool, ws0 = st["pool"], st["ws0"]
        ctl = self._ctl(P)
        c = None
        while not assume_done:
            c = ctx.pop("ctl_host", None)
            if c is None:
                c = self._ctl_read(P)               # the one host wait per CHAIN of attempts
            c = c.tolist()                          # (plain floats: see first_step_done)
            if any(c[p][13] > 0 for p in range(P)):
                # out of step slots: the solve was stopped; start it again in a pool with room for twice as many steps
                self._dopri_begin_chain(ctx["y0"], ctx["u"], P, rpp, min_slots=2 * pool.n_slots)
                st = ctx["chain"]
                pool, ws0 = st["pool"], st["ws0"]
                continue
            if all(c[p][4] > 0 for p in range(P)):
                break
            if st["attempts"] >= 1000:
                raise _lib.NlbacError("dopri5: max_num_steps exceeded")
            self._chain_attempts(2)
        out = self._out_buf(n)
        if st.get("ip"):
            # the attempt that finished each problem has written its rows of `out` (and of the owner's map) itself
            ctx["out_mapped"] = st["ip_om"]
        else:
            # (the owner's map of the output — the Unicycle tasks' look-ahead point — is evaluated by this launch)
            om = self._out_map_fwd(n)
            _lib.call("nlbac_dopri_interp_fwd", ctx["y0"].data_ptr(), ws0.Y[6].data_ptr(), ws0.K.data_ptr(), None, None,
                      ctl.data_ptr(), P, rpp, ns, out.data_ptr(), pool.slot_floats, C.byref(om) if om is not None else None,
                      stream_ptr())
            ctx["out_mapped"] = om is not None
        if c is not None:
            nst = [int(c[p][10]) for p in range(P)]
            nacc = [int(c[p][12]) for p in range(P)]
            self._chain_len = max(1, max(nst))
            key = "single_step" if max(nst) == 1 else "multi_attempt"
            self.stats[key] += 1
            alog = self._buf("alog", P, self.ALOG_CAP, 3, dtype=torch.float64).cpu() if max(nst) > 1 else None
            info = []
            for k in range(min(max(nst), self.ALOG_CAP)):
                row = []
                for p in range(P):
                    if alog is None:
                        row.append((c[p][11], c[p][2], True))
                    elif k < nst[p]:
                        row.append((float(alog[p, k, 0]), float(alog[p, k, 1]), bool(alog[p, k, 2] > 0)))
                    else:
                        row.append(None)
                info.append(row)
            ctx.update(nacc=nacc, info=info)
            ctx["steps"] = [dict(ws=pool.ws(n, i), first=(i == 0)) for i in range(max(nacc) + 1)]
        else:
            ctx.update(nacc=None, steps=[dict(ws=ws0, first=True)])
        ctx["out"] = out
        return out

    def _backward_chain(self, dout, need_du, need_params, need_dy0):
        ctx = self.ctx
        st = ctx["chain"]
        P, rpp, n, u = ctx["P"], ctx["rpp"], ctx["n"], ctx["u"]
        ns, nu, s = self.n_s, self.n_u, stream_ptr()
        pool, ws0, ch = st["pool"], st["ws0"], st["ch"]
        ctl = self._ctl(P)
        # launches: one per accepted step of the slowest problem (unknown on the host inside a graph capture: then one
        # per attempt that was enqueued — launches beyond a problem's first step return at once)
        nb = (max(ctx["nacc"]) + 1) if ctx.get("nacc") is not None else st["attempts"]
        for i in range(nb):
            pool.ws(n, i).bwd(self)
        du = self._buf("du", n, nu) if need_du else None
        om = self._out_map_bwd(n) if dout is None else None
        assert dout is not None or om is not None, "backward(None) needs an output map (set_out_map) with its gradients"
        bch = _lib.RkChain()
        bch.ctl, bch.slot_floats, bch.n_slots, bch.hslots, bch.norm_mode = ch.ctl_w, pool.slot_floats, pool.n_slots, ch.hslots, -1
        if st.get("ip") and not (om is not None and isinstance(self, ConcatNodeSolver)):
            # the backward of the interpolant is the prologue of each problem's last-step launch (back_idx 0)
            bch.interp_bwd = 1
            if om is not None:
                bch.interp_kind, bch.interp_l, bch.interp_dp, bch.interp_dp2, bch.interp_x = om.kind, om.l, om.dp, om.dp2, om.x
            else:
                assert dout.is_contiguous() and dout.shape == (n, ns)
                bch.interp_dout = dout.data_ptr()
        else:
            _lib.call("nlbac_dopri_interp_bwd", dout.data_ptr() if dout is not None else None, None, None, ctl.data_ptr(), P,
                      rpp, ns, ws0.dy0.data_ptr(), ws0.dy1.data_ptr(), ws0.dK.data_ptr(), pool.slot_floats,
                      C.byref(om) if om is not None else None, s)
        for b in range(nb):
            self._rk_fused_bwd(ws0, u, P, rpp, "dopri5", True, need_dy0, need_params, None, None, 0, ws0.dy1, du,
                               b == 0, chain=bch, back_idx=b)
        return du, (ws0.dy0 if need_dy0 else None)

    def _dopri_accept_first(self, c):
        """First step accepted and past dt: interpolate.  Step size and abscissa are read from the device
        control block by the kernels (identical arithmetic with or without a host copy of them)."""
        ctx = self.ctx
        self.stats["single_step"] += 1
        P, rpp, n, ns = ctx["P"], ctx["rpp"], ctx["n"], self.n_s
        ws = self._step_ws(n, 7, 0)
        ctl = self._ctl(P)
        out = self._out_buf(n)       # (y1 is the input of stage 6: read in place, no copy)
        _lib.call("nlbac_dopri_interp_fwd", ctx["y0"].data_ptr(), ws.Y[6].data_ptr(), ws.K.data_ptr(), None, None,
                  ctl.data_ptr(), P, rpp, ns, out.data_ptr(), 0, None, stream_ptr())
        step = dict(ws=ws, first=True, dev=True)
        if c is not None:
            step["h"] = [float(c[p, 11]) for p in range(P)]
            step["x"] = [float(c[p, 5]) for p in range(P)]
            ctx["info"] = [[(float(c[p, 11]), float(c[p, 2]), True) for p in range(P)]]
        ctx.update(steps=[step], out=out)
        return out

    def _dopri_continue(self, resume=None):
        """The attempt loop after the first attempted step has been queued.  ``resume``: state taken over from a joint
        solve (accepted steps so far, index of the step being attempted, attempt number) — see ``_adopt``."""
        ctx = self.ctx
        P, rpp, n, u, y0 = ctx["P"], ctx["rpp"], ctx["n"], ctx["u"], ctx["y0"]
        ns, S = self.n_s, 7
        s = stream_ptr()
        ctl = self._ctl(P)
        steps, info = [], []
        cur_y0, idx, start = y0, 0, 0
        if resume is not None:
            steps, info, cur_y0, idx, start = (resume["steps"], resume["info"], resume["cur_y0"], resume["idx"],
                                               resume["attempt"])
        ws = self._step_ws(n, S, idx)
        for attempt in range(start, 1000):
            c = ctx.pop("ctl_host", None)
            if c is None:
                c = self._ctl_read(P)             # the one host wait per attempted step
            acc = [bool(c[p, 3] > 0) for p in range(P)]
            done = [bool(c[p, 4] > 0) for p in range(P)]
            if any(a != acc[0] for a in acc) or any(d != done[0] for d in done):
                return self._solve_split(c, dict(steps=steps, info=info, idx=idx, attempt=attempt))
            if attempt == 0 and acc[0] and done[0]:
                return self._dopri_accept_first(c)
            info.append([(float(c[p, 11]), float(c[p, 2]), acc[p]) for p in range(P)])
            if attempt == 0:
                self.stats["multi_attempt"] += 1
            if acc[0]:
                steps.append(dict(ws=ws, h=[float(c[p, 11]) for p in range(P)], first=(idx == 0)))
                if done[0]:
                    x = [float(c[p, 5]) for p in range(P)]
                    steps[-1]["x"] = x
                    out = self._out_buf(n)
                    _lib.call("nlbac_dopri_interp_fwd", cur_y0.data_ptr(), ws.Y[6].data_ptr(), ws.K.data_ptr(),
                              fptr(*steps[-1]["h"]), fptr(*x), None, P, rpp, ns, out.data_ptr(), 0, None, s)
                    ctx.update(steps=steps, out=out, info=info)
                    return out
                cur_y0 = ws.Y[6]                  # y1 of an accepted step = its stage-6 input (ws is not reused)
                idx += 1
                prev = ws
                ws = self._step_ws(n, S, idx)
                ws.Y[0].copy_(prev.Y[6])
                ws.K[0].copy_(prev.K[6])          # FSAL
            self._dopri_attempt(ws, cur_y0, u, P, rpp)
        raise _lib.NlbacError("dopri5: max_num_steps exceeded")

    def _adopt(self, k, p, c, st):
        """Hand per-problem solver ``k`` everything the joint solve has done for problem ``p``: its rows of every step
        workspace so far (stage derivatives, stage inputs, g(x), error estimate, activations / ReLU masks — one
        strided-copy launch per buffer) and its control block, so that it continues from the accept decision ``c``
        instead of starting the solve again.  Returns the state ``k._dopri_continue`` resumes from."""
        ctx = self.ctx
        P, rpp, n, S = ctx["P"], ctx["rpp"], ctx["n"], 7
        rows = slice(p * rpp, (p + 1) * rpp)
        k._touch(rpp)
        k.stats["solves"] += 1
        k.ctx = dict(method="dopri5", P=1, rpp=rpp, n=rpp, u=ctx["u"][rows], y0=ctx["y0"][rows], steps=[],
                     t_end=ctx["t_end"], atol=ctx["atol"], rtol=ctx["rtol"])
        s = stream_ptr()
        kws = []
        for j in range(st["idx"] + 1):                    # accepted steps 0 .. idx-1 and the step being attempted
            src, dst = self._step_ws(n, S, j), k._step_ws(rpp, S, j)
            for name in src.ADOPT:
                a, b = getattr(src, name), getattr(dst, name)
                w = a.shape[-1]                           # [.., rows, w] with rows = n or S*n (stage-major)
                _lib.call("nlbac_copy_blocks", a.data_ptr() + 4 * p * rpp * w, n * w, b.data_ptr(), rpp * w, rpp * w,
                          a.numel() // (n * w), s)
            kws.append(dst)
        _lib.call("nlbac_copy_blocks", self._ctl(P).data_ptr() + 8 * _lib.DOPRI_CTL * p, 2 * _lib.DOPRI_CTL,
                  k._ctl(1).data_ptr(), 2 * _lib.DOPRI_CTL, 2 * _lib.DOPRI_CTL, 1, s)
        k.ctx["ctl_host"] = c[p:p + 1].clone()
        steps = [dict(ws=kws[j], h=[step["h"][p]], first=step["first"]) for j, step in enumerate(st["steps"])]
        return dict(steps=steps, info=[[e[p]] for e in st["info"]], idx=st["idx"], attempt=st["attempt"],
                    cur_y0=kws[st["idx"] - 1].Y[6] if st["idx"] else k.ctx["y0"])

    def _solve_split(self, c, st):
        """The problems of one batch want different step sequences (one accepted / finished, another not): each has its
        own adaptive step size in the reference too (separate odeint calls), so the solve is finished problem by
        problem by child solvers on the row ranges, which take over what has been done jointly (``_adopt``; ``c``: host
        copy of the control block, ``st``: accepted steps / attempt number at the point of disagreement)."""
        ctx = self.ctx
        self.stats["split"] += 1
        P, rpp, n = ctx["P"], ctx["rpp"], ctx["n"]
        out = self._out_buf(n)
        kids, info = [], []
        for p in range(P):
            if p not in self._children:
                self._children[p] = type(self)(self.node, self.device)
            k = self._children[p]
            k.comm, k.fused, k.keep_acts = self.comm, self.fused, self.keep_acts
            self._ctl_io(1)
            k._side, k._ev_ctl, k._ctl_pin = self._side, self._ev_ctl, self._ctl_pin
            k.before_wait = self.__dict__.get("before_wait")
            rows = slice(p * rpp, (p + 1) * rpp)
            if p == 0:
                key = "adopted" if st["attempt"] == 0 else "adopted_late"
                self.stats[key] = self.stats.get(key, 0) + 1
            o = k._dopri_continue(self._adopt(k, p, c, st))
            _lib.call("nlbac_copy_blocks", o.data_ptr(), o.numel(), out.data_ptr() + 4 * p * rpp * self.n_s, o.numel(),
                      o.numel(), 1, stream_ptr())
            kids.append(k)
            info.append(k.ctx.get("info"))
        ctx.update(split=kids, out=out, steps=[], info_split=info)
        ctx.pop("info", None)
        return out

    # -- backward --------------------------------------------------------------
    def backward(self, dout, need_du=True, need_params=False, need_dy0=False):
        """dout: (n, n_s).  Returns (du or None, dy0 or None).  With
        ``need_params`` the pre-activation grads of every stage are kept for
        ``accumulate_param_grads``."""
        ctx = self.ctx
        if self.adjoint:
            return self.backward_adjoint(dout, need_du, need_params, need_dy0)
        if ctx.get("chain"):
            return self._backward_chain(dout, need_du, need_params, need_dy0)
        if ctx.get("split"):
            assert not need_params, "parameter gradients are only taken on single-problem solves"
            rpp = ctx["rpp"]
            du = self._buf("du", ctx["n"], self.n_u) if need_du else None
            dy0 = self._buf("dy0_split", ctx["n"], self.n_s) if need_dy0 else None
            for p, k in enumerate(ctx["split"]):
                rows = slice(p * rpp, (p + 1) * rpp)
                du_p, dy0_p = k.backward(dout[rows], need_du=need_du, need_dy0=need_dy0)
                if du is not None:
                    du[rows].copy_(du_p)
                if dy0 is not None:
                    dy0[rows].copy_(dy0_p)
            return du, dy0
        P, rpp, n, u, method = ctx["P"], ctx["rpp"], ctx["n"], ctx["u"], ctx["method"]
        ns, nu = self.n_s, self.n_u
        s = stream_ptr()
        assert not (need_params and not self.keep_acts), "this solver keeps ReLU masks only (keep_acts=False)"
        du = self._buf("du", n, nu) if need_du else None
        if du is not None and not self.fused:
            du.zero_()           # (the fused kernel overwrites du on the first step it processes)
        steps = ctx["steps"]
        dy_carry = None          # grad wrt the y1 of the step being processed
        dk_carry = None          # grad wrt f1 (=K[6]) of that step, from the next step's FSAL stage 0
        for si in range(len(steps) - 1, -1, -1):
            step = steps[si]
            ws = step["ws"].bwd(self)
            S = ws.S
            dev = step.get("dev", False)       # step size lives in the device control block
            h_host = None if dev else fptr(*step["h"])
            h_dev = self._ctl(P).data_ptr() + 8 * 11 if dev else None      # C_HUSED
            h_stride = _lib.DOPRI_CTL if dev else 0
            last = si == len(steps) - 1
            if not (method == "dopri5" and last):
                ws.dK.zero_()              # (the interpolant's backward assigns every stage of dK itself)
            if method == "dopri5":
                beta, first_eval = DP_BETA, step["first"]
                if last:
                    _lib.call("nlbac_dopri_interp_bwd", dout.data_ptr(), h_host, None if dev else fptr(*step["x"]),
                              self._ctl(P).data_ptr() if dev else None, P, rpp, ns,
                              ws.dy0.data_ptr(), ws.dy1.data_ptr(), ws.dK.data_ptr(), 0, None, s)
                else:
                    ws.dy0.zero_()
                    ws.dy1.copy_(dy_carry)
                    # (own kernel: the first use of an ATen op lazily loads its code object - ~60 ms in the middle of training)
                    _lib.call("nlbac_axpby", 1.0, ws.dK[6].data_ptr(), 1.0, dk_carry.data_ptr(), ws.dK[6].numel(),
                              ws.dK[6].data_ptr(), s)
                top_up = ws.dy1           # y1 == stage-6 input
            else:
                tab = TABLEAU[method]
                beta, first_eval = tab["beta"], True
                # out = y0 + h sum c_j K_j
                _lib.call("nlbac_rk_stage_bwd", dout.data_ptr(), None, None, 0, S, fptr(*tab["c_sol"]), h_host,
                          None, 0, P, rpp, ns, ws.dK.data_ptr(), ws.dy0.data_ptr(), 0, None, 0, s)
                top_up = None
            if self.fused:
                self._rk_fused_bwd(ws, u, P, rpp, method, first_eval, need_dy0, need_params, h_host, h_dev, h_stride,
                                   top_up, du, last)
                dy_carry = ws.dy0
                dk_carry = ws.dK[0]
                continue
            for st in range(S - 1, -1, -1):
                if st == 0 and not first_eval:
                    break                  # FSAL alias of the previous step's last stage
                need_dx = (st > 0) or need_dy0
                up = top_up if (st == S - 1 and top_up is not None) else None
                coef = beta[st - 1] if st > 0 else []
                self._stage_backward(ws, st, need_dx, need_params, du, up, coef, h_host, h_dev, h_stride)
            dy_carry = ws.dy0
            dk_carry = ws.dK[0]
        dy0 = steps[0]["ws"].dy0 if need_dy0 else None
        return du, dy0

    # -- continuous adjoint (odeint_adjoint) ---------------------------------------------------------
    # torchdiffeq 0.2.3 OdeintAdjointMethod.backward restated on the device: the augmented state z = [y | a_x | a_u]
    # (+ the parameter adjoint, a quadrature) is integrated from t1 back to t0 with the forward's method and
    # tolerances and the mixed default adjoint norm; one nlbac_node_adj_step launch per RK step re-computes the nets on
    # every stage input and back-propagates a_x through them, so nothing of the forward solve is kept.  The dopri5
    # attempts are a device-driven chain (kernels skip problems whose solve is done, an accepted step is handed over
    # by nlbac_adj_commit); the host looks at the control block once per chain, not once per attempt.
    ADJ_MAX_ATTEMPTS = 1000

    def _adj_ws(self, n, S):
        key = ("adj", n, S)
        pool = self._scratch.setdefault(n, {})
        w = pool.get(key)
        if w is None:
            W = 2 * self.n_s + self.n_u
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
            w = pool[key] = dict(Z0=z(n, W), Z1=z(n, W), KZ=z(S, n, W), ERR=z(n, W), OUT=z(n, W), W=W)
        return w

    def _adj_interp_fold(self):
        """The attempt launches of the adjoint solve write the interpolant of z at t_end themselves (no
        nlbac_dopri_interp_fwd launch behind the solve) where the kernel that serves the nets does so."""
        ok = self.__dict__.get("_adj_ip_ok")
        if ok is None:
            ok = self._adj_ip_ok = bool(_lib.load().nlbac_node_adj_interp_ok(C.byref(self.f.desc), C.byref(self.g.desc)))
        return ok and self._interp_fold_on()

    def _interp_fold_on(self):
        on = self.__dict__.get("interp_fold")
        if on is None:
            on = self.interp_fold = os.environ.get("NLBAC_INTERP_FOLD", "1") != "0"
        return on

    def _adj_step(self, w, u, P, rpp, method, st0, st1, h_host=None, h_dev=None, ctl=None, c_out=None, c_err=None,
                  keep=None, interp=False):
        beta, S = self._beta(method)
        f, g = self.f, self.g
        k = keep or {}
        dp = lambda t: t.data_ptr() if t is not None else None
        _lib.call("nlbac_node_adj_step", C.byref(f.desc), C.byref(g.desc), u.data_ptr(), P, rpp, st0, st1, S, beta,
                  c_out, len(c_out) if c_out is not None else 0, c_err, len(c_err) if c_err is not None else 0,
                  fptr(*h_host) if h_host is not None else None, h_dev, _lib.DOPRI_CTL if h_dev else 0, ctl,
                  w["Z0"].data_ptr(), w["KZ"].data_ptr(), w["Z1"].data_ptr() if c_out is not None else None,
                  w["ERR"].data_ptr() if c_err is not None else None, dp(k.get("ZS")), dp(k.get("dG")),
                  dp(k.get("acts_f")), k.get("ls_f", 0), dp(k.get("acts_g")), k.get("ls_g", 0), dp(k.get("dz_f")),
                  dp(k.get("dz_g")), w["OUT"].data_ptr() if interp else None, self.ctx["t_end"], stream_ptr())
        self.nfe += st1 - st0
        if keep:
            for st in range(st0, st1):
                self._adj_stage_dw(self._adj_par_cur, st)

    def _adj_norm_control(self, a, b, w, u, mode, P, rpp, ctl, pnorm=None):
        ctx = self.ctx
        ns, nu, s = self.n_s, self.n_u, stream_ptr()
        nblk = (rpp + 255) // 256
        part = self._buf("adj_part", P, nblk, 4)
        dp = lambda t: t.data_ptr() if t is not None else None
        if self.comm is not None and self.comm.world > 1:
            _lib.call("nlbac_adj_norm_control", dp(a), dp(b), w["Z0"].data_ptr(), w["Z1"].data_ptr(), dp(u), mode,
                      ctx["rtol"], ctx["atol"], ns, nu, rpp, P, ctx["t_end"], None, part.data_ptr(), None,
                      ctl.data_ptr(), None, 0.0, s)
            sums = self._buf("adj_psum", P, 1, 4)
            for p in range(P):
                _lib.call("nlbac_sum_partials", part[p].data_ptr(), nblk, 4, 1.0, sums[p].data_ptr(), s)
            self.comm.all_reduce_(sums)
            _lib.call("nlbac_adj_control", sums.data_ptr(), 1, mode, ns, nu, rpp * self.comm.world, P, ctx["t_end"],
                      dp(pnorm), ctl.data_ptr(), s)
            return
        tickets = self._buf("adj_tickets", P, dtype=torch.int32)
        # (an attempt's controller leaves the host's copy of the control block in pinned memory itself: no copy launch
        #  between the decision and the host; see _ctl_posted)
        host, seq = None, 0.0
        if mode == 2 and not torch.cuda.is_current_stream_capturing() and HOST_COPY != "side":
            host = self._ctl_io(P)[1].data_ptr()
            seq = self._seq_next()
        ctx["adj_ctl_host"] = host is not None
        _lib.call("nlbac_adj_norm_control", dp(a), dp(b), w["Z0"].data_ptr(), w["Z1"].data_ptr(), dp(u), mode,
                  ctx["rtol"], ctx["atol"], ns, nu, rpp, P, ctx["t_end"], dp(pnorm), part.data_ptr(),
                  tickets.data_ptr(), ctl.data_ptr(), host, seq, s)

    # -- parameter adjoint (a quadrature beside the per-row state; single-problem solves) -----------------
    ADJ_SUB_SLABS = 40       # row slabs of one stage's weight-gradient GEMM (workgroups: layers x slabs x nets)

    def _adj_params_begin(self, w, n, S):
        ctx = self.ctx
        assert ctx["P"] == 1, "parameter gradients are only taken on single-problem solves"
        key = ("adj_par", n, S)
        pool = self._scratch.setdefault(n, {})
        par = pool.get(key)
        if par is None:
            f, g, ns, nu, dev = self.f, self.g, self.n_s, self.n_u, self.device
            z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
            arena = f.arena
            NP = arena.n
            keep = dict(ZS=z(S, n, w["W"]), dG=z(S, n, ns * nu), acts_f=z(f.n_layers - 1, S * n, f.hid),
                        acts_g=z(g.n_layers - 1, S * n, g.hid), dz_f=z(f.n_layers - 1, S * n, f.hid),
                        dz_g=z(g.n_layers - 1, S * n, g.hid), ls_f=S * n * f.hid, ls_g=S * n * g.hid)
            segs = [(arena.offset_of[id(p)], p.numel()) for p in self.node.parameters()]
            par = pool[key] = dict(
                keep=keep, NP=NP, K=z(S, NP), th0=z(NP), th1=z(NP), out=z(NP), slabs=z(self.ADJ_SUB_SLABS, NP),
                seg_off=torch.tensor([o for o, _ in segs], dtype=torch.int32, device=dev),
                seg_len=torch.tensor([l for _, l in segs], dtype=torch.int32, device=dev), n_seg=len(segs),
                pseg=z(2 * len(segs)), ticket=z(1, dtype=torch.int32), pnorm=z(2), io={}, n=n, S=S, w=w)
        _lib.call("nlbac_fill", par["th0"].data_ptr(), 0.0, par["NP"], stream_ptr())
        par["grad"] = None
        return par

    def _adj_stage_dw(self, par, st):
        """K_theta[st] = sum over the rows of stage ``st`` of (dF/dtheta)^T a_x: nlbac_mlp_bwd_weights on what the
        step kernel kept of that stage (row slabs), then the slab sum."""
        k, n, S, w = par["keep"], par["n"], par["S"], par["w"]
        W, ns, nu = w["W"], self.n_s, self.n_u
        io = par["io"].get(st)
        if io is None:
            io = par["io"][st] = io_array(2)
            ZS = k["ZS"][st]
            for i, (net, acts, dz) in enumerate(((self.f, k["acts_f"], k["dz_f"]), (self.g, k["acts_g"], k["dz_g"]))):
                io[i].x0, io[i].x0_dim, io[i].x0_ld = ZS.data_ptr(), ns, W
                io[i].acts, io[i].dz = acts[:, st * n:].data_ptr(), dz[:, st * n:].data_ptr()
                io[i].acts_ls = S * n * net.hid
                io[i].grad = par["slabs"].data_ptr()
            io[0].dy, io[0].dy_ld = ZS.data_ptr() + 4 * ns, W              # cotangent of f_net's output: a_x
            io[1].dy, io[1].dy_ld = k["dG"][st].data_ptr(), ns * nu        # of g_net's: a_x u^T
        bwd_weights(self._nets(), io, 2, n, self.ADJ_SUB_SLABS, par["NP"], self.device)
        _lib.call("nlbac_reduce_slabs", par["K"][st].data_ptr(), par["slabs"].data_ptr(), self.ADJ_SUB_SLABS,
                  par["NP"], par["NP"], stream_ptr())
        if self.comm is not None and self.comm.world > 1:
            # sample-sharded solve: the parameter adjoint is a sum over ALL rows, and its norm takes part in the step
            # control — every rank must form it from the same (global) stage derivative, or the ranks' accept / done
            # decisions part ways and their collectives no longer pair up
            self.comm.all_reduce_(par["K"][st])

    def _adj_params_norm(self, par, mode, cp, h_host=None, c_sol=None):
        ctx = self.ctx
        _lib.call("nlbac_adj_param_norm", mode, par["th0"].data_ptr(), par["K"].data_ptr(), par["NP"], par["S"],
                  c_sol if c_sol is not None else self._coef("sol"), self._coef("err") if c_sol is None else fptr(*([0.0] * par["S"])),
                  fptr(h_host) if h_host is not None else None, cp if h_host is None else None,
                  par["seg_off"].data_ptr(), par["seg_len"].data_ptr(), par["n_seg"], ctx["rtol"], ctx["atol"],
                  cp if (mode == 2 and h_host is None) else None, par["th1"].data_ptr(), par["pseg"].data_ptr(),
                  par["ticket"].data_ptr(), par["pnorm"].data_ptr(), stream_ptr())
        return par["pnorm"]

    def _adj_params_commit(self, par, cp):
        NP = par["NP"]
        _lib.call("nlbac_adj_commit", cp, NP // 4, NP // 4, 4, par["th0"].data_ptr(), par["th1"].data_ptr(),
                  par["K"][0].data_ptr(), par["K"][6].data_ptr(), stream_ptr())

    def _adj_params_finish(self, par, cp):
        NP = par["NP"]
        _lib.call("nlbac_dopri_interp_fwd", par["th0"].data_ptr(), par["th1"].data_ptr(), par["K"].data_ptr(), None,
                  None, cp, 1, NP // 4, 4, par["out"].data_ptr(), 0, None, stream_ptr())
        par["grad"] = par["out"]
        self.ctx["adj_par"] = par

    def _adj_params_fixed(self, par, w, c_sol, h):
        """fixed grid: theta_bar(t0) = h sum_j c_sol[j] K_theta[j]"""
        self._adj_params_norm(par, 2, None, h_host=h, c_sol=fptr(*c_sol))
        par["grad"] = par["th1"]
        self.ctx["adj_par"] = par

    def backward_adjoint(self, dout, need_du=True, need_params=False, need_dy0=False):
        """dL/du, dL/dy0 (and, with ``need_params``, the parameter adjoint for ``accumulate_param_grads``) from
        dL/dy(t1) = ``dout`` by solving the adjoint system backwards from the forward's y(t1)."""
        ctx = self.ctx
        P, rpp, n, u, method = ctx["P"], ctx["rpp"], ctx["n"], ctx["u"], ctx["method"]
        ns, nu, s = self.n_s, self.n_u, stream_ptr()
        assert dout.shape == (n, ns) and dout.is_contiguous()
        self._cur_n = n
        S = 7 if method == "dopri5" else len(TABLEAU[method]["c_sol"])
        w = self._adj_ws(n, S)
        par = self._adj_par_cur = self._adj_params_begin(w, n, S) if need_params else None
        _lib.call("nlbac_adj_pack", ctx["out"].data_ptr(), dout.data_ptr(), ns, nu, n, w["Z0"].data_ptr(), s)
        if method in ("euler", "rk4"):
            tab = TABLEAU[method]
            h = [ctx["t_end"]] * P
            self._adj_step(w, u, P, rpp, method, 0, S, h_host=h, c_out=fptr(*tab["c_sol"]),
                           keep=par and par["keep"])
            if par:
                self._adj_params_fixed(par, w, tab["c_sol"], h[0])
            res = w["Z1"]
            ctx["adjoint_info"] = None
        else:
            res = self._adj_dopri(w, u, P, rpp, par)
        du = self._buf("du", n, nu) if need_du else None
        dy0 = self._buf("dy0_adj", n, ns) if need_dy0 else None
        if du is not None or dy0 is not None:
            _lib.call("nlbac_adj_unpack", res.data_ptr(), ns, nu, n, dy0.data_ptr() if dy0 is not None else None,
                      du.data_ptr() if du is not None else None, s)
        return du, dy0

    def _adj_dopri(self, w, u, P, rpp, par):
        ctx = self.ctx
        n, S, s = ctx["n"], 7, stream_ptr()
        ctl = self._buf("adj_ctl", P, _lib.DOPRI_CTL, dtype=torch.float64)
        cp = ctl.data_ptr()
        KZ = w["KZ"]
        keep = par and par["keep"]
        ctx["ctl_seq"] = None         # (the adjoint solve's own range of stamps: _seq_next)
        # f0 = G(z(t1)) and Hairer's initial step
        self._adj_step(w, u, P, rpp, "dopri5", 0, 1, h_host=[0.0] * P, keep=keep)
        pn = self._adj_params_norm(par, 0, cp) if par else None
        self._adj_norm_control(KZ[0], None, w, u, 0, P, rpp, ctl, pn)
        self._adj_step(w, u, P, rpp, "probe", 1, 2, h_dev=cp + 8 * 6, keep=keep)             # C_H0
        pn = self._adj_params_norm(par, 1, cp) if par else None
        self._adj_norm_control(KZ[1], KZ[0], w, None, 1, P, rpp, ctl, pn)
        c_sol, c_err = self._coef("sol"), self._coef("err")
        chain = max(1, int(self.__dict__.get("_adj_chain", 1)))
        attempts = 0
        ip = self._adj_interp_fold()
        while True:
            for i in range(chain):
                if attempts:
                    # accepted and not finished: z0 <- z1, first stage <- last stage (FSAL); decided on the device
                    _lib.call("nlbac_adj_commit", cp, rpp, n, w["W"], w["Z0"].data_ptr(), w["Z1"].data_ptr(),
                              KZ[0].data_ptr(), KZ[6].data_ptr(), s)
                    if par:
                        self._adj_params_commit(par, cp)
                self._adj_step(w, u, P, rpp, "dopri5", 1, S, h_dev=cp, ctl=cp, c_out=c_sol, c_err=c_err, keep=keep, interp=ip)
                pn = self._adj_params_norm(par, 2, cp) if par else None
                self._adj_norm_control(w["ERR"], None, w, None, 2, P, rpp, ctl, pn)
                attempts += 1
            if ctx.get("adj_ctl_host"):
                self._ctl_posted(P)
            else:
                self._ctl_post(P, ctl)
            c = self._ctl_read(P) if ctx.get("ctl_pending") == P else ctl.cpu()
            if all(bool(c[p, 4] > 0) for p in range(P)):
                break
            if attempts >= self.ADJ_MAX_ATTEMPTS:
                raise _lib.NlbacError("odeint_adjoint (dopri5): max_num_steps exceeded")
            chain = 2
        used = int(max(float(c[p, 10]) for p in range(P)))       # C_NSTEPS: attempts of the slowest problem
        self._adj_chain = max(1, used)
        ctx["adjoint_info"] = [[(float(c[p, 11]), float(c[p, 2]), int(c[p, 10])) for p in range(P)]]
        # the interpolant of the last accepted step at t0 (steps are not clipped), all columns of z at once: written by
        # the attempt that finished each problem (interp), or by a launch of its own
        if not ip:
            _lib.call("nlbac_dopri_interp_fwd", w["Z0"].data_ptr(), w["Z1"].data_ptr(), KZ.data_ptr(), None, None, cp, P,
                      rpp, w["W"], w["OUT"].data_ptr(), 0, None, s)
        if par:
            self._adj_params_finish(par, cp)
        return w["OUT"]

    def _stage_backward(self, ws, st, need_dx, need_params, du, up, coef, h_host, h_dev, h_stride):
        """Un-fused backward of one stage of the control-affine field: affine_bwd -> mlp_bwd_data[f,g] ->
        rk_stage_bwd (kept as the cross-check of nlbac_node_rk_bwd)."""
        ctx = self.ctx
        P, rpp, n, u = ctx["P"], ctx["rpp"], ctx["n"], ctx["u"]
        ns, nu, S = self.n_s, self.n_u, ws.S
        s = stream_ptr()
        _lib.call("nlbac_affine_combine_bwd", ws.dK[st].data_ptr(), ws.gout[st].data_ptr(), u.data_ptr(),
                  ns, nu, n, 1.0, ws.dG[st].data_ptr() if (need_dx or need_params) else None,
                  du.data_ptr() if du is not None else None, 1, s)
        if need_dx or need_params:
            key = (st, need_params, need_dx)
            io = ws.io_bwd.get(key)
            if io is None:
                io = ws.io_bwd[key] = io_array(2)
                f, g = self.f, self.g
                for i, (net, dy, ld, acts, dz, dx) in enumerate((
                        (f, ws.dK[st], ns, ws.acts_f, ws.dz_f, ws.dXf),
                        (g, ws.dG[st], ns * nu, ws.acts_g, ws.dz_g, ws.dXg))):
                    io[i].dy, io[i].dy_ld = dy.data_ptr(), ld
                    io[i].acts = acts[:, st * ws.n:].data_ptr()
                    io[i].acts_ls = S * ws.n * net.hid
                    if need_params:
                        io[i].dz = dz[:, st * ws.n:].data_ptr()
                    if need_dx:
                        io[i].dx, io[i].dx_ld = dx.data_ptr(), net.in_dim
            _lib.call("nlbac_mlp_bwd_data", self._nets(), io, 2, n, s)
        if need_dx:
            _lib.call("nlbac_rk_stage_bwd", up.data_ptr() if up is not None else None, ws.dXf.data_ptr(),
                      ws.dXg.data_ptr(), self.f.in_dim, st, fptr(*coef) if coef else None, h_host, h_dev,
                      h_stride, P, rpp, ns, ws.dK.data_ptr(), ws.dy0.data_ptr(), 1, None, 0, s)

    def accumulate_param_grads(self, arena, slabs_per_step):
        """dW/db of f_net and g_net over every evaluated stage of every accepted
        step; step i writes slabs [i*slabs_per_step, (i+1)*slabs_per_step).
        Returns the number of slabs written."""
        ctx = self.ctx
        s = stream_ptr()
        if self.adjoint:       # the adjoint solve has integrated the parameter adjoint already: one finished vector
            par = ctx["adj_par"]
            # (under data parallelism the stage derivatives were summed over the ranks already: every rank holds the
            # global vector and hands 1 / world of it to the gradient all-reduce that follows)
            world = self.comm.world if self.comm is not None else 1
            _lib.call("nlbac_axpby", 1.0 / world, par["grad"].data_ptr(), 0.0, None, arena.n, arena.grad.data_ptr(), s)
            return 1
        n_used = 0
        for si, step in enumerate(ctx["steps"]):
            ws = step["ws"]
            S, n = ws.S, ws.n
            st0 = 0 if step["first"] or ctx["method"] != "dopri5" else 1
            rows = (S - st0) * n
            io = io_array(2)
            for i, (net, x, dy, ld, acts, dz) in enumerate((
                    (self.f, ws.Y, ws.dK, self.n_s, ws.acts_f, ws.dz_f),
                    (self.g, ws.Y, ws.dG, self.n_s * self.n_u, ws.acts_g, ws.dz_g))):
                io[i].x0, io[i].x0_dim, io[i].x0_ld = x[st0:].data_ptr(), self.n_s, self.n_s
                io[i].dy, io[i].dy_ld = dy[st0:].data_ptr(), ld
                io[i].acts = acts[:, st0 * n:].data_ptr()
                io[i].dz = dz[:, st0 * n:].data_ptr()
                io[i].acts_ls = S * n * net.hid
            if n_used + slabs_per_step > arena.n_slabs:
                n_used = self._fold_slabs(arena, n_used)
            for i in range(2):
                io[i].grad = arena.grad[n_used:].data_ptr()
            bwd_weights(self._nets(), io, 2, rows, slabs_per_step, arena.n, self.device)
            n_used += slabs_per_step
        return n_used

    def _fold_slabs(self, arena, n_used):
        """A solve with more accepted steps than the arena has gradient slabs: sum what has been written into slab 0,
        clear the rest (the skinny-layer gradients of a step sit in its first slab only) and carry on behind it — the
        one place where the order of the final slab sum differs from one slab set per step."""
        if arena.n_slabs < 2:
            raise _lib.NlbacError("arena has too few gradient slabs (%d) for this solve" % arena.n_slabs)
        tmp = self._buf("grad_fold", arena.n)
        s = stream_ptr()
        _lib.call("nlbac_reduce_slabs", tmp.data_ptr(), arena.grad.data_ptr(), n_used, arena.n, arena.n, s)
        _lib.call("nlbac_axpby", 1.0, tmp.data_ptr(), 0.0, None, arena.n, arena.grad.data_ptr(), s)
        _lib.call("nlbac_fill", arena.grad[1:].data_ptr(), 0.0, (arena.n_slabs - 1) * arena.n, s)
        return 1


# ---------------------------------------------------------------------------
# Non-affine field  dx/dt = net([x, c])  with carried inputs c = (u, t, ...) constant over the step
# (SimulatedCars: C/sac_cbf_clf/model.py:179-205).  Same RK machinery, stage by stage on nlbac_mlp_*.
# ---------------------------------------------------------------------------
class _ConcatStepWS:
    ADOPT = ("K", "Y", "err", "acts")

    def __init__(self, solver, n, S, store):
        dev, ns, nc = solver.device, solver.n_s, solver.n_u
        net = solver.net
        self.n, self.S = n, S
        self._store = store
        z = self._store.zeros
        self.K = z(S, n, ns)
        self.Y = z(S, n, ns)
        # a rollout that is only differentiated w.r.t. its inputs keeps ReLU mask words instead of the activations, where
        # the fused kernels can (the register-resident ones: nlbac_concat_rk_mask_words)
        words = _lib.load().nlbac_concat_rk_mask_words(C.byref(net.desc)) if (solver.fused and not solver.keep_acts) else 0
        self.bits = words > 0
        if self.bits:
            self.wa = words
            self.acts = self._store.zeros(net.n_layers - 1, S * n, words, dtype=torch.int32)
        else:
            self.wa = net.hid
            self.acts = z(net.n_layers - 1, S * n, net.hid)
        self.y1 = z(n, ns)
        self.err = z(n, ns)
        self.io_fwd, self.io_bwd = {}, {}
        self._bwd = None
        # a normalised field keeps the normalised net inputs of every stage for the first layer's weight gradient
        self.Xn = z(S, n, net.in_dim) if solver.norm is not None else None

    def bwd(self, solver):
        if self._bwd is None:
            dev, ns, nc, n, S = solver.device, solver.n_s, solver.n_u, self.n, self.S
            net = solver.net
            z = self._store.zeros
            self.dK = z(S, n, ns)
            self.dz = z(net.n_layers - 1, S * n, net.hid)
            self.dX = z(n, net.in_dim)
            self.dy0 = z(n, ns)
            self.dy1 = z(n, ns)
            self.c_rep = z(S * n, nc)      # carried inputs repeated per stage (first-layer weight gradients)
            self.dyn = z(S, n, ns) if solver.norm is not None else None      # d/d(net output) = dK * out_std
            self._bwd = True
        return self


class ConcatNodeSolver(AffineNodeSolver):
    """``u`` here is the (n, n_carry) block of carried inputs; ``backward`` returns its gradient."""
    STEP_WS = _ConcatStepWS

    def __init__(self, node, device):
        self.node, self.net = node, node.net_handle
        self.f = self.g = self.net            # (base-class bookkeeping only)
        self.n_s, self.n_u = node.n_s, node.n_carry
        self.device = torch.device(device)
        self._ws, self._scratch = {}, {}
        self.nfe = 0
        self._net_arr, self._coefs, self._children = None, {}, {}
        # one nlbac_concat_rk_fwd / _bwd launch per RK step (nets of <= 128 hidden units; wider ones run stage by stage)
        self.fused = self.net.hid <= 128
        self.keep_acts = True
        self.stats = dict(solves=0, single_step=0, multi_attempt=0, split=0)
        self.comm = None
        self.row_groups = 1
        self.adjoint = False
        self.generation = 0
        self.device_loop = True
        # input normalisation / output de-normalisation lives inside the fused step kernels only
        self.norm = node.norm_device() if getattr(node, "normalized", False) else None
        if self.norm is not None and not self.fused:
            raise _lib.NlbacError("a normalised NODE needs the fused step kernels (hidden width <= 128)")

    def _nets(self):
        if self._net_arr is None:
            self._net_arr = mlp_array([self.net.desc])
        return self._net_arr

    def _interp_nets(self):
        return C.byref(self.net.desc), None

    def _begin_persistent(self, ws0, ch, y0, u, P, rpp):
        return False

    # -- continuous adjoint (odeint_adjoint) of the single-net field ---------------------------------------------
    # The base class drives the solve (initial step, attempts, mixed norm, commit, interpolation, parameter-adjoint
    # quadrature); what differs is one RK step of the augmented system z = [y | a_y | a_c] and the stage derivative of
    # the parameter adjoint: stage by stage on the MLP entry points (nlbac_concat_adj_in -> nlbac_mlp_fwd ->
    # nlbac_mlp_bwd_data -> nlbac_concat_adj_out), the RK combinations by nlbac_rk_combine on the w-wide rows.
    def _adj_scratch(self, n, S):
        net, ns, nc = self.net, self.n_s, self.n_u
        return dict(ZS=self._buf("cadj_ZS", n, 2 * ns + nc), Xin=self._buf("cadj_Xin", n, net.in_dim),
                    Ay=self._buf("cadj_Ay", n, ns), f=self._buf("cadj_f", n, ns), dX=self._buf("cadj_dX", n, net.in_dim),
                    acts=self._buf("cadj_acts", net.n_layers - 1, n, net.hid))

    def _adj_fused(self):
        """One nlbac_concat_adj_step launch per attempted step (the reference's depth at widths 64 / 100 / 128);
        ``adj_fused = False`` keeps the stage-by-stage launches (other shapes; the cross-check)."""
        f = self.__dict__.get("adj_fused")
        if f is None:
            f = self.adj_fused = bool(_lib.load().nlbac_concat_adj_step_ok(C.byref(self.net.desc)))
        return f

    def _adj_interp_fold(self):
        return self._adj_fused() and self._interp_fold_on()

    def _adj_step(self, w, u, P, rpp, method, st0, st1, h_host=None, h_dev=None, ctl=None, c_out=None, c_err=None,
                  keep=None, interp=False):
        """(Stage by stage: problems whose solve is done are recomputed to the same values — their control block, z0
        and first stage no longer change — instead of being skipped; the fused launch leaves their rows alone.)"""
        if self._adj_fused():
            beta, S = self._beta(method)
            k = keep or {}
            dp = lambda t: t.data_ptr() if t is not None else None
            _lib.call("nlbac_concat_adj_step", C.byref(self.net.desc), u.data_ptr(), P, rpp, st0, st1, S, beta,
                      c_out, len(c_out) if c_out is not None else 0, c_err, len(c_err) if c_err is not None else 0,
                      fptr(*h_host) if h_host is not None else None, h_dev, _lib.DOPRI_CTL if h_dev else 0, ctl,
                      w["Z0"].data_ptr(), w["KZ"].data_ptr(), w["Z1"].data_ptr() if c_out is not None else None,
                      w["ERR"].data_ptr() if c_err is not None else None,
                      self.norm.data_ptr() if self.norm is not None else None, dp(k.get("Xin")), dp(k.get("Ay")),
                      dp(k.get("acts")), k.get("ls", 0), dp(k.get("dz")), w["OUT"].data_ptr() if interp else None,
                      self.ctx["t_end"], stream_ptr())
            self.nfe += st1 - st0
            if keep:
                for st in range(st0, st1):
                    self._adj_stage_dw(self._adj_par_cur, st)
            return
        n, W, ns, nc, net, s = P * rpp, w["W"], self.n_s, self.n_u, self.net, stream_ptr()
        rows = TABLEAU[method]["beta"]
        S = len(rows) + 1
        sc = self._adj_scratch(n, S)
        hh = fptr(*h_host) if h_host is not None else None
        stride = _lib.DOPRI_CTL if h_dev else 0
        norm = self.norm.data_ptr() if self.norm is not None else None
        KZ = w["KZ"]
        for st in range(st0, st1):
            if st == 0:
                ZS = w["Z0"]
            else:
                ZS = sc["ZS"]
                _lib.call("nlbac_rk_combine", w["Z0"].data_ptr(), KZ.data_ptr(), st, fptr(*rows[st - 1]), hh, h_dev, stride,
                          P, rpp, W, ZS.data_ptr(), s)
            if keep:      # the parameter adjoint's quadrature reads every stage's net inputs / cotangents / activations
                Xin, Ay = keep["Xin"][st], keep["Ay"][st]
                acts, dz, ls = keep["acts"][:, st * n:], keep["dz"][:, st * n:], keep["ls"]
            else:
                Xin, Ay, acts, dz, ls = sc["Xin"], sc["Ay"], sc["acts"], None, n * net.hid
            _lib.call("nlbac_concat_adj_in", ZS.data_ptr(), W, u.data_ptr(), ns, nc, norm, n, Xin.data_ptr(), Ay.data_ptr(), s)
            io = io_array(1)
            io[0].x0, io[0].x0_dim, io[0].x0_ld = Xin.data_ptr(), net.in_dim, net.in_dim
            io[0].y, io[0].y_ld = sc["f"].data_ptr(), ns
            io[0].acts, io[0].acts_ls = acts.data_ptr(), ls
            _lib.call("nlbac_mlp_fwd", self._nets(), io, 1, n, s)
            io[0].dy, io[0].dy_ld = Ay.data_ptr(), ns
            io[0].dx, io[0].dx_ld = sc["dX"].data_ptr(), net.in_dim
            if dz is not None:
                io[0].dz = dz.data_ptr()
            _lib.call("nlbac_mlp_bwd_data", self._nets(), io, 1, n, s)
            _lib.call("nlbac_concat_adj_out", sc["f"].data_ptr(), sc["dX"].data_ptr(), ns, nc, norm, n, W, KZ[st].data_ptr(), s)
            if keep:
                self._adj_stage_dw(self._adj_par_cur, st)
        if c_out is not None:
            _lib.call("nlbac_rk_combine", w["Z0"].data_ptr(), KZ.data_ptr(), len(c_out), c_out, hh, h_dev, stride, P, rpp, W,
                      w["Z1"].data_ptr(), s)
        if c_err is not None:
            _lib.call("nlbac_rk_combine", None, KZ.data_ptr(), len(c_err), c_err, hh, h_dev, stride, P, rpp, W,
                      w["ERR"].data_ptr(), s)
        self.nfe += st1 - st0

    def _adj_params_begin(self, w, n, S):
        ctx = self.ctx
        assert ctx["P"] == 1, "parameter gradients are only taken on single-problem solves"
        key = ("adj_par", n, S)
        pool = self._scratch.setdefault(n, {})
        par = pool.get(key)
        if par is None:
            net, ns, dev = self.net, self.n_s, self.device
            z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)
            arena = net.arena
            NP = arena.n
            keep = dict(Xin=z(S, n, net.in_dim), Ay=z(S, n, ns), acts=z(net.n_layers - 1, S * n, net.hid),
                        dz=z(net.n_layers - 1, S * n, net.hid), ls=S * n * net.hid)
            segs = [(arena.offset_of[id(p)], p.numel()) for p in self.node.parameters()]
            par = pool[key] = dict(
                keep=keep, NP=NP, K=z(S, NP), th0=z(NP), th1=z(NP), out=z(NP), slabs=z(self.ADJ_SUB_SLABS, NP),
                seg_off=torch.tensor([o for o, _ in segs], dtype=torch.int32, device=dev),
                seg_len=torch.tensor([l for _, l in segs], dtype=torch.int32, device=dev), n_seg=len(segs),
                pseg=z(2 * len(segs)), ticket=z(1, dtype=torch.int32), pnorm=z(2), io={}, n=n, S=S, w=w)
        _lib.call("nlbac_fill", par["th0"].data_ptr(), 0.0, par["NP"], stream_ptr())
        par["grad"] = None
        return par

    def _adj_stage_dw(self, par, st):
        """K_theta[st] = sum over the rows of stage ``st`` of (d net / d theta)^T (a_y out_sig) at the stage's
        (normalised) inputs: nlbac_mlp_bwd_weights on what the step kept of that stage, then the slab sum."""
        k, n, S, net = par["keep"], par["n"], par["S"], self.net
        io = par["io"].get(st)
        if io is None:
            io = par["io"][st] = io_array(1)
            io[0].x0, io[0].x0_dim, io[0].x0_ld = k["Xin"][st].data_ptr(), net.in_dim, net.in_dim
            io[0].dy, io[0].dy_ld = k["Ay"][st].data_ptr(), self.n_s
            io[0].acts, io[0].dz = k["acts"][:, st * n:].data_ptr(), k["dz"][:, st * n:].data_ptr()
            io[0].acts_ls = S * n * net.hid
            io[0].grad = par["slabs"].data_ptr()
        bwd_weights(self._nets(), io, 1, n, self.ADJ_SUB_SLABS, par["NP"], self.device)
        _lib.call("nlbac_reduce_slabs", par["K"][st].data_ptr(), par["slabs"].data_ptr(), self.ADJ_SUB_SLABS,
                  par["NP"], par["NP"], stream_ptr())
        if self.comm is not None and self.comm.world > 1:
            self.comm.all_reduce_(par["K"][st])       # (see AffineNodeSolver._adj_stage_dw)

    def _rk_fused(self, ws, y0, u, P, rpp, method, st0, st1, h_host=None, h_dev=None, c_out=None, out=None,
                  c_err=None, err=None, save_acts=True, chain=None):
        beta, S = self._beta(method)
        save_acts = save_acts and not self.adjoint       # (the adjoint re-computes every stage it differentiates)
        _lib.call("nlbac_concat_rk_fwd", C.byref(self.net.desc), y0.data_ptr(), u.data_ptr(), P, rpp, st0, st1, S, beta,
                  c_out, len(c_out) if c_out is not None else 0, c_err, len(c_err) if c_err is not None else 0,
                  fptr(*h_host) if h_host is not None else None, h_dev, _lib.DOPRI_CTL if h_dev else 0,
                  ws.K.data_ptr(), ws.Y.data_ptr(), ws.acts.data_ptr() if save_acts else None,
                  ws.S * P * rpp * ws.wa, 1 if ws.bits else 0, out.data_ptr() if out is not None else None,
                  err.data_ptr() if err is not None else None,
                  self.norm.data_ptr() if self.norm is not None else None,
                  ws.Xn.data_ptr() if (self.norm is not None and save_acts and self.keep_acts) else None,
                  C.byref(chain) if chain is not None else None, stream_ptr())
        self.nfe += st1 - st0

    def _rk_fused_bwd(self, ws, u, P, rpp, method, first_eval, need_dy0, need_params, h_host, h_dev, h_stride, top_up,
                      du, last, chain=None, back_idx=0):
        beta_arr, _ = self._beta(method)
        S = ws.S
        _lib.call("nlbac_concat_rk_bwd", C.byref(self.net.desc), P, rpp, S, 0 if first_eval else 1, S,
                  1 if need_dy0 else 0, beta_arr, h_host, h_dev, h_stride, ws.acts.data_ptr(),
                  S * ws.n * ws.wa, 1 if ws.bits else 0, ws.dz.data_ptr() if need_params else None, ws.dK.data_ptr(),
                  top_up.data_ptr() if top_up is not None else None, ws.dy0.data_ptr(), 1,
                  du.data_ptr() if du is not None else None, 0 if last else 1,
                  self.norm.data_ptr() if self.norm is not None else None,
                  ws.dyn.data_ptr() if (self.norm is not None and need_params) else None,
                  C.byref(chain) if chain is not None else None, back_idx, stream_ptr())

    def _eval_io(self, x, k_out, c, acts=None, ls=0):
        io = io_array(1)
        io[0].x0, io[0].x0_dim, io[0].x0_ld = x.data_ptr(), self.n_s, self.n_s
        io[0].x1, io[0].x1_dim, io[0].x1_ld = c.data_ptr(), self.n_u, self.n_u
        io[0].y, io[0].y_ld = k_out.data_ptr(), self.n_s
        if acts is not None:
            io[0].acts, io[0].acts_ls = acts.data_ptr(), ls
        return io

    def _eval(self, x, u, n, k_out, g_out, io=None):
        if io is None:
            io = self._eval_io(x, k_out, u)
        _lib.call("nlbac_mlp_fwd", self._nets(), io, 1, n, stream_ptr())
        self.nfe += 1

    def _probe_eval(self, ytmp, u, n, ktmp, gtmp):
        self._eval(ytmp, u, n, ktmp, None)

    def _stage_eval(self, ws, st, u):
        n, S = ws.n, ws.S
        io = ws.io_fwd.get(st)
        if io is None:
            io = ws.io_fwd[st] = self._eval_io(ws.Y[st], ws.K[st], u, ws.acts[:, st * n:], S * n * self.net.hid)
        self._eval(ws.Y[st], u, n, ws.K[st], None, io)

    def _stage_backward(self, ws, st, need_dx, need_params, du, up, coef, h_host, h_dev, h_stride):
        ctx = self.ctx
        P, rpp, n = ctx["P"], ctx["rpp"], ctx["n"]
        ns, S, net = self.n_s, ws.S, self.net
        s = stream_ptr()
        key = (st, need_params)
        io = ws.io_bwd.get(key)
        if io is None:
            io = ws.io_bwd[key] = io_array(1)
            io[0].dy, io[0].dy_ld = ws.dK[st].data_ptr(), ns
            io[0].acts, io[0].acts_ls = ws.acts[:, st * n:].data_ptr(), S * n * net.hid
            if need_params:
                io[0].dz = ws.dz[:, st * n:].data_ptr()
            io[0].dx, io[0].dx_ld = ws.dX.data_ptr(), net.in_dim
        _lib.call("nlbac_mlp_bwd_data", self._nets(), io, 1, n, s)
        # dY = [up] + dX[:, :n_s] ; dy0 += dY ; dK[j] += beta h dY ; d carried += dX[:, n_s:]
        _lib.call("nlbac_rk_stage_bwd", up.data_ptr() if up is not None else None, ws.dX.data_ptr(), None,
                  net.in_dim, st, fptr(*coef) if coef else None, h_host, h_dev, h_stride, P, rpp, ns,
                  ws.dK.data_ptr(), ws.dy0.data_ptr(), 1, du.data_ptr() if du is not None else None, self.n_u, s)

    def accumulate_param_grads(self, arena, slabs_per_step):
        ctx = self.ctx
        if self.adjoint:       # the adjoint solve has integrated the parameter adjoint already (see the base class)
            return AffineNodeSolver.accumulate_param_grads(self, arena, slabs_per_step)
        n_used = 0
        for si, step in enumerate(ctx["steps"]):
            ws = step["ws"]
            S, n = ws.S, ws.n
            st0 = 0 if step["first"] or ctx["method"] != "dopri5" else 1
            rows = (S - st0) * n
            io = io_array(1)
            if self.norm is not None:      # the net saw normalised inputs and its output is scaled: use what the kernels kept
                io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.Xn[st0:].data_ptr(), self.net.in_dim, self.net.in_dim
                io[0].dy, io[0].dy_ld = ws.dyn[st0:].data_ptr(), self.n_s
            else:
                ws.c_rep.view(S, n, self.n_u).copy_(ctx["u"].unsqueeze(0).expand(S, n, self.n_u))
                io[0].x0, io[0].x0_dim, io[0].x0_ld = ws.Y[st0:].data_ptr(), self.n_s, self.n_s
                io[0].x1, io[0].x1_dim, io[0].x1_ld = ws.c_rep[st0 * n:].data_ptr(), self.n_u, self.n_u
                io[0].dy, io[0].dy_ld = ws.dK[st0:].data_ptr(), self.n_s
            io[0].acts = ws.acts[:, st0 * n:].data_ptr()
            io[0].dz = ws.dz[:, st0 * n:].data_ptr()
            io[0].acts_ls = S * n * self.net.hid
            if n_used + slabs_per_step > arena.n_slabs:
                n_used = self._fold_slabs(arena, n_used)
            io[0].grad = arena.grad[n_used:].data_ptr()
            bwd_weights(self._nets(), io, 1, rows, slabs_per_step, arena.n, self.device)
            n_used += slabs_per_step
        return n_used


# ---------------------------------------------------------------------------
# torchdiffeq-shaped entry (SURVEY.md §8b "Solver entry"): the call the reference makes at
# U/sac_cbf_clf/sac_cbf_clf.py:453,577 and U/sac_cbf_clf/model.py:252 —
#     odeint(model, cat(state, action), tensor([0, dt]), method=..., atol=1e-7, rtol=1e-5)[-1][:, :n_s]
# — on the device kernels, differentiable w.r.t. y0 and the model's parameters through torch.autograd.
# The agent itself drives the solvers directly (no autograd graph); this entry is for reference-shaped code.
# ---------------------------------------------------------------------------
def _solver_of(func, adjoint=False):
    """The (cached) solver of a ``NeuralODEModel`` of this build; a model that is not part of an agent gets its own
    parameter arena on the current device.  The adjoint entry keeps a solver of its own (it saves nothing in forward)."""
    from .sac_cbf_clf.model import NeuralODEModel
    if not isinstance(func, NeuralODEModel):
        raise TypeError("nlbac_amd.odeint integrates this build's NeuralODEModel (its field runs as HIP kernels); "
                        "got %s" % type(func).__name__)
    key = "_odeint_solver_adj" if adjoint else "_odeint_solver"
    sv = func.__dict__.get(key)
    if sv is None:
        handles = func.device_handles()
        sv = (AffineNodeSolver if func.affine else ConcatNodeSolver)(func, handles[0].arena.device)
        sv.keep_acts = True                  # parameter gradients need the pre-activation gradients of every stage
        sv.adjoint = bool(adjoint)
        func.__dict__[key] = sv
    return sv


class _OdeintFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, func, method, dt, atol, rtol, adjoint, y0, *params):
        sv = _solver_of(func, bool(adjoint))
        ns, nu = sv.n_s, sv.n_u
        assert y0.dim() == 2 and y0.shape[1] == ns + nu, "y0 must be (batch, %d)" % (ns + nu)
        y0 = y0.detach().float().contiguous()
        x0, u = y0[:, :ns].contiguous(), y0[:, ns:].contiguous()
        x1 = sv.forward(x0, u, 1, y0.shape[0], method, dt, atol, rtol)
        ctx.func, ctx.sv, ctx.n_params = func, sv, len(params)
        ctx.solve_id = sv.stats["solves"]
        ctx.with_params = adjoint != "no-params"
        return torch.stack([y0, torch.cat([x1, u], dim=1)])

    @staticmethod
    def backward(ctx, g):
        sv, func = ctx.sv, ctx.func
        assert sv.stats["solves"] == ctx.solve_id, \
            "odeint: backward must run before the next solve with the same model (the solver keeps one solve's state)"
        ns = sv.n_s
        g = g.float()
        need_p = any(ctx.needs_input_grad[7:]) and ctx.with_params
        du, dy0 = sv.backward(g[1][:, :ns].contiguous(), need_du=True, need_params=need_p, need_dy0=True)
        gy0 = g[0] + torch.cat([dy0, du + g[1][:, ns:]], dim=1) if ctx.needs_input_grad[6] else None
        gp = [None] * ctx.n_params
        if need_p:
            arena = func.device_handles()[0].arena
            n_steps = max(1, len(sv.ctx.get("steps") or [None]))
            used = sv.accumulate_param_grads(arena, max(1, arena.n_slabs // n_steps))
            flat = torch.empty(arena.n, dtype=torch.float32, device=arena.device)
            _lib.call("nlbac_reduce_slabs", flat.data_ptr(), arena.grad.data_ptr(), used, arena.n, arena.n, stream_ptr())
            gp = []
            for p in func.parameters():
                off = arena.offset_of[id(p)]
                gp.append(flat[off:off + p.numel()].view(p.shape))
        return (None, None, None, None, None, None, gy0, *gp)


def _odeint(func, y0, t, method, atol, rtol, adjoint, options):
    if options:
        raise TypeError("odeint: unsupported options %s" % sorted(options))
    from .sac_cbf_clf.model import NeuralODEModel
    if not isinstance(func, NeuralODEModel):
        raise TypeError("nlbac_amd.odeint integrates this build's NeuralODEModel (its field runs as HIP kernels); "
                        "got %s" % type(func).__name__)
    t = torch.as_tensor(t)
    if t.numel() != 2:
        raise NotImplementedError("odeint: the reference only ever integrates over t = [0, dt]; got %d time points"
                                  % t.numel())
    dt = float(t[1]) - float(t[0])
    func.refresh_device_weights()
    return _OdeintFunction.apply(func, method, dt, float(atol), float(rtol), adjoint, y0, *func.parameters())


def odeint(func, y0, t, *, method="dopri5", atol=1e-7, rtol=1e-5, **options):
    """``torchdiffeq.odeint`` for this build's NODE models on ``t = [t0, t1]``: returns ``stack([y0, y(t1)])`` with the
    carried control columns passed through, differentiable w.r.t. ``y0`` and ``func.parameters()``.  ``method`` is
    ``'euler'`` / ``'rk4'`` (one step over the interval, torchdiffeq's fixed-grid semantics) or ``'dopri5'``.
    The packed MFMA copies of the weights are refreshed first, so a ``torch.optim`` step on ``func.parameters()``
    between calls is picked up."""
    return _odeint(func, y0, t, method, atol, rtol, False, options)


def odeint_adjoint(func, y0, t, *, method="dopri5", atol=1e-7, rtol=1e-5, adjoint_params=None, **options):
    """``torchdiffeq.odeint_adjoint`` (0.2.3) on ``t = [t0, t1]``: the same forward solve, keeping only y(t1); the
    backward integrates the augmented state [y, adj_y, adj_params] from t1 to t0 with the same method and tolerances
    (``adjoint_rtol`` / ``adjoint_atol`` / ``adjoint_method`` default to the forward's in torchdiffeq; only those
    defaults are offered) under torchdiffeq's default mixed adjoint norm.  Memory does not grow with the number of
    steps, and the gradient equals direct back-propagation only to solver tolerance.  ``adjoint_params``: ``None`` —
    every parameter of ``func`` (torchdiffeq's default, ``find_parameters``); ``()`` — no parameter adjoint (it then
    also stays out of the step-size norm).  The reference never calls this; semantics follow the
    published algorithm."""
    if adjoint_params is not None and len(tuple(adjoint_params)) != 0:
        raise NotImplementedError("odeint_adjoint: adjoint_params is None (all of func's parameters) or ()")
    return _odeint(func, y0, t, method, atol, rtol, "no-params" if adjoint_params is not None else "params", options)
