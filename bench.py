"""Headline benchmark: ODE-integrate+update samples/s (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one ``update_parameters``-equivalent pass of the hot path over one
synthetic minibatch already resident in HBM: NODE rollouts (primary + backup
controller), actor / twin-Q / Lyapunov forward+backward, CBF/CLF augmented-
Lagrangian loss, Adam steps, Polyak update, and — every 10th step — the NODE
regression step on 32768 transitions.  Workload at N=1:
BASELINE.json configs[1] (Unicycle, batch 4096, dopri5).  N > 1: weak scaling by
default (every rank runs its own 4096-row shard of a 4096*N global batch; the
headline ``value``), and the same line carries ``strong`` — the metric's literal
split, ONE 4096-row global batch sharded over the N ranks — measured right after.
``--global-batch G`` makes the strong split the headline instead.
"""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import nlbac_amd  # noqa: F401
from nlbac_amd import _lib, synth
from nlbac_amd.envspec import make_env

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, chip-level parameters
NODE_FIT_ROWS, NODE_FIT_INTERVAL, REPLAY_ROWS = 32768, 10, 65536
GAMMA_B = {"QuadrotorLike": 1.0, "Unicycle": 50.0, "SimulatedCars": 0.5, "UnicycleBarrier": 5.0, "Pvtol": 0.8, "PvtolBarrier": 1.0}       # the reference README's run commands


# what each --env is in terms of BASELINE.json (the learned-barrier copies are the reference's NU / NP agents on a
# control-affine NODE: NOT configs[4], whose Quadrotor-like task is --env QuadrotorLike)
WORKLOAD_NOTE = {"Unicycle": "BASELINE.json configs[1] at B=4096 dopri5", "SimulatedCars": "BASELINE.json configs[2] at B=8192 rk4",
                 "Pvtol": "BASELINE.json configs[3] at B=16384 dopri5 --adjoint",
                 "UnicycleBarrier": "the reference's learned-barrier Unicycle copy NU; no BASELINE config",
                 "PvtolBarrier": "the reference's learned-barrier Pvtol copy NP; no BASELINE config",
                 "QuadrotorLike": "BASELINE.json configs[4] at B=32768: Quadrotor-like SYNTHETIC task — the reference has "
                                  "no code for it (empty submodule), parity vs the oracle's reading of README.md:190-192 only"}


class Args:
    gamma, gamma_b, tau, alpha, lr = 0.99, 50.0, 0.005, 0.2, 3e-4
    hidden_size, target_update_interval, Lagrangian_multiplier_update_interval = 256, 1, 8
    automatic_entropy_tuning, policy, seed, cuda = True, "Gaussian", 0, True

    def __init__(self, batch_size):
        self.batch_size = batch_size


def replay_rows(agent, tr):
    """The synthetic replay as minibatch-layout rows (the agent's HBM row layout), built once on the host."""
    return agent._rows_from_host(tuple(tr[f] for f in (synth.FIELDS_BARRIER if agent.lay.sig is not None else synth.FIELDS)))


# ---- algorithmic FLOPs of one MLP launch (real dims, 1 MAC = 2 FLOP) ------------------------------
def _layer_macs(net):
    dims = [net.in_dim] + [net.hid] * (net.n_layers - 1) + [net.out_dim]
    return [dims[i] * dims[i + 1] for i in range(net.n_layers)]


def launch_flops(name, args):
    if name == "nlbac_node_rk_fwd":       # (f, g, y0, u, P, rpp, st0, st1, ...): stages x rows x NODE eval
        f, g, P, rpp, st0, st1 = args[0]._obj, args[1]._obj, args[4], args[5], args[6], args[7]
        return 2 * P * rpp * (st1 - st0) * (sum(_layer_macs(f)) + sum(_layer_macs(g)) + g.out_dim)
    if name == "nlbac_node_rk_bwd":       # (f, g, u, G, P, rpp, S, st_lo, st_hi, dx0, ...): data backward
        f, g, P, rpp, st_lo, st_hi, dx0 = args[0]._obj, args[1]._obj, args[4], args[5], args[7], args[8], args[9]
        per = 0
        for net in (f, g):
            m = _layer_macs(net)
            per += m[-1] + sum(m[1:-1]) + m[0]
        stages = (st_hi - st_lo) - (0 if (dx0 or st_lo > 0) else 1)
        return 2 * P * rpp * stages * (per + 2 * g.out_dim)
    if name == "nlbac_node_adj_step":     # (f, g, u, P, rpp, st_lo, st_hi, ...): every stage = field eval + its data backward
        f, g, P, rpp, st_lo, st_hi = args[0]._obj, args[1]._obj, args[3], args[4], args[5], args[6]
        per = 0
        for net in (f, g):
            m = _layer_macs(net)
            per += sum(m) + m[-1] + sum(m[1:-1]) + m[0]
        return 2 * P * rpp * (st_hi - st_lo) * (per + 3 * g.out_dim)
    if name == "nlbac_concat_adj_step":   # (net, c, P, rpp, st_lo, st_hi, ...): every stage = net eval + its data backward
        net, P, rpp, st_lo, st_hi = args[0]._obj, args[2], args[3], args[4], args[5]
        m = _layer_macs(net)
        return 2 * P * rpp * (st_hi - st_lo) * (sum(m) + m[-1] + sum(m[1:-1]) + m[0])
    if name == "nlbac_concat_rk_fwd":     # (net, y0, c, P, rpp, st0, st1, ...): stages x rows x NODE eval
        net, P, rpp, st0, st1 = args[0]._obj, args[3], args[4], args[5], args[6]
        return 2 * P * rpp * (st1 - st0) * sum(_layer_macs(net))
    if name == "nlbac_concat_rk_bwd":     # (net, P, rpp, S, st_lo, st_hi, dx0, ...): data backward
        net, P, rpp, st_lo, st_hi, dx0 = args[0]._obj, args[1], args[2], args[4], args[5], args[6]
        m = _layer_macs(net)
        stages = (st_hi - st_lo) - (0 if (dx0 or st_lo > 0) else 1)
        return 2 * P * rpp * stages * (m[-1] + sum(m[1:-1]) + m[0])
    nets, io, n_nets, B = args[0], args[1], args[2], args[3]
    total = 0
    for i in range(n_nets):
        macs = _layer_macs(nets[i])
        if name == "nlbac_mlp_bwd_data":
            m = macs[-1] + sum(macs[1:-1]) + (macs[0] if io[i].dx else 0)
        else:
            m = sum(macs)
        total += 2 * B * m
    return total


class FlopCounter:
    """Executed algorithmic FLOP of the MFMA kernel families, counted on the host as the launches go by — no events, no
    device work: what the replay of the timed region runs under (identical updates, so its count IS the timed region's)."""
    FAM = {"nlbac_mlp_bwd_data_head": "nlbac_mlp_bwd_data", "nlbac_mlp_fwd_gauss": "nlbac_mlp_fwd"}

    def __init__(self):
        self.flop, self.launches, self._cache, self._orig = 0, 0, {}, _lib.call

    def __enter__(self):
        names = set(KernelTimer.NAMES)

        def call(name, *args):
            fam = self.FAM.get(name, name)
            if fam in names:
                # (descriptor arrays are built once per plan: their identity + the integer arguments name the launch)
                key = (fam,) + tuple(a if isinstance(a, int) else id(a) for a in args[:10])
                fl = self._cache.get(key)
                if fl is None:
                    fl = self._cache[key] = launch_flops(fam, args)
                self.flop += fl
                self.launches += 1
            self._orig(name, *args)
        _lib.call = call
        return self

    def __exit__(self, *a):
        _lib.call = self._orig


class KernelTimer:
    """HIP events (torch.cuda.Event on the launch stream) around every launch of the MLP kernels."""
    NAMES = ("nlbac_mlp_fwd", "nlbac_mlp_bwd_data", "nlbac_mlp_bwd_weights", "nlbac_node_rk_fwd", "nlbac_node_rk_bwd",
             "nlbac_node_adj_step", "nlbac_concat_rk_fwd", "nlbac_concat_rk_bwd", "nlbac_concat_adj_step")

    def __init__(self):
        self.records = {n: [] for n in self.NAMES}
        self._orig = _lib.call

    def __enter__(self):
        def call(name, *args):
            fam = {"nlbac_mlp_bwd_data_head": "nlbac_mlp_bwd_data", "nlbac_mlp_fwd_gauss": "nlbac_mlp_fwd"}.get(name)
            if fam:      # the MLP launches that also produce their dL/dy / draw the policy sample: same families
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._orig(name, *args)
                e1.record()
                self.records[fam].append((e0, e1, launch_flops(fam, args)))
            elif name in self.records:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._orig(name, *args)
                e1.record()
                self.records[name].append((e0, e1, launch_flops(name, args)))
            else:
                self._orig(name, *args)
        _lib.call = call
        return self

    def __exit__(self, *a):
        _lib.call = self._orig

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, recs in self.records.items():
            if recs:
                ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
                fl = sum(f for _, _, f in recs)
                out[n] = dict(launches=len(recs), ms=ms, flops=fl, avg_us=1e3 * ms / len(recs),
                              tflops=fl / (ms * 1e-3) / 1e12)
        return out


def pmc_traffic(kernel, env_name, solver, B):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 --pmc passes of this same workload
    (tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 read-side x2 correction applied in
    tools/pmc_summary.py).  Counters cannot be collected from inside the timed process, so this is read from
    profiles/; None when no pass exists for the workload."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for r in ("r04", "r03", "r02", "r01"):
        path = os.path.join(here, "%s_pmc_hbm_traffic_%s_%s_B%d.json" % (r, env_name.lower(), solver, B))
        if os.path.exists(path):
            k = json.load(open(path))["kernels"].get(kernel.split("+")[0])
            return None if k is None else k["hbm_bytes_per_launch"]
    return None


PEAK_HBM_BYTES_PER_S = 8.0e12      # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def algorithmic_bytes_per_update(agent, B):
    """SURVEY.md §8(d) "Algorithmic bytes per sample-update", from the agent's real sizes: the minibatch rows read
    once; every trained parameter read with its gradient and Adam moments and written back with them (28 B), every
    target parameter read and written plus the live one read (12 B), the NODE's weights read once by the rollouts
    (4 B); plus, amortised over the fit interval, the NODE-fit rows and the NODE's own optimiser step."""
    row = 4 * agent.lay.width
    trained = sum(a.n for a in agent.arenas if a is not agent.ar_n)
    target = agent.ar_c.n
    node = agent.ar_n.n
    fit_row = 4 * (2 * agent.lay.obs_dim + agent.lay.act_dim)
    per_update = B * row + 28 * trained + 12 * target + 4 * node
    fit = (NODE_FIT_ROWS * fit_row + 28 * node) / NODE_FIT_INTERVAL
    return dict(total=per_update + fit, minibatch=B * row, optimiser=28 * trained + 12 * target + 4 * node, node_fit=fit)


def pmc_update_traffic(env_name, solver, B, adjoint=False):
    """HBM bytes per update (all kernels) from the committed counter passes of this workload (tools/gpu_pmc.sh writes
    ``per_update_bytes`` from a lean run of exactly warmup + steps updates), or None."""
    tag = "%s_%s_B%d%s" % (env_name.lower(), solver, B, "_adjoint" if adjoint else "")
    for r in ("r04", "r03", "r02"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "%s_pmc_hbm_traffic_%s.json" % (r, tag))
        if os.path.exists(path):
            return json.load(open(path)).get("per_update_bytes")
    return None


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    """CPU threads this process may really use: cgroup quota, affinity, capped at the GPU box's 16-core share."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(B, solver, env_name="Unicycle", seed=0, adjoint=False):
    """The oracle (CPU restatement pinned to the reference) timed on this box's host cores on a
    bounded sample of the same workload (about 12 s of CPU work): updates at batch B and NODE fits on 32768 rows."""
    from oracle import nlbac_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu_baseline: oracle on %d host threads" % cores)
    env = make_env(env_name, seed)
    W = synth.agent_weights(env_name, 256, seed)
    oargs = O.Args(batch_size=B, hidden_size=256, seed=seed)
    oargs.gamma_b = GAMMA_B[env_name]
    agent = O.make_oracle(env, oargs, W, solver=solver, adjoint=adjoint)
    tr = synth.transitions(env_name, REPLAY_ROWS, seed=1, env=env)
    fields = synth.fields(env_name)
    rs = np.random.RandomState(0)

    def mk(n):
        idx = rs.choice(REPLAY_ROWS, n, replace=False)
        return {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in fields}
    eps = [torch.from_numpy(e) for e in synth.normal_eps(agent.N_EPS, B, env.n_u, seed=1)]
    agent.update(mk(B), eps, 1)                      # warm-up (allocator, thread pool)
    log("cpu_baseline: warm-up update done")
    # a bounded sample of about 12 s of CPU work: updates for ~8 s (at least 4), NODE fits for ~4 s (at least 1)
    n_upd, t0 = 0, time.perf_counter()
    while n_upd < 4 or (time.perf_counter() - t0 < 8.0 and n_upd < 400):
        agent.update(mk(B), eps, 1 + n_upd)
        n_upd += 1
    t_upd = (time.perf_counter() - t0) / n_upd
    log("cpu_baseline: %.3f s per update (%d updates)" % (t_upd, n_upd))
    node_fields = ("obs", "action", "next_obs", "t") if env_name == "SimulatedCars" else ("obs", "action", "next_obs")
    n_fit, t0 = 0, time.perf_counter()
    while n_fit < 1 or (time.perf_counter() - t0 < 4.0 and n_fit < 40):
        nb = mk(NODE_FIT_ROWS)
        agent.train_step(*[nb[f] for f in node_fields])
        n_fit += 1
    t_fit = (time.perf_counter() - t0) / n_fit
    per_update = t_upd + t_fit / NODE_FIT_INTERVAL
    return dict(value=B / per_update, unit="samples/s", cores=cores, kind="port",
                sample="oracle (PyTorch-CPU restatement): %d updates at B=%d (%.3f s each) + %d NODE fits on %d rows "
                       "(%.3f s each, amortised /%d), solver %s%s" % (n_upd, B, t_upd, n_fit, NODE_FIT_ROWS, t_fit,
                                                                      NODE_FIT_INTERVAL, solver,
                                                                      " + odeint_adjoint" if adjoint else ""))


def node_odeint_submetric(agent, env, B, solver, iters=50):
    """Isolated NODE solve over [0, dt] + its backward (gradient to the controls), P = 2 problems of B rows — the two
    rollouts (policy / backup policy) one update performs.  Times whole solves with HIP events on the launch stream."""
    sol = agent.node_solver
    P = 2
    n = P * B
    g = torch.Generator(device="cpu").manual_seed(7)
    y0 = ((torch.rand(n, sol.n_s, generator=g) * 2 - 1)).to(agent.device)
    u = ((torch.rand(n, sol.n_u, generator=g) * 2 - 1)).to(agent.device)
    dout = torch.randn(n, sol.n_s, generator=g).to(agent.device)
    dt = float(env.dt)

    def once():
        sol.forward(y0, u, P, B, solver, dt, agent.atol, agent.rtol)
        sol.backward(dout, need_du=True)

    for _ in range(5):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"value": n / (ms * 1e-3), "unit": "rows/s", "ms_per_solve_fwd_bwd": ms, "rows": n, "problems": P,
            "solver": solver}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--solver", default="dopri5", choices=["euler", "rk4", "dopri5"])
    ap.add_argument("--env", default="Unicycle", choices=["Unicycle", "SimulatedCars", "UnicycleBarrier", "Pvtol", "PvtolBarrier", "QuadrotorLike"])
    ap.add_argument("--adjoint", action="store_true",
                    help="differentiate every NODE solve by the continuous adjoint (odeint_adjoint; BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-step-control", default="shard", choices=["shard", "global"],
                    help="N > 1, dopri5: every rank controls the steps of its own rows (no collective inside a solve; "
                         "default) or the error norms are all-reduced (the single-device decisions over the global batch)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay the update as hipGraphs (measured equal to eager launches once descriptors are cached)")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: ONE global batch of this many rows sharded over the ranks (default 0: weak "
                         "scaling, --batch rows per rank)")
    ap.add_argument("--lean", action="store_true",
                    help="warm-up + timed region only (no pipelined replay, event-timed pass, sub-metric, CPU baseline): "
                         "the process runs exactly warmup + steps updates — what the counter passes profile")
    a = ap.parse_args()
    if a.lean:
        a.profile_steps, a.no_cpu_baseline = 0, True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    # one rank per GPU; NLBAC_BENCH_BACKEND=gloo lets several ranks share one card for a rehearsal of the N>1 path
    backend = os.environ.get("NLBAC_BENCH_BACKEND", "nccl")
    n_dev = max(1, torch.cuda.device_count())
    assert backend != "nccl" or local_rank < n_dev, "rank %d has no GPU of its own (%d visible)" % (local_rank, n_dev)
    torch.cuda.set_device(local_rank % n_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if a.env.endswith("Barrier") or a.env == "QuadrotorLike":
        from nlbac_amd.neural_barrier_certificate.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
    else:
        from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
    from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
    env = make_env(a.env, 0)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def measure(B, global_B, steps, warmup, extras, strong=False, dp_step_control=None):
        """One agent at ``B`` rows per rank (losses normalised by ``global_B``): warm-up, the timed region bracketed by
        barrier + synchronize, max over ranks.  ``extras``: also the pipelined replay / event-timed pass / sub-metric."""
        args = Args(global_B)                                         # global batch in the loss normalisation
        args.gamma_b = GAMMA_B[a.env]
        agent = SAC_CBF_CLF(env.obs_dim, env.action_space, env, args)
        agent.solver = a.solver
        agent.adjoint = a.adjoint
        agent.use_graphs = (world == 1) and a.graphs
        if world > 1:
            agent.enable_data_parallel(dist, step_control=dp_step_control or a.dp_step_control)
        dev = agent.device
        # the replay lives in HBM in the agent's row layout; minibatches are drawn and gathered on the device
        replay = DeviceReplayMemory(REPLAY_ROWS, 1234 + rank, agent, device_rng=True)
        replay.push_rows(replay_rows(agent, synth.transitions(a.env, REPLAY_ROWS, seed=1 + rank, env=env)))
        ws = agent._workspace(B)
        fit_local = max(1, NODE_FIT_ROWS // world) if strong else NODE_FIT_ROWS     # (strong: the fit batch is sharded too)
        fit_rows = torch.empty(fit_local, agent.lay.LD, device=dev)

        def draw():
            replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)       # index draw + gather + policy noise: one launch

        def step(i, sync=True):
            # (the minibatch is drawn by the agent through ``prefetch``: this update's unless the previous call already
            #  queued it — with the policy forward — behind its last launch, before blocking on the returned floats:
            #  SAC_CBF_CLF.update_on_device.  Same draws in the same order as drawing here.)
            if i % NODE_FIT_INTERVAL == 0:
                if ws.__dict__.get("_prefetched") is None:
                    agent.update_prefetch(ws, i, draw)             # (first update: keep the order minibatch, fit rows)
                agent.fit_node_rows(replay.sample_rows(fit_local, out=fit_rows))
            return agent.update_on_device(ws, i, sync=sync, prefetch=draw)

        def drop_prefetch():
            """Forget a queued draw (and rewind the replay's draw counter: the next call draws the same rows again)."""
            if ws.__dict__.get("_prefetched") is not None:
                ws._prefetched = None
                replay._draws -= 1

        for i in range(warmup):
            step(i)
        fence()
        drop_prefetch()
        snap = io.BytesIO()            # agent + replay-draw state at the start of the timed region (replayed below)
        agent.save_checkpoint(snap)
        draws0 = replay._draws
        fence()
        stats0 = dict(agent.node_solver.stats)
        t0 = time.perf_counter()
        marks = []
        for i in range(steps):
            ret = step(warmup + i)
            marks.append(time.perf_counter())       # (the step has returned its six floats: the host clock, no extra sync)
        fence()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if os.environ.get("NLBAC_BENCH_STEP_TIMES") == "1" and rank == 0:
            log("per-step ms (host clock at each step's return): " +
                " ".join("%.3f" % (1e3 * (b - a)) for a, b in zip([t0] + marks[:-1], marks)))
        stats = {k: v - stats0.get(k, 0) for k, v in agent.node_solver.stats.items()}      # of the timed region alone
        stats["node_fits"] = sum(1 for i in range(steps) if (warmup + i) % NODE_FIT_INTERVAL == 0)
        res = dict(agent=agent, B=B, global_B=global_B, elapsed=elapsed, value=global_B * steps / elapsed,
                   ms=1e3 * elapsed / steps, ret=[float(x) for x in ret], stats=stats,
                   fit_rows_per_rank=fit_local, step=step, base=warmup + steps)
        if not extras:
            return res
        # ---- secondary figure: the SAME updates again (state and replay draws rewound to the start of the timed
        #      region) with the 6 returned floats read one call late (sync="lagged"), the way a driver that only logs
        #      them can run; the headline above keeps the reference's blocking return
        snap.seek(0)
        agent.load_checkpoint(snap)
        ws._prefetched = None
        replay._draws = draws0
        fence()
        stats1 = dict(agent.node_solver.stats)
        t0 = time.perf_counter()
        with FlopCounter() as fc:
            for i in range(steps):
                step(warmup + i, sync="lagged")
        fence()
        el2 = max_over_ranks(time.perf_counter() - t0)
        again = {k: v - stats1.get(k, 0) for k, v in agent.node_solver.stats.items()}
        same = all(again.get(k, 0) == v for k, v in stats.items() if k != "node_fits")
        res["replay_same_solver_steps"] = bool(same)      # (deterministic kernels: the replay IS the region; reported, not assumed)
        res["timed_region_flop_per_update"] = fc.flop / steps      # (the replay runs the timed region's updates again)
        res["timed_region_mfma_launches_per_update"] = fc.launches / steps
        res["pipelined"] = {"value": global_B * steps / el2, "unit": "samples/s", "ms_per_step": 1e3 * el2 / steps,
                            "steps": steps,
                            "note": "the timed region replayed from the same state with the returned losses read one "
                                    "update late (pinned memory): identical updates, launch stream never drains"}
        res["base"] = warmup + 2 * steps
        return res

    strong_first = a.global_batch > 0
    if strong_first:
        assert a.global_batch % world == 0, "--global-batch must divide over the ranks"
        B, GB = a.global_batch // world, a.global_batch
    else:
        B, GB = a.batch, a.batch * world
    log("setup; warm-up + timed region: %d steps, %d rows per rank, global batch %d" % (a.steps, B, GB))
    main_run = measure(B, GB, a.steps, a.warmup, extras=not a.lean, strong=strong_first)
    agent, value, elapsed, step, base = (main_run[k] for k in ("agent", "value", "elapsed", "step", "base"))
    log("timed region done: %.3f ms/step, %.0f samples/s" % (main_run["ms"], value))
    pipelined = main_run.get("pipelined")

    # ---- N > 1: the other split of the same metric, measured right after (its own agent)
    other = None
    if world > 1 and not a.lean:
        if strong_first:
            o = measure(a.batch, a.batch * world, a.steps, a.warmup, extras=False)
            kind = "weak"
        else:
            assert a.batch % world == 0
            o = measure(a.batch // world, a.batch, a.steps, a.warmup, extras=False, strong=True)
            kind = "strong"
        other = {"scaling": kind, "value": o["value"], "unit": "samples/s", "ms_per_step": o["ms"],
                 "batch_per_gpu": o["B"], "global_batch": o["global_B"], "node_fit_rows_per_gpu": o["fit_rows_per_rank"],
                 "rollout_solver_stats": o["stats"]}
        del o
    # ... and, for dopri5, the same line under the OTHER step control: "global" (error norms all-reduced: every rank takes
    # the accept / reject decisions and step sizes the single-device run over the global batch takes — the computation
    # the N = 1 line and the parity tests run) next to "shard" (every rank controls its own rows' steps: no collective
    # inside a solve; a different computation within solver tolerance, see README).  Whoever reads a scaling curve off
    # these lines has both.
    other_ctl = None
    if world > 1 and not a.lean and a.solver == "dopri5":
        alt = "global" if a.dp_step_control == "shard" else "shard"
        o = measure(B, GB, a.steps, a.warmup, extras=False, strong=strong_first, dp_step_control=alt)
        other_ctl = {"dp_step_control": alt, "value": o["value"], "unit": "samples/s", "ms_per_step": o["ms"],
                     "batch_per_gpu": o["B"], "global_batch": o["global_B"], "rollout_solver_stats": o["stats"],
                     "matches_single_device_decisions": alt == "global"}
        del o

    # ---- roofline of the dominant kernel: separate pass, HIP events around each MLP launch ----------
    roofline = None
    if rank == 0 and a.profile_steps > 0:
        graphs_on = agent.use_graphs
        agent.use_graphs = False       # the event-timed pass launches the same kernels one by one
        stats0 = dict(agent.node_solver.stats)
        torch.cuda.synchronize()
        t_pass = time.perf_counter()
        with KernelTimer() as kt:
            for i in range(a.profile_steps):
                step(base + i)
            torch.cuda.synchronize()       # (not fence(): the other ranks run these updates without the timers and
            t_pass = time.perf_counter() - t_pass      # meet this one at the barrier after the pass)
            ks = kt.summary()
        agent.use_graphs = graphs_on
        stats_pass = {k: v - stats0.get(k, 0) for k, v in agent.node_solver.stats.items()}
        dom = max(ks, key=lambda k: ks[k]["ms"])
        kname = {"nlbac_mlp_fwd": "mlp_rrq_fwd_kernel", "nlbac_mlp_bwd_data": "mlp_rrq_bwd_kernel",
                 "nlbac_mlp_bwd_weights": "mlp_bwd_wide64_kernel",
                 "nlbac_node_rk_fwd": "node_rr_fwd_kernel", "nlbac_node_rk_bwd": "node_rr_bwd_kernel",
                 "nlbac_node_adj_step": "node_adj_rr_kernel",
                 "nlbac_concat_rk_fwd": "concat_rr_fwd_kernel", "nlbac_concat_rk_bwd": "concat_rr_bwd_kernel",
                 "nlbac_concat_adj_step": "concat_adj_rr_kernel"}[dom]
        roofline = dict(bound="mfma", kernel=kname, achieved=ks[dom]["tflops"], peak=PEAK_F32_MFMA_TFLOPS,
                        unit="TFLOP/s", frac=ks[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS,
                        traffic=pmc_traffic(kname, a.env, a.solver, B),
                        avg_launch_us=ks[dom]["avg_us"], launches=ks[dom]["launches"],
                        flops_per_launch=ks[dom]["flops"] / ks[dom]["launches"],
                        all={k: dict(avg_us=round(v["avg_us"], 2), tflops=round(v["tflops"], 2),
                                     launches=v["launches"]) for k, v in ks.items()})
        # The whole update against the MFMA roof, numerator and denominator from ONE window: the executed FLOP of the
        # MFMA tile kernels in the event-timed pass (updates that follow the timed region: solver regime and NODE fits
        # of their own, see solver_stats) over that pass's own wall time (host clock around the pass, launches bracketed
        # by events and hipGraphs off: a little slower than the timed region) and over the kernels' summed durations.
        flop_upd = sum(v["flops"] for v in ks.values()) / a.profile_steps
        busy_ms = sum(v["ms"] for v in ks.values()) / a.profile_steps
        pass_ms = 1e3 * t_pass / a.profile_steps
        event_pass = dict(window="event-timed pass: the %d updates after the timed region, every MFMA launch bracketed by "
                                 "two event records (they cost the stream ~0.4 ms per update: a lower bound)" % a.profile_steps,
                          executed_mfma_kernel_flop_per_update=flop_upd, ms_per_update=pass_ms,
                          achieved=flop_upd / (pass_ms * 1e-3) / 1e12, unit="TFLOP/s",
                          frac=flop_upd / (pass_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                          mfma_kernel_ms_per_update=busy_ms,
                          frac_of_kernel_time=flop_upd / (busy_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                          solver_stats=stats_pass,
                          node_fits=sum(1 for i in range(a.profile_steps) if (base + i) % NODE_FIT_INTERVAL == 0))
        # The whole update against the MFMA roof over the TIMED REGION itself (numerator and denominator of one window,
        # unperturbed): the FLOP its MFMA launches executed — counted on the host, launch by launch, while the pipelined
        # replay runs the very same updates again from the same state (FlopCounter; every attempted dopri5 step and every
        # NODE fit of the region is in it) — over the region's wall time.
        fl_t = main_run.get("timed_region_flop_per_update")
        if fl_t:
            roofline["update"] = dict(window="the timed region (%d updates): executed MFMA-kernel FLOP counted on the host over the "
                                             "identical replay, divided by the region's wall time" % a.steps,
                                      executed_mfma_kernel_flop_per_update=fl_t, ms_per_update=main_run["ms"],
                                      achieved=fl_t / (main_run["ms"] * 1e-3) / 1e12, unit="TFLOP/s",
                                      frac=fl_t / (main_run["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                      mfma_launches_per_update=main_run["timed_region_mfma_launches_per_update"],
                                      solver_stats=main_run["stats"],
                                      replay_same_solver_steps=main_run.get("replay_same_solver_steps"),
                                      event_pass=event_pass)
        else:
            roofline["update"] = event_pass
    elif a.profile_steps:
        for i in range(a.profile_steps):
            step(base + i)
    fence()
    if rank == 0:
        alg = algorithmic_bytes_per_update(agent, B)
        cnt = pmc_update_traffic(a.env, a.solver, B, a.adjoint)
        sec = main_run["ms"] * 1e-3
        hbm = dict(algorithmic_bytes=alg["total"], algorithmic_split=alg, counter_bytes=cnt,
                   achieved_bw=alg["total"] / sec / 1e9, unit="GB/s", peak=PEAK_HBM_BYTES_PER_S / 1e9,
                   frac=alg["total"] / sec / PEAK_HBM_BYTES_PER_S,
                   counter_bw=(cnt / sec / 1e9) if cnt else None,
                   counter_frac=(cnt / sec / PEAK_HBM_BYTES_PER_S) if cnt else None,
                   note="per update and GPU: algorithmic = minibatch rows + optimiser / target / NODE parameter traffic "
                        "(SURVEY.md 8d) + the amortised NODE fit; counter = FETCH_SIZE x2 + WRITE_SIZE of all kernels "
                        "from the committed rocprofv3 --pmc passes (includes Infinity-Cache hits)")
        if roofline is None:
            roofline = dict(bound="mfma", kernel=None, achieved=None, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=None, traffic=None)
        roofline["hbm"] = hbm

    # ---- sub-metric: the NODE odeint alone, forward + backward to the controls, on the rollout
    #      shape of the update (one problem per controller, B rows each)
    ode_sub = None
    if rank == 0 and world == 1 and not a.lean:
        ode_sub = node_odeint_submetric(agent, env, B, a.solver)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(B, a.solver, a.env, adjoint=a.adjoint)

    if rank == 0:
        out = {
            "metric": "ODE-integrate+update samples/sec, Unicycle batch 4096" if a.env == "Unicycle" else
                      "ODE-integrate+update samples/sec, %s batch %d" % (a.env, GB if strong_first else B),
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": main_run["ms"], "higher_is_better": True, "scaling": "strong" if strong_first else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s B=%d %s%s (%s); NODE fit on %d rows every %d "
                                   "updates; replay of %d synthetic transitions resident in HBM"
                                   % (a.env, GB if strong_first else B, a.solver, " + odeint_adjoint" if a.adjoint else "",
                                      WORKLOAD_NOTE[a.env], NODE_FIT_ROWS, NODE_FIT_INTERVAL, REPLAY_ROWS),
                       "solver": a.solver, "adjoint": bool(a.adjoint), "batch_per_gpu": B, "global_batch": GB,
                       "node_fit_rows_per_gpu": main_run["fit_rows_per_rank"],
                       "parallelism": "dp%d" % world,
                       "dp_step_control": (a.dp_step_control if world > 1 and a.solver == "dopri5" else None), "hipgraph": bool(agent.use_graphs),
                       "rollout_solver_stats": main_run["stats"], "last_losses": main_run["ret"],
                       "hbm_peak_allocated_bytes": int(torch.cuda.max_memory_allocated())},
            "roofline": roofline, "cpu_baseline": cpu, "node_odeint_fwd_bwd": ode_sub,
            "pipelined": pipelined,
        }
        if other is not None:
            out[other["scaling"]] = other
        if other_ctl is not None:
            out["other_step_control"] = other_ctl
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
