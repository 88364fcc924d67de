"""GPU micro-benchmark of the fused RK-step kernel (A/B variants in one process, interleaved rounds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nlbac_amd
from nlbac_amd.odeint import AffineNodeSolver
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_agent_parity_gpu import make_agent

agent, env = make_agent(128, 256, 0, "dopri5")
node = agent.neural_ode_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
y0 = (torch.rand(n, 3) * 4 - 2).cuda()
u = (torch.rand(n, 2) * 2 - 1).cuda()
sol = AffineNodeSolver(node, "cuda")
sol.ctx = {}
ws = sol._step_ws(n, 7, 0)
ctl = sol._ctl(2)
ctl[:, 0] = 0.02


def run(save_acts, st0=1, st1=7):
    sol._rk_fused(ws, y0, u, 2, n // 2, "dopri5", st0, st1, h_dev=ctl.data_ptr(), save_acts=save_acts)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


sol_m = AffineNodeSolver(node, "cuda")          # mask mode: what the update's rollouts run (ReLU mask words, f / g split)
sol_m.keep_acts = False
sol_m.ctx = {}
ws_m = sol_m._step_ws(n, 7, 0)
ctl_m = sol_m._ctl(2)
ctl_m[:, 0] = 0.02


def run_m(st0=1, st1=7):
    sol_m._rk_fused(ws_m, y0, u, 2, n // 2, "dopri5", st0, st1, h_dev=ctl_m.data_ptr(), save_acts=True)


variants = {
    "6 stages, mask words": lambda: run_m(),
    "6 stages, acts saved": lambda: run(True),
    "6 stages, no acts": lambda: run(False),
    "1 stage, mask words": lambda: run_m(0, 1),
    "1 stage, acts saved": lambda: run(True, 0, 1),
    "1 stage, no acts": lambda: run(False, 0, 1),
}
res = {k: [] for k in variants}
for rnd in range(5):
    for k, fn in variants.items():
        res[k].append(timeit(fn))
for k, v in res.items():
    v = sorted(v)
    print("%-28s median %.1f us  min %.1f us" % (k, v[len(v) // 2], v[0]))

# ---- the same launches with the caches as the update leaves them: a 96 MB streaming kernel in front of every launch (what
#      the head kernels' activation / dz traffic does to the XCDs' L2s), timed alone with events
scratch = torch.empty(24 * 1024 * 1024, device="cuda")


def timeit_cold(fn, iters=30):
    ts = []
    for _ in range(iters):
        scratch.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for k, fn in variants.items():
    med, mn = timeit_cold(fn)
    print("%-28s behind a 96 MB stream: median %.1f us  min %.1f us (event pair around ONE launch: ~+3 us)" % (k, med, mn))
