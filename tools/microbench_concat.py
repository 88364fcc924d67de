"""GPU micro-benchmark of the single-net NODE's fused RK step (nlbac_concat_rk_fwd / _bwd) on SimulatedCars' net
(12 -> 64 x3 -> 10): whole solves timed per C-ABI call, euler (1 stage), rk4 (4) and dopri5 (1 + 1 + 6).
    python tools/microbench_concat.py [rows per problem = 8192]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
import nlbac_amd.odeint as od
from test_agent_parity_gpu import make_agent

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
agent, env = make_agent(128, 256, 0, "rk4", "SimulatedCars", 0.5)
sol = agent.task.solver1
n = 2 * B
y0 = torch.randn(n, sol.n_s, device="cuda")
u = torch.randn(n, sol.n_u, device="cuda")
dout = torch.randn(n, sol.n_s, device="cuda")
orig = _lib.call
evs = []


def timed(name, *a):
    if not name.startswith("nlbac_concat_rk"):
        return orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(name, *a)
    e1.record()
    evs.append((name, e0, e1))
    return r


for keep in (False, True):
    sol.keep_acts = keep
    for method in ("euler", "rk4", "dopri5"):
        for _ in range(4):
            sol.forward(y0, u, 2, B, method, 0.02)
            sol.backward(dout, need_du=True)
        torch.cuda.synchronize()
        evs.clear()
        od._lib.call = timed
        for _ in range(20):
            sol.forward(y0, u, 2, B, method, 0.02)
            sol.backward(dout, need_du=True)
        torch.cuda.synchronize()
        od._lib.call = orig
        acc = {}
        for name, e0, e1 in evs:
            acc.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
        print("rows 2 x %d  %-6s keep_acts=%-5s  " % (B, method, keep) +
              "   ".join("%s: %d launches / solve, median %.1f us, sum %.1f us / solve"
                         % (k[6:], len(v) // 20, sorted(v)[len(v) // 2], sum(v) / 20) for k, v in sorted(acc.items())))
