"""GPU micro-benchmark of the fused RK-step backward (nlbac_node_rk_bwd) on a dopri5 step; run it against
ablation builds (NLBAC_HIP_LIB=...) to attribute time inside the kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.odeint import AffineNodeSolver
from test_agent_parity_gpu import make_agent

agent, env = make_agent(128, 256, 0, "dopri5")
node = agent.neural_ode_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
y0 = (torch.rand(n, 3) * 4 - 2).cuda()
u = (torch.rand(n, 2) * 2 - 1).cuda()
dout = torch.randn(n, 3).cuda()
times = {}
orig = _lib.call


def timed_call(name, *args):
    if name != "nlbac_node_rk_bwd":
        return orig(name, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(name, *args)
    e1.record()
    times.setdefault("ev", []).append((e0, e1))
    return r


for keep in (False, True):
    sol = AffineNodeSolver(node, "cuda")
    sol.keep_acts = keep
    sol.forward(y0, u, 2, n // 2, "dopri5", 0.02)
    for _ in range(5):
        sol.backward(dout, need_du=True)
    torch.cuda.synchronize()
    import nlbac_amd.odeint as od
    od._lib.call = timed_call
    times.clear()
    for _ in range(40):
        sol.backward(dout, need_du=True)
    torch.cuda.synchronize()
    od._lib.call = orig
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in times["ev"])
    print("rows %d keep_acts=%s: node_rk_bwd median %.1f us  min %.1f us" % (n, keep, t[len(t) // 2], t[0]))
