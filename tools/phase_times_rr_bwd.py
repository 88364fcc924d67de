"""Shader-clock stamps inside the register-resident fused RK backward (node_rr_bwd_kernel; needs a -DRR_TIMING build:
NLBAC_HIP_LIB=<variant> python tools/phase_times_rr_bwd.py [rows = 8192]).  Workgroup 0, wave 0 of f_net and of g_net."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.odeint import AffineNodeSolver
from test_agent_parity_gpu import make_agent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
agent, env = make_agent(128, 256, 0, "dopri5")
sol = AffineNodeSolver(agent.neural_ode_model, "cuda")
sol.keep_acts = False
y0 = (torch.rand(n, 3) * 4 - 2).cuda()
u = (torch.rand(n, 2) * 2 - 1).cuda()
dout = torch.randn(n, 3).cuda()
for _ in range(3):
    sol.forward(y0, u, 2, n // 2, "dopri5", 0.02)
    sol.backward(dout, need_du=True)
torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
assert _lib.load().nlbac_debug_bwd_stamps(buf) == 0
for grp, name in ((0, "f_net wave"), (1, "g_net wave")):
    s = [buf[grp * 256 + k] for k in range(256)]
    t0 = s[0]
    print("%s: prologue %d" % (name, s[1] - t0))
    labels = ("dy/du", "top", "first product (split) + hand-over", "products", "dX", "barrier", "stage algebra")
    for st in range(6, 0, -1):
        b = 2 + 8 * st
        d = [s[b + k + 1] - s[b + k] for k in range(7)]
        print("  stage %d: " % st + "  ".join("%s %5d" % (l, v) for l, v in zip(labels, d)) + " | stage %6d" % (s[b + 7] - s[b]))
    print("  first stamp -> end of the stage loop %d" % (s[2 + 56] - t0))
