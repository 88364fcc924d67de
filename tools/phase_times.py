"""Phase timestamps of workgroup 0 of the fused forward (needs the -DEXP_TIMING build via NLBAC_HIP_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import nlbac_amd
from nlbac_amd.odeint import AffineNodeSolver
from test_agent_parity_gpu import make_agent

agent, env = make_agent(128, 256, 0, "dopri5")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
y0 = (torch.rand(n, 3) * 4 - 2).cuda()
u = (torch.rand(n, 2) * 2 - 1).cuda()
sol = AffineNodeSolver(agent.neural_ode_model, "cuda")
sol.keep_acts = False
ws = sol._step_ws(n, 7, 0)
ctl = sol._ctl(2)
ctl[:, 0] = 0.02
FWD = not os.environ.get("PHASE_BWD_ONLY")       # (-DEXP_TIMING_BWD build: the forward runs normally, only the backward stamps)
stamps = torch.zeros(4096, dtype=torch.int64, device="cuda")
err = stamps.view(torch.float32)
from nlbac_amd.odeint import fptr
for it in range(3 if FWD else 0):
    sol._rk_fused(ws, y0, u, 2, n // 2, "dopri5", 1, 7, h_dev=ctl.data_ptr(), c_err=fptr(0.0), err=err)
    torch.cuda.synchronize()
if FWD:
    # call with err pointing at the stamp buffer
    sol._rk_fused(ws, y0, u, 2, n // 2, "dopri5", 1, 7, h_dev=ctl.data_ptr(), c_err=fptr(0.0), err=err)
    torch.cuda.synchronize()
    t = stamps.cpu().numpy()
    base = t[0]
    names = ["stage input", "wide layers", "skinny out", "k = f + g u"]
    print("clock ticks (shader clock), workgroup 0; prologue ends at 0")
    for st in range(6):
        row = t[1 + 8 * st: 1 + 8 * st + 5] - base
        d = np.diff(row)
        print("stage %d: start %7d  " % (st + 1, row[0]) + "  ".join("%s %6d" % (nm, x) for nm, x in zip(names, d)))

    d = t[64:64 + 16].reshape(4, 4)
    print("wave 0, stage 2, per wide layer (ticks): GEMM issue+A reads | epilogue | barrier wait")
    for l in range(4):
        a = d[l]
        print("  layer %d: %6d | %6d | %6d   (layer total %6d)" % (l, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[3] - a[0]))

if os.environ.get("PHASE_FWD_ONLY"):
    sys.exit(0)
# ---- backward phases (mask mode): stamps come back through the dz_g pointer in the timing build
import nlbac_amd.odeint as od
dout = torch.randn(n, 3).cuda()
sol2 = AffineNodeSolver(agent.neural_ode_model, "cuda")
sol2.keep_acts = False
sol2.forward(y0, u, 2, n // 2, "dopri5", 0.02)
bst = torch.zeros(256, dtype=torch.int64, device="cuda")
real = od._lib.call
def patched(name, *a):
    if name == "nlbac_node_rk_bwd":
        a = list(a); a[20] = bst.data_ptr(); a = tuple(a)
    return real(name, *a)
od._lib.call = patched
for _ in range(3):
    sol2.backward(dout, need_du=True)
    torch.cuda.synchronize()
od._lib.call = real
b = bst.cpu().numpy()
names = ["dy fill + du", "top layer", "wide layers", "dX", "algebra"]
print("backward, workgroup 0 (ticks)")
for st in range(6, 0, -1):
    row = b[8 * st: 8 * st + 6]
    d = np.diff(row)
    print("stage %d: " % st + "  ".join("%s %6d" % (nm, x) for nm, x in zip(names, d)) + "   total %6d" % (row[5] - row[0]))
