"""Shader-clock stamps inside the register-resident fused RK forward (needs the -DRR_TIMING build:
tools/build_variant.sh rrtiming "-DRR_TIMING"; NLBAC_HIP_LIB=<that .so> python tools/phase_times_rr.py [rows] [bits]).
Waves 0 (f_net) and 2 (g_net) of workgroup 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import nlbac_amd
from nlbac_amd.odeint import AffineNodeSolver, fptr
from test_agent_parity_gpu import make_agent

agent, env = make_agent(128, 256, 0, "dopri5")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
bits = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
y0 = (torch.rand(n, 3) * 4 - 2).cuda()
u = (torch.rand(n, 2) * 2 - 1).cuda()
sol = AffineNodeSolver(agent.neural_ode_model, "cuda")
sol.keep_acts = not bits
sol.ctx = {}
ws = sol._step_ws(n, 7, 0)
ctl = sol._ctl(2)
ctl[:, 0] = 0.02
stamps = torch.zeros(4096, dtype=torch.int64, device="cuda")
err = stamps.view(torch.float32)
for st0, st1 in ((1, 7), (0, 1)):
    for it in range(4):
        stamps.zero_()
        sol._rk_fused(ws, y0, u, 2, n // 2, "dopri5", st0, st1, h_dev=ctl.data_ptr(), c_err=fptr(0.0), err=err)
        torch.cuda.synchronize()
    t = stamps.cpu().numpy()
    print("stages [%d, %d), %d rows, %s: ticks relative to the kernel's first stamp" % (st0, st1, n, "mask bits" if bits else "activations"))
    for grp, name in ((0, "f_net wave"), (1, "g_net wave")):
        a = t[grp * 256: grp * 256 + 256]
        base = a[0]
        print(" %s: tile constants %d" % (name, a[1] - a[0]))
        # (with the f / g split, NLBAC_NODE_SPLIT != 0: "L3" is the wave's part of f_net's last layer for the f_net wave, and
        #  g_net's output layer + the wait for f_net's layer-2 activations for the g_net wave; "out" the part itself + output)
        split = os.environ.get("NLBAC_NODE_SPLIT", "1") != "0"
        nl = 4 if (grp == 0 or split) else 3
        for k in range(st1 - st0):
            sb = 2 + 8 * k
            parts = ["start %6d" % (a[sb] - base), "L0 %5d" % (a[sb + 1] - a[sb])]
            for l in range(1, nl):
                parts.append("L%d %5d" % (l, a[sb + 1 + l] - a[sb + l]))
            parts.append("out %4d | wait %5d | combine %4d | stage %6d" % (a[sb + 5] - a[sb + nl], a[sb + 6] - a[sb + 5], a[sb + 7] - a[sb + 6], a[sb + 7] - a[sb]))
            print("  stage %d: " % (st0 + k) + "  ".join(parts))
