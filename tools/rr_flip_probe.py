"""Where do the device's ReLU masks differ from the oracle's fp32 run (tests/test_solver_gpu.py's multi-step case)?
Run with NLBAC_NODE_RR=1 / 0 to compare the register-resident and the LDS-tiled kernels."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nlbac_amd  # noqa
from nlbac_amd import synth
from test_agent_parity_gpu import make_agent
from test_solver_gpu import oracle_solve
from nlbac_amd.odeint import AffineNodeSolver

T = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
agent, env = make_agent(64, 64, 0, "dopri5")
W = synth.agent_weights("Unicycle", 64, 0)["node"]
gen = torch.Generator().manual_seed(int(T * 10))
n = 200
y0 = torch.cat([torch.rand(n, 2, generator=gen) * 4 - 2, torch.rand(n, 1, generator=gen) * 6 - 3], 1)
u = (torch.rand(n, 2, generator=gen) * 2 - 1) * torch.tensor([3.5, 12.0])
dout = torch.randn(n, 3, generator=gen)
out_o, dy0_o, du_o, gp_o, info = oracle_solve(W, y0, u, T, dout)
sol = AffineNodeSolver(agent.neural_ode_model, "cuda")
out = sol.forward(y0.cuda(), u.cuda(), 1, n, "dopri5", T)
print("RR" if os.environ.get("NLBAC_NODE_RR", "1") != "0" else "tiled", "steps", [(round(s[0], 6), s[2]) for s in info["steps"]])
print("x(T) max err", float((out.cpu() - out_o).abs().max()), "scale", float(out_o.abs().max()))
acc = [i for i, st in enumerate(info["steps"]) if st[2]]
tot = np.zeros(n, dtype=bool)
for k, a in enumerate(acc):
    ws = sol.ctx["steps"][k]["ws"]
    for st in ([0] if k == 0 else []) + list(range(1, 7)):
        call = 0 if st == 0 else 2 + 6 * a + (st - 1)
        for net, acts in ((0, ws.acts_f), (1, ws.acts_g)):
            for l, m in enumerate(info["masks"][call][net]):
                dev = (acts[l, st * n:(st + 1) * n] > 0).cpu().numpy()
                d = dev != m
                if d.any():
                    rows = np.nonzero(d.any(1))[0]
                    print("step %d stage %d net %d layer %d: %d units in %d rows differ; rows %s" % (k, st, net, l, d.sum(), len(rows), rows[:12]))
                tot |= d.any(1)
print("flipped rows:", tot.sum(), "of", n, " margins of flipped rows (max):", np.asarray(info["margin"])[tot].max() if tot.any() else None)
