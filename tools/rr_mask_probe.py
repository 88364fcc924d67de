"""Do the mask words of the register-resident forward (mask mode) agree with the activations it saves in activation mode?
word[layer][stage*n + row][q] bit (KS-1-k)  <->  unit(k, q) = 16 (k//4) + 4 q + k%4  (last block: 16 (NB-1) + R q + r)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nlbac_amd  # noqa
from test_agent_parity_gpu import make_agent
from nlbac_amd.odeint import AffineNodeSolver

solver = sys.argv[1] if len(sys.argv) > 1 else "rk4"
rows = 96
agent, env = make_agent(128, 256, 0, solver)
node = agent.neural_ode_model
g = torch.Generator().manual_seed(rows)
y0 = torch.cat([torch.rand(2 * rows, 2, generator=g) * 4 - 2, torch.rand(2 * rows, 1, generator=g) * 6 - 3], 1).cuda()
u = (torch.rand(2 * rows, 2, generator=g) * 2 - 1).cuda() * torch.tensor([3.5, 12.0]).cuda()
dout = torch.randn(2 * rows, 3, generator=g).cuda()
res = {}
for keep in (True, False):
    sol = AffineNodeSolver(node, "cuda")
    sol.keep_acts = keep
    out = sol.forward(y0, u, 2, rows, solver, 0.02).clone()
    du, dy0 = sol.backward(dout, need_du=True, need_dy0=True)
    ws = sol.ctx["steps"][-1]["ws"]
    res[keep] = (out, du.clone(), dy0.clone(), ws.acts_f.clone().cpu().numpy(), ws.acts_g.clone().cpu().numpy(), ws.S, ws.n)
hid = node.f.hid
NB, KS = (hid + 15) // 16, hid // 4
R = (hid - 16 * (NB - 1)) // 4
def unit(k, q):
    return 16 * (k // 4) + 4 * q + k % 4 if k < 4 * (NB - 1) else 16 * (NB - 1) + R * q + (k - 4 * (NB - 1))
for name, idx in (("f", 3), ("g", 4)):
    A = res[True][idx] > 0           # [layer][S*n][hid]
    Wd = res[False][idx].view(np.uint32)   # [layer][S*n][4]
    bad = 0
    for l in range(A.shape[0]):
        for k in range(KS):
            for q in range(4):
                bits = (Wd[l, :, q] >> np.uint32(KS - 1 - k)) & np.uint32(1)
                d = bits.astype(bool) != A[l, :, unit(k, q)]
                if d.any():
                    bad += int(d.sum())
                    if bad < 40:
                        print("net %s layer %d k %d q %d: %d rows differ (first %s)" % (name, l, k, q, d.sum(), np.nonzero(d)[0][:6]))
    print("net", name, "mismatching mask bits:", bad, "of", A.size)
print("out equal", torch.equal(res[True][0], res[False][0]), "du equal", torch.equal(res[True][1], res[False][1]),
      "dy0 max diff", float((res[True][2] - res[False][2]).abs().max()))
