"""GPU: the one-stage RK launches of dopri5's initial-step selection (f0, probe) with the scaled norm + step controller in
their epilogue (nlbac_rk_chain.norm_mode 0 / 1), without them, and followed by the separate nlbac_dopri_norm_control
launch — 8192 rollout rows."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.odeint import AffineNodeSolver
from test_agent_parity_gpu import make_agent
agent, env = make_agent(128, 256, 0, "dopri5")
n = 8192
y0 = (torch.rand(n, 3) * 4 - 2).cuda(); u = (torch.rand(n, 2) * 2 - 1).cuda()
sol = AffineNodeSolver(agent.neural_ode_model, "cuda")
sol.keep_acts = False
sol.forward(y0, u, 2, n // 2, "dopri5", 0.02)      # sets ctx, pools
ctx = sol.ctx; st = ctx["chain"]; pool, ws0 = st["pool"], st["ws0"]
cp = sol._ctl(2).data_ptr()
def timeit(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for mode in (0, 1):
    ch_f = sol._chain(ws0, pool, 2, n // 2, mode, read_ctl=False)
    ch_n = sol._chain(ws0, pool, 2, n // 2, mode, read_ctl=False); ch_n.norm_mode = -1
    name, s0, s1 = (("dopri5", 0, 1) if mode == 0 else ("probe", 1, 2))
    hd = cp if mode == 0 else cp + 8 * 6
    a = timeit(lambda: sol._rk_fused(ws0, y0, u, 2, n // 2, name, s0, s1, h_dev=hd, save_acts=(mode == 0), chain=ch_f))
    b = timeit(lambda: sol._rk_fused(ws0, y0, u, 2, n // 2, name, s0, s1, h_dev=hd, save_acts=(mode == 0), chain=ch_n))
    c = timeit(lambda: (sol._rk_fused(ws0, y0, u, 2, n // 2, name, s0, s1, h_dev=hd, save_acts=(mode == 0), chain=ch_n), sol._chain_control(ws0, pool, ch_n, y0, u, mode, 2, n // 2)))
    print("mode %d: fused epilogue %.1f us | no norm %.1f us | RK + separate norm_control %.1f us" % (mode, a, b, c))
