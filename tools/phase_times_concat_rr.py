"""Shader-clock stamps inside the register-resident single-net RK forward (needs the -DRR_TIMING build:
tools/build_variant.sh rrtiming -DRR_TIMING; NLBAC_HIP_LIB=<that .so> python tools/phase_times_concat_rr.py [rows per problem]).
Waves 0 and 1 of workgroup 0 (each wave is a 16-row tile of its own)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from test_agent_parity_gpu import make_agent

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
agent, env = make_agent(128, 256, 0, "rk4", "SimulatedCars", 0.5)
sol = agent.task.solver1
sol.keep_acts = False
n = 2 * B
y0 = torch.randn(n, sol.n_s, device="cuda")
u = torch.randn(n, sol.n_u, device="cuda")
for _ in range(3):
    sol.forward(y0, u, 2, B, "rk4", 0.02)
torch.cuda.synchronize()
ws = sol.ctx["steps"][0]["ws"]
stamps = torch.zeros(2048, dtype=torch.int64, device="cuda")
sol._rk_fused(ws, y0, u, 2, B, "rk4", 0, 4, h_host=[0.02, 0.02], err=stamps.view(torch.float32))
torch.cuda.synchronize()
t = stamps.cpu().numpy()
for half in (0, 1):
    s = t[half * 256:half * 256 + 64]
    print("wave %d: prime..constants+tile rows %d" % (half, s[1] - s[0]))
    for st in range(4):
        b = 2 + 8 * st
        print("  stage %d: input %5d  L0 %5d  L1 %5d  L2 %5d  out+stores %5d | stage %6d" %
              (st, s[b + 1] - s[b], s[b + 2] - s[b + 1], s[b + 3] - s[b + 2], s[b + 4] - s[b + 3], s[b + 5] - s[b + 4], s[b + 5] - s[b]))
    print("  after the last stage (barrier) %d; first stamp -> last %d" % (s[2 + 32] - s[2 + 24 + 5], s[2 + 32] - s[0]))

# ---- the backward of the same solve (stamps land in the `dyn` argument, unused without a normaliser)
from nlbac_amd import _lib
import nlbac_amd.odeint as od
sol.forward(y0, u, 2, B, "rk4", 0.02)
dout = torch.randn(n, sol.n_s, device="cuda")
sol.backward(dout, need_du=True)
torch.cuda.synchronize()
stamps.zero_()
orig = _lib.call


def patched(name, *a):
    if name == "nlbac_concat_rk_bwd":
        a = list(a)
        a[-4] = stamps.data_ptr()
    return orig(name, *a)


od._lib.call = patched
sol.backward(dout, need_du=True)
torch.cuda.synchronize()
od._lib.call = orig
t = stamps.cpu().numpy()
for half in (0, 1):
    s = t[half * 256:half * 256 + 64]
    first = 2 + 8 * 3
    print("backward, wave %d: prologue %d" % (half, s[first] - s[0]))
    for st in (3, 2, 1, 0):
        b = 2 + 8 * st
        print("  stage %d: dy + top %5d  prod 1 %5d  prod 2 %5d  dX %5d  stage algebra %5d | stage %6d" %
              (st, s[b + 1] - s[b], s[b + 2] - s[b + 1], s[b + 3] - s[b + 2], s[b + 4] - s[b + 3], s[b + 5] - s[b + 4], s[b + 5] - s[b]))
    print("  first stamp -> end of the stage loop %d" % (s[1] - s[0]))
