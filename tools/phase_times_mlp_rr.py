"""Shader-clock stamps of wave 0 / workgroup 0 of the register-resident MLP forward (-DRR_TIMING build; the stamps
land behind the first net's output rows, so y must have 64 spare bytes: this script allocates its own)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.arena import stream_ptr, mlp_array, io_array
from test_agent_parity_gpu import make_agent

B = 4096
agent, env = make_agent(B, 256, 0, "euler")
hs = agent.h_crit[:3] + agent.h_crit[:3]
for n_nets in (1, 3, 6):
    descs = [h.desc for h in hs[:n_nets]]
    arr = mlp_array(descs)
    io = io_array(n_nets)
    x = torch.randn(B, 16, device="cuda")
    ys = [torch.zeros(B + 8, 4, device="cuda") for _ in range(n_nets)]
    acts = [torch.zeros(2, B, 256, device="cuda") for _ in range(n_nets)]
    for i in range(n_nets):
        io[i].x0, io[i].x0_dim, io[i].x0_ld = x.data_ptr(), descs[i].in_dim, 16
        io[i].y, io[i].y_ld = ys[i].data_ptr(), 4
        io[i].acts = acts[i].data_ptr()
    for _ in range(5):
        _lib.call("nlbac_mlp_fwd", arr, io, n_nets, B, stream_ptr())
    torch.cuda.synchronize()
    t = ys[0].view(-1)[B * 4:].view(torch.int64).cpu().numpy()
    names = ["prologue", "layer 0", "panel", "output half", "stores+barrier"]
    print("%d nets:" % n_nets, "  ".join("%s %d" % (nm, t[k + 1] - t[k]) for k, nm in enumerate(names)), " total", t[5] - t[0])
