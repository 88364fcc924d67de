"""Shader-clock stamps of wave 0 / workgroup 0 of the register-resident MLP data backward (-DRR_TIMING build; the stamps
land behind the first net's two dz layers: this script allocates a third)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.arena import stream_ptr, mlp_array, io_array, skinny_partials_ws
from test_agent_parity_gpu import make_agent

B = 4096
agent, env = make_agent(B, 256, 0, "euler")
hs = agent.h_crit[:3] + agent.h_crit[:3]
for n_nets, sk, want_dx in ((2, False, True), (2, True, True), (3, True, False), (5, False, True)):
    descs = [h.desc for h in hs[:n_nets]]
    arr = mlp_array(descs)
    io = io_array(n_nets)
    x = torch.randn(B, 16, device="cuda")
    dy = torch.randn(B, 4, device="cuda")
    ys = [torch.zeros(B + 8, 4, device="cuda") for _ in range(n_nets)]
    acts = [torch.zeros(2, B, 256, device="cuda") for _ in range(n_nets)]
    dz = [torch.zeros(3, B, 256, device="cuda") for _ in range(n_nets)]
    dx = [torch.zeros(B, 16, device="cuda") for _ in range(n_nets)]
    for i in range(n_nets):
        io[i].x0, io[i].x0_dim, io[i].x0_ld = x.data_ptr(), descs[i].in_dim, 16
        io[i].y, io[i].y_ld = ys[i].data_ptr(), 4
        io[i].acts, io[i].dz = acts[i].data_ptr(), dz[i].data_ptr()
        io[i].dy, io[i].dy_ld = dy.data_ptr(), 4
        if want_dx:
            io[i].dx, io[i].dx_ld = dx[i].data_ptr(), 16
    ws = skinny_partials_ws(arr, (io,), n_nets, B, "cuda") if sk else None
    _lib.call("nlbac_mlp_fwd", arr, io, n_nets, B, stream_ptr())
    for _ in range(5):
        _lib.call("nlbac_mlp_bwd_data", arr, io, n_nets, B, stream_ptr())
    torch.cuda.synchronize()
    t = dz[0][2].view(-1)[:16].view(torch.int64).cpu().numpy()
    names = ["prologue+dy", "top layer", "panel", "dx half", "stores+lds+barrier", "dx out+partials"]
    print("%d nets sk=%d dx=%d:" % (n_nets, sk, want_dx), "  ".join("%s %d" % (nm, t[k + 1] - t[k]) for k, nm in enumerate(names)),
          " total", t[6] - t[0])
