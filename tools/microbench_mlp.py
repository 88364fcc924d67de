"""GPU micro-benchmark of nlbac_mlp_fwd / nlbac_mlp_bwd_data on the agent's own launch shapes (B=4096)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.arena import stream_ptr, bwd_weights
from test_agent_parity_gpu import make_agent

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
agent, env = make_agent(B, 256, 0, "euler")
ws = agent._workspace(B)
ws.np_now, ws.blam_upd = 2, 0
P = agent._plan(ws, 2)
s = stream_ptr()
cases = {
    "fwd 1 net  pi(s')": lambda: _lib.call("nlbac_mlp_fwd", P.n_pol, P.io_pol_next, 1, B, s),
    "fwd 3 nets pi(s'),pi(s)x2": lambda: _lib.call("nlbac_mlp_fwd", P.n_pol3, P.io_pol3, 3, B, s),
    "fwd 6 nets critics": lambda: _lib.call("nlbac_mlp_fwd", P.n_six, P.io_six, 6, B, s),
    "fwd 2 nets actors": lambda: _lib.call("nlbac_mlp_fwd", P.n_act, P.io_act, 2, B, s),
    "fwd 5 nets Q(s,pi)": lambda: _lib.call("nlbac_mlp_fwd", P.n_q5, P.io_q5, 5, B, s),
    "bwd_data 3 nets critics": lambda: _lib.call("nlbac_mlp_bwd_data", P.n_crit, P.io_crit, 3, B, s),
    "bwd_data 4 nets Q(s,pi)": lambda: _lib.call("nlbac_mlp_bwd_data", P.n_q5, P.io_q5, 4, B, s),
    "bwd_data 2 nets actors": lambda: _lib.call("nlbac_mlp_bwd_data", P.n_act, P.io_act, 2, B, s),
    "bwd_weights 3 nets critics": lambda: bwd_weights(P.n_crit, P.io_crit, 3, B, agent.ar_c.n_slabs, agent.ar_c.n, agent.device, ws=P.sk_crit),
    "bwd_weights 2 nets actors": lambda: [bwd_weights(nets, gio, cnt, B, g.arena.n_slabs, g.arena.n, agent.device, ws=sk) for g, cnt, nets, gio, sk in P.act_groups],
}


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for k, fn in cases.items():
    t = sorted(timeit(fn) for _ in range(5))
    print("%-26s median %.1f us  min %.1f us" % (k, t[2], t[0]))
