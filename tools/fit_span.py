"""GPU timing of one NODE fit (SAC_CBF_CLF.fit_node_rows) on the bench's Unicycle workload: the whole fit under events
(graphs on, as the bench runs it) and, with graphs off, the time between consecutive C-ABI calls (each entry point's
kernels + what follows them up to the next call).

    python tools/fit_span.py [rows=32768] [slabs=48]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib, synth
from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
env = bench.make_env("Unicycle", 0)
args = bench.Args(4096)
args.gamma_b = bench.GAMMA_B["Unicycle"]
args.fit_grad_slabs = slabs
agent = SAC_CBF_CLF(env.obs_dim, env.action_space, env, args)
agent.solver = "dopri5"
replay = DeviceReplayMemory(65536, 1234, agent, device_rng=True)
replay.push_rows(bench.replay_rows(agent, synth.transitions("Unicycle", 65536, seed=1, env=env)))
rows = torch.empty(N, agent.lay.LD, device=agent.device)


def span(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(k):
        replay.sample_rows(N, out=rows)
        torch.cuda.synchronize()
        e0.record()
        agent.fit_node_rows(rows)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], sum(1 for t in ts if t < 1.15 * ts[0])


span(5)
print("fit on %d rows, %d slabs, graphs on : median %.0f us  min %.0f us (%d of 30 within 15%% of it)" % ((N, slabs) + span(30)))
agent.use_graphs = False
span(3)
print("                         graphs off: median %.0f us  min %.0f us (%d of 30 within 15%% of it)" % span(30))

orig = _lib.call
evs = []


def timed(name, *a):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append((name, e))
    return orig(name, *a)


import nlbac_amd.odeint as od, nlbac_amd.arena as ar, nlbac_amd.sac_cbf_clf.sac_cbf_clf as sc, nlbac_amd.sac_cbf_clf.tasks as tk
seqs = {}
for it in range(30):
    replay.sample_rows(N, out=rows)
    torch.cuda.synchronize()
    evs.clear()
    _lib.call = timed
    agent.fit_node_rows(rows)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(("end", e))
    _lib.call = orig
    torch.cuda.synchronize()
    names = tuple(n for n, _ in evs)
    seqs.setdefault(names, []).append([evs[i][1].elapsed_time(evs[i + 1][1]) * 1e3 for i in range(len(evs) - 1)])
# fits differ in how many dopri5 attempts their batch took: the table is of the shortest call sequence seen
names = min(seqs, key=len)
runs = seqs[names]
print("shortest call sequence (%d calls, %d of 30 fits):" % (len(names) - 1, len(runs)))
tot = 0.0
for i in range(len(names) - 1):
    v = sorted(r[i] for r in runs)
    tot += v[len(v) // 2]
    print("  %2d %-28s %8.1f us" % (i, names[i], v[len(v) // 2]))
print("  sum %.0f us" % tot)
