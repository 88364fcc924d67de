"""Per-step wall time of the headline workload split by what the dopri5 rollout had to do (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nlbac_amd import synth
from nlbac_amd.envspec import make_env
from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory

B = 4096
env = make_env("Unicycle", 0)
agent = SAC_CBF_CLF(7, env.action_space, env, bench.Args(B))
agent.solver = "dopri5"
replay = DeviceReplayMemory(65536, 1234, agent, device_rng=True)
replay.push_rows(bench.replay_rows(agent, synth.transitions("Unicycle", 65536, seed=1, env=env)))
ws = agent._workspace(B)
fit_rows = torch.empty(32768, agent.lay.LD, device=agent.device)
st = agent.node_solver.stats
kinds = {}
for i in range(260):
    replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)
    fit = i % 10 == 0
    before = dict(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fit:
        agent.fit_node_rows(replay.sample_rows(32768, out=fit_rows))
    agent.update_on_device(ws, i, eps_ready=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    kind = ("fit+" if fit else "") + ("split" if st["split"] > before["split"] else "multi" if st["multi_attempt"] > before["multi_attempt"] else "single")
    if i >= 20:
        kinds.setdefault(kind, []).append(dt)
    if i % 20 == 0:
        print(i, kind, "%.3f ms" % dt, dict(st), flush=True)
for k, v in sorted(kinds.items()):
    print("%-12s n=%3d  mean %.3f ms  median %.3f ms" % (k, len(v), np.mean(v), np.median(v)))
