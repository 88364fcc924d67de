"""cProfile of the bench-shaped update loop (B=4096, device replay): host time per update_on_device call."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd  # noqa: F401
from nlbac_amd import synth
import bench
from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
from test_agent_parity_gpu import make_agent

B = 4096
agent, env = make_agent(B, 256, 0, "dopri5")
replay = DeviceReplayMemory(65536, 1234, agent, device_rng=True)
replay.push_rows(bench.replay_rows(agent, synth.transitions("Unicycle", 65536, seed=1, env=env)))
ws = agent._workspace(B)
for i in range(1, 30):
    replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)
    agent.update_on_device(ws, i, eps_ready=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(31, 131):
    if i % 10 == 0:
        continue
    replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)
    agent.update_on_device(ws, i, eps_ready=True)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
