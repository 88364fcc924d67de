import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nlbac_amd
from nlbac_amd import synth
from test_agent_parity_gpu import make_agent
agent, env = make_agent(8, 256, 0, "euler")
tr = synth.transitions("Unicycle", 64, seed=5, env=env)
for i in range(50): agent.select_action(tr["obs"][i % 64])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(2000): agent.select_action(tr["obs"][i % 64])
print("select_action: %.1f us per call" % ((time.perf_counter() - t0) / 2000 * 1e6))
