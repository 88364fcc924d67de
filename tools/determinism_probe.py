"""GPU: run-to-run determinism probe of the training pattern (push one transition, two updates, one select_action per
step) without an environment: prints a checksum of all parameters every 50 steps.  Run twice and diff."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import nlbac_amd  # noqa: F401
from nlbac_amd import synth
from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
from test_agent_parity_gpu import make_agent

B = 128
solver = sys.argv[1] if len(sys.argv) > 1 else "euler"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
agent, env = make_agent(B, 256, 0, solver)
tr = synth.transitions("Unicycle", steps + 8, seed=1, env=env)
mem, nmem = DeviceReplayMemory(1 << 16, 0, agent), DeviceReplayMemory(1 << 16, 0, agent)
updates = 0


def checksum():
    torch.cuda.synchronize()
    return " ".join("%.17g" % float(a.theta.double().sum()) for a in agent.arenas) + " sc %.17g" % float(agent.sc.double().sum())


for i in range(steps):
    if len(mem) > B:
        for _ in range(2):
            agent.update_parameters(mem, B, updates, None, nmem, 10)
            updates += 1
    a = agent.select_action(tr["obs"][i])
    row = [tr[f][i] for f in synth.FIELDS]
    mem.push(*row)
    nmem.push(*row)
    if i % 50 == 49:
        print(i, updates, "%.9g" % float(np.sum(a)), checksum())
