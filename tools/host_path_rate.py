"""GPU: throughput of the update when the caller hands over HOST buffers, as the reference's driver does
(SURVEY.md §8b): (1) update_from_host(batch) with a pre-sampled numpy minibatch — adds the pack + one H2D copy per
update (393 KB at B=4096); (2) update_parameters(memory, ...) with the reference-shaped host ReplayMemory — adds
random.sample + np.stack on the host as well.  bench.py's `value` is the device-resident path; these are the
PCIe-inclusive figures quoted in DESIGN.md §6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import nlbac_amd  # noqa: F401
from nlbac_amd import synth
from nlbac_amd.sac_cbf_clf.replay_memory import ReplayMemory
from test_agent_parity_gpu import make_agent

B, N = 4096, 65536
agent, env = make_agent(B, 256, 0, "dopri5")
tr = synth.transitions("Unicycle", N, seed=1, env=env)
fields = synth.FIELDS
mem = ReplayMemory(N + 1, 7)          # (a full ring wraps `position` to 0, and the NODE batch is min(position, 32768))
for i in range(N):
    mem.push(*[tr[f][i] for f in fields])


def timed(fn, n, warm=10):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


rs = np.random.RandomState(0)
batches = [tuple(tr[f][idx] for f in fields) for idx in (rs.choice(N, B, replace=False) for _ in range(8))]
ms1 = timed(lambda i: agent.update_from_host(batches[i % 8], i, None), 100)
ms2 = timed(lambda i: agent.update_parameters(mem, B, i, None, mem, 10), 60)
print("update_from_host (numpy minibatch -> pack -> H2D -> update): %.3f ms/update = %.2f M samples/s" % (ms1, B / ms1 / 1e3))
print("update_parameters(host ReplayMemory: random.sample + np.stack + H2D, NODE fit every 10th): %.3f ms/update = %.2f M samples/s"
      % (ms2, B / ms2 / 1e3))
