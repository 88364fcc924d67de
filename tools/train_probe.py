"""GPU: where the time of a long dopri5 training run goes (solver step counts, fit vs update time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nlbac_amd  # noqa: F401
from nlbac_amd import train

agents = []
orig = train.train
train.train = lambda agent, *a, **k: (agents.append(agent), orig(agent, *a, **k))[1]
steps = sys.argv[1] if len(sys.argv) > 1 else "5000"
train.main(["--env", "Unicycle", "--gamma_b", "50", "--max_episodes", "40", "--cuda", "--updates_per_step", "2",
            "--batch_size", "256", "--seed", "1", "--start_steps", "1000", "--device_replay", "--solver", "dopri5",
            "--max_steps", steps])
a = agents[0]
for name, sv in (("rollout", a.node_solver), ("fit", a.fit_solver)):
    print(name, sv.stats, "pools", {k[:2]: p.n_slots for k, p in sv.__dict__.get("_pools", {}).items()},
          "chain length", sv.__dict__.get("_chain_len"), "last solve steps", len(sv.ctx.get("steps") or []),
          "last attempts", (sv.ctx.get("info") or [])[-2:])
ws = a._workspace(256)
mem = agents[0]
torch.cuda.synchronize()
print("allocated GiB", torch.cuda.memory_allocated() / 2**30, "reserved GiB", torch.cuda.memory_reserved() / 2**30)
