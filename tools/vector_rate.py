"""GPU: throughput of the vectorised driver (train.train_vectorized): N device environments + one update per vector step."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nlbac_amd
from nlbac_amd.envs import device as denv
from nlbac_amd.train import train_vectorized
from test_agent_parity_gpu import make_agent

N, B, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, 300
agent, _ = make_agent(B, 256, 0, "dopri5", "Unicycle", 50.0)
env = denv.make("Unicycle", N, seed=0)
args = types.SimpleNamespace(replay_size=1 << 20, seed=0, start_steps=2 * N, batch_size=B, updates_per_step=1,
                             NODE_model_update_interval=10)
train_vectorized(agent, env, args, 20 * N, log=lambda *a: None)        # warm-up (allocations, first launches)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = train_vectorized(agent, denv.make("Unicycle", N, seed=1), args, iters * N, log=lambda *a: None)
dt = time.perf_counter() - t0
print("N=%d lanes, batch %d: %d env steps + %d updates in %.3f s = %.2f M env steps/s, %.2f ms per vector step"
      % (N, B, res["steps"], res["updates"], dt, res["steps"] / dt / 1e6, 1e3 * dt / iters))
