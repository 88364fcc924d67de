"""Reference-on-CPU timing — runs ONLY in the build container (needs /root/reference).

Imports the reference's own Unicycle agent exactly as oracle/gen_golden.py does (torchdiffeq stand-in = one explicit
Euler step, the only solver configuration the reference executes; model.device rebound to CPU; plain-object env) and
times ``update_parameters`` on a synthetic replay: 20 warm-up + N timed updates, ``torch.set_num_threads(8)``,
including the host ``ReplayMemory.sample`` and the every-10th NODE ``train_step`` on min(len, 32768) rows — the
metric's definition.  A second figure replaces the host sampling by a pre-sampled minibatch.

    python tools/time_reference_cpu.py [--batch 4096] [--updates 200] [--threads 8]

The reference cannot travel to the GPU box, so this number lives in BASELINE.md / README.md next to the oracle's
(``bench.py``'s ``cpu_baseline``, timed on the GPU box's host cores)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import gen_golden as G  # noqa: E402
from oracle import nlbac_oracle as O  # noqa: E402
from nlbac_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--replay", type=int, default=65536)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    M, S = G.import_reference("Unicycle")
    sys.path.insert(0, G.REFS["Unicycle"])
    from sac_cbf_clf.replay_memory import ReplayMemory
    from sac_cbf_clf.dynamics import DynamicsModel
    env = synth.fixture_env("Unicycle", 0)
    args = O.Args(batch_size=a.batch, hidden_size=256, seed=0)
    args.gamma_b = 50.0
    agent = S.SAC_CBF_CLF(7, env.action_space, env, args)       # solver stays the reference's hard-coded 'euler'
    dyn = DynamicsModel(env, args)
    tr = synth.transitions("Unicycle", a.replay, seed=1, env=env)
    mem, node = ReplayMemory(10000000, 0), ReplayMemory(10000000, 0)
    for i in range(a.replay):
        row = tuple(tr[f][i] for f in synth.FIELDS[:8])
        mem.push(*row, t=tr["t"][i], next_t=tr["next_t"][i])
        node.push(*row, t=tr["t"][i], next_t=tr["next_t"][i])

    def run(n, first, memory, node_memory):
        t0 = time.perf_counter()
        for u in range(first, first + n):
            agent.update_parameters(memory, a.batch, u, dyn, node_memory, 10)
        return time.perf_counter() - t0
    run(a.warmup, 0, mem, node)
    t_full = run(a.updates, a.warmup, mem, node)

    class Fixed:            # one pre-sampled minibatch / NODE batch: the update without the host-side sampling
        def __init__(self, src, n):
            self.rows, self.position = src.sample(batch_size=n), src.position

        def sample(self, batch_size):
            return self.rows
    fm, fn = Fixed(mem, a.batch), Fixed(node, min(node.position, 32768))
    t_fixed = run(a.updates, a.warmup + a.updates, fm, fn)
    out = dict(what="reference U/ agent on CPU (build container)", solver="euler", batch=a.batch, updates=a.updates,
               threads=a.threads, cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
               ms_per_update=1e3 * t_full / a.updates, samples_per_s=a.batch * a.updates / t_full,
               ms_per_update_presampled=1e3 * t_fixed / a.updates,
               samples_per_s_presampled=a.batch * a.updates / t_fixed, torch=torch.__version__)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
