"""GPU one-off: random (env, batch, hidden width, solver) configurations, two updates each (the first with a NODE fit),
six returned floats against the CPU oracle.  Not part of the test suite; run it after touching kernels' edge handling."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import nlbac_amd  # noqa: F401
from nlbac_amd import synth
from oracle import nlbac_oracle as O
from test_agent_parity_gpu import make_agent

GAMMA_B = {"Unicycle": 50.0, "Pvtol": 0.8, "SimulatedCars": 0.5, "UnicycleBarrier": 5.0, "PvtolBarrier": 1.0}
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for case in range(n_cases):
    env_name = rng.choice(list(GAMMA_B))
    B = rng.choice([2, 3, 5, 17, 31, 32, 33, 64, 100, 255, 257, 513, 1025])
    H = rng.choice([32, 64, 100, 128, 160, 256])
    solver = rng.choice(["euler", "rk4", "dopri5"])
    try:
        agent, env = make_agent(B, H, 0, solver, env_name, GAMMA_B[env_name])
        oargs = O.Args(batch_size=B, hidden_size=H, seed=0)
        oargs.gamma_b = agent.gamma_b
        oracle = O.make_oracle(synth.fixture_env(env_name, 0), oargs, synth.agent_weights(env_name, H, 0), solver=solver)
        tr = synth.transitions(env_name, 2048, seed=3 + case, env=env)
        fields = synth.fields(env_name)
        node_fields = ("obs", "action", "next_obs", "t") if env_name == "SimulatedCars" else ("obs", "action", "next_obs")
        worst = 0.0
        for u in (0, 1):
            idx = np.random.RandomState(u + case).choice(2048, B, replace=False)
            batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in fields}
            eps = [torch.from_numpy(e) for e in synth.normal_eps(agent.task.n_eps, B, env.n_u, seed=u)]
            node = tuple(batch[f] for f in node_fields) if u == 0 else None
            R = oracle.update(batch, eps, u, node_batch=node)
            agent.set_noise(eps)
            ret = agent.update_from_host(tuple(batch[f].numpy() for f in fields), u,
                                         tuple(x.numpy() for x in node) if node else None)
            worst = max(worst, max(abs(a - b) / (abs(b) + 1e-3) for a, b in zip(ret, R["ret"])))
        flag = "" if worst < 1e-4 else "   <-- MISMATCH"
        bad += worst >= 1e-4
        print("%-16s B=%-5d H=%-4d %-7s max rel err %.2e%s" % (env_name, B, H, solver, worst, flag), flush=True)
    except Exception as e:                                   # noqa: BLE001
        bad += 1
        print("%-16s B=%-5d H=%-4d %-7s ERROR %s: %s" % (env_name, B, H, solver, type(e).__name__, str(e)[:160]), flush=True)
print("cases with problems:", bad)
