"""Worker for the data-parallel tests (one process per rank; launched by test_data_parallel.py).

  --device cpu : gloo; exercises nlbac_amd.parallel.DataParallel with the oracle's nets as the compute
                 engine: sharded gradients with global normalisation + all-reduce == full-batch gradients.
  --device cuda: gloo (both ranks share the one GPU of the test box); the product agent in data-parallel
                 mode on row shards of a golden-fixture minibatch; results are written for the parent to
                 compare against the single-device fixture.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import nlbac_amd  # noqa: E402,F401
from nlbac_amd import synth  # noqa: E402
from nlbac_amd.envspec import make_env  # noqa: E402
from nlbac_amd.parallel import DataParallel  # noqa: E402


def run_cpu(dp, out):
    from oracle import nlbac_oracle as O
    torch.set_num_threads(1)
    B, hidden, seed = 64, 64, 3
    W = synth.unicycle_agent_weights(hidden, seed)
    tr = synth.unicycle_transitions(256, seed=5)
    obs = torch.tensor(tr["obs"][:B], dtype=torch.float32)
    act = torch.tensor(tr["action"][:B], dtype=torch.float32)
    y = torch.tensor(tr["reward"][:B], dtype=torch.float32).unsqueeze(1)
    crit = {k: torch.tensor(v, requires_grad=True) for k, v in W["critic"].items()}
    lo, hi = dp.shard(B)
    q1, q2 = O.qnet(crit, obs[lo:hi], act[lo:hi])
    loss = (((q1 - y[lo:hi]) ** 2).sum() + ((q2 - y[lo:hi]) ** 2).sum()) / B      # global normalisation
    g = torch.autograd.grad(loss, list(crit.values()))
    flat = torch.cat([t.reshape(-1) for t in g] + [loss.detach().reshape(1)])
    dp.all_reduce_(flat)
    # parameter broadcast: rank 1 starts from garbage and must end up with rank 0's values
    theta = torch.cat([v.detach().reshape(-1) for v in crit.values()]).clone()
    if dp.rank != 0:
        theta.fill_(123.0)
    dp.broadcast_(theta)
    np.savez(out, flat=flat.numpy(), theta=theta.numpy(), lo=lo, hi=hi)


def run_cuda(dp, out, solver, env_name="Unicycle"):
    from common import case_inputs, load_golden
    from test_agent_parity_gpu import make_agent
    from nlbac_amd.sac_cbf_clf import _layout as SC
    B = 128
    g = load_golden(solver, B, env_name)
    seed, hidden = int(g["meta_seed"]), int(g["meta_hidden"])
    gamma_b = float(g["meta_gamma_b"]) if "meta_gamma_b" in g.files else 50.0
    torch.cuda.set_device(0)
    agent, env = make_agent(B, hidden, seed, solver, env_name, gamma_b)          # batch_size = global batch
    agent.enable_data_parallel(dist)
    tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    res = {}
    for ci in range(len(g["meta_calls"])):
        batch, eps, node, updates = case_inputs(g, ci, tr)
        lo, hi = dp.shard(B)
        nlo, nhi = dp.shard(node[0].shape[0])
        agent.set_noise([e[lo:hi] for e in eps])
        host = tuple(batch[f][lo:hi].numpy() for f in synth.fields(env_name))
        node_np = tuple(t[nlo:nhi].numpy() for t in node) if updates % 10 == 0 else None
        ret = agent.update_from_host(host, updates, node_np)
        torch.cuda.synchronize()
        sc = agent.sc.cpu().numpy()
        p = "c%d_" % ci
        res[p + "ret"] = np.array(ret)
        res[p + "required"] = sc[SC.SC_REQ:SC.SC_REQ + agent.num_constraints]
        res[p + "lambdas"] = np.array(agent.lambda_values)
        for name, mod in (("critic", agent.critic), ("policy", agent.policy), ("node", agent.neural_ode_model)):
            res[p + "p_" + name] = torch.cat([q.detach().reshape(-1) for q in mod.parameters()]).cpu().numpy()
    np.savez(out, **res)


def oracle_case_inputs(env_name, agent, env, B, ci, updates, tr):
    """The minibatch / noise / NODE rows of update ``ci`` of the oracle-checked cases (the pattern of
    test_adjoint_gpu.test_update_with_adjoint_matches_the_oracle), identical in the workers and in the parent."""
    rs = np.random.RandomState(ci)
    idx, nidx = rs.choice(4096, B, replace=False), rs.choice(4096, 512, replace=False)
    fields = synth.fields(env_name)
    batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in fields}
    eps = [torch.from_numpy(e) for e in synth.normal_eps(agent.task.n_eps if agent is not None else 3, B, env.n_u, seed=ci)]
    node = tuple(torch.tensor(tr[f][nidx], dtype=torch.float32) for f in ("obs", "action", "next_obs"))
    return batch, eps, node


ORACLE_CASE = dict(B=128, hidden=64, seed=0, updates=(0, 1, 20),
                   gamma_b={"Pvtol": 0.8, "Unicycle": 50.0, "PvtolBarrier": 0.8, "QuadrotorLike": 1.0})


def run_cuda_oracle_case(dp, out, solver, env_name, adjoint, step_control="global"):
    """Row shards of the oracle-checked updates (no reference fixture exists for them: the parent holds the results
    against the single-device oracle): returned floats and post-update parameters."""
    from test_agent_parity_gpu import make_agent
    c = ORACLE_CASE
    B = c["B"]
    torch.cuda.set_device(0)
    agent, env = make_agent(B, c["hidden"], c["seed"], solver, env_name, c["gamma_b"][env_name])
    agent.adjoint = bool(adjoint)
    agent.enable_data_parallel(dist, step_control=step_control)
    tr = synth.transitions(env_name, 4096, seed=c["seed"] + 1, env=env)
    fields = synth.fields(env_name)
    res = {"n_eps": np.array(agent.task.n_eps)}
    for ci, updates in enumerate(c["updates"]):
        batch, eps, node = oracle_case_inputs(env_name, agent, env, B, ci, updates, tr)
        lo, hi = dp.shard(B)
        nlo, nhi = dp.shard(node[0].shape[0])
        agent.set_noise([e[lo:hi] for e in eps])
        host = tuple(batch[f][lo:hi].numpy() for f in fields)
        node_np = tuple(t[nlo:nhi].numpy() for t in node) if updates % 10 == 0 else None
        n0, b0 = agent.dp.stats["all_reduce"]
        ret = agent.update_from_host(host, updates, node_np)
        torch.cuda.synchronize()
        p = "c%d_" % ci
        res[p + "ret"] = np.array(ret)
        res[p + "collectives"] = np.array([agent.dp.stats["all_reduce"][0] - n0, agent.dp.stats["all_reduce"][1] - b0])
        for name, mod in (("critic", agent.critic), ("policy", agent.policy), ("node", agent.neural_ode_model)):
            res[p + "p_" + name] = torch.cat([q.detach().reshape(-1) for q in mod.parameters()]).cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--out", required=True)
    ap.add_argument("--solver", default="euler")
    ap.add_argument("--env", default="Unicycle")
    ap.add_argument("--oracle-case", action="store_true", help="the oracle-checked update pattern instead of a golden fixture's")
    ap.add_argument("--adjoint", action="store_true")
    ap.add_argument("--step-control", default="global", help="dopri5 under data parallelism: global | shard")
    a = ap.parse_args()
    dist.init_process_group("gloo")
    dp = DataParallel(dist)
    out = "%s.rank%d.npz" % (a.out, dp.rank)
    if a.device == "cpu":
        run_cpu(dp, out)
    elif a.oracle_case:
        run_cuda_oracle_case(dp, out, a.solver, a.env, a.adjoint, a.step_control)
    else:
        run_cuda(dp, out, a.solver, a.env)
    dist.barrier()
    dist.destroy_process_group()
