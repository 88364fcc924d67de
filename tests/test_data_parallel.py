"""Data-parallel path: world_size-2 runs (gloo, rendezvous on 127.0.0.1).

CPU test: the DataParallel exchange layer reproduces full-batch gradients from row shards
(global normalisation + SUM all-reduce), broadcasts parameters, and shards rows contiguously.
GPU test: two ranks sharing the test box's one MI355X run the product agent on half of a
golden-fixture minibatch each; every rank must land on the single-device reference outputs.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from common import load_golden, vec_close
from nlbac_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(world, args, timeout=300):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py")] + args, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])


def test_exchange_layer_world2_cpu(tmp_path):
    from oracle import nlbac_oracle as O
    out = str(tmp_path / "dp")
    launch(2, ["--device", "cpu", "--out", out])
    r0, r1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 32, 32, 64)
    np.testing.assert_array_equal(r0["flat"], r1["flat"])          # every rank holds the same reduced buffer
    np.testing.assert_array_equal(r0["theta"], r1["theta"])        # broadcast from rank 0
    # full-batch reference
    B, hidden, seed = 64, 64, 3
    W = synth.unicycle_agent_weights(hidden, seed)
    tr = synth.unicycle_transitions(256, seed=5)
    obs = torch.tensor(tr["obs"][:B], dtype=torch.float32)
    act = torch.tensor(tr["action"][:B], dtype=torch.float32)
    y = torch.tensor(tr["reward"][:B], dtype=torch.float32).unsqueeze(1)
    crit = {k: torch.tensor(v, requires_grad=True) for k, v in W["critic"].items()}
    q1, q2 = O.qnet(crit, obs, act)
    loss = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
    g = torch.cat([t.reshape(-1) for t in torch.autograd.grad(loss, list(crit.values()))])
    vec_close(r0["flat"][:-1], g.numpy(), 1e-5, "sharded grads")
    assert abs(float(r0["flat"][-1]) / float(loss) - 1) < 1e-5
    np.testing.assert_array_equal(r0["theta"], np.concatenate([v.reshape(-1) for v in W["critic"].values()]))


@pytest.mark.gpu
@pytest.mark.parametrize("env_name,solver", [("Unicycle", "euler"), ("Unicycle", "dopri5"), ("SimulatedCars", "rk4"),
                                             ("SimulatedCars", "dopri5"), ("UnicycleBarrier", "rk4"), ("Pvtol", "dopri5")])
def test_two_rank_sharded_update_matches_single_device_reference(tmp_path, env_name, solver):
    out = str(tmp_path / "dpgpu")
    launch(2, ["--device", "cuda", "--out", out, "--solver", solver, "--env", env_name], timeout=600)
    g = load_golden(solver, 128, env_name)
    for r in range(2):
        res = np.load(out + ".rank%d.npz" % r)
        for ci in range(len(g["meta_calls"])):
            p = "c%d_" % ci
            vec_close(res[p + "ret"], g[p + "ret"], 1e-4, "rank %d %s ret" % (r, p))
            vec_close(res[p + "required"], g[p + "required"], 1e-4, "rank %d %s required" % (r, p))
            vec_close(res[p + "lambdas"], g[p + "lambdas"], 1e-4, "rank %d %s lambdas" % (r, p))
            for name in ("critic", "policy", "node"):
                v = res[p + "p_" + name]
                assert abs(np.linalg.norm(v.astype(np.float64)) / float(g[p + "p_%s_norm" % name]) - 1) < 1e-5
                vec_close(v[:48], g[p + "p_%s_head" % name], 1e-4, "rank %d %s params %s" % (r, p, name))
    a, b = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    for k in a.files:                                            # replicas stay bit-identical
        np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.gpu
@pytest.mark.parametrize("env_name,solver,adjoint", [("Pvtol", "dopri5", True), ("QuadrotorLike", "dopri5", False)],
                         ids=["Pvtol-dopri5-adjoint", "QuadrotorLike-dopri5"])
def test_two_rank_sharded_update_matches_single_device_oracle(tmp_path, env_name, solver, adjoint):
    """BASELINE configs[3] (Pvtol, dopri5, adjoint backward: the adjoint's state / adjoint-state norms AND the parameter
    adjoint's stage derivatives are all-reduced so that every rank takes the same step decisions) and configs[4]
    (Quadrotor-like single-net NODE + learned barrier) on two ranks that share the test box's GPU, each with half of
    the rows, against the single-device ORACLE on the whole batch (no reference fixture exists for either)."""
    from oracle import nlbac_oracle as O
    from dp_worker import ORACLE_CASE, oracle_case_inputs
    from test_agent_parity_gpu import params_close
    from nlbac_amd.envspec import make_env
    out = str(tmp_path / "dporacle")
    launch(2, ["--device", "cuda", "--out", out, "--solver", solver, "--env", env_name, "--oracle-case"]
           + (["--adjoint"] if adjoint else []), timeout=900)
    c = ORACLE_CASE
    B, seed = c["B"], c["seed"]
    torch.set_num_threads(4)
    oargs = O.Args(batch_size=B, hidden_size=c["hidden"], seed=seed)
    oargs.gamma_b = c["gamma_b"][env_name]
    env = synth.fixture_env(env_name, seed)
    kw = dict(adjoint=True) if adjoint else {}
    oracle = O.make_oracle(env, oargs, synth.agent_weights(env_name, c["hidden"], seed), solver=solver, **kw)
    tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    res = [np.load(out + ".rank%d.npz" % r) for r in range(2)]

    class _Eps:                      # (oracle_case_inputs only needs the number of noise draws of the task)
        class task:
            n_eps = int(res[0]["n_eps"])
    lr = dict(critic=4e-4, policy=3e-4, node=1e-3)
    for ci, updates in enumerate(c["updates"]):
        batch, eps, node = oracle_case_inputs(env_name, _Eps, env, B, ci, updates, tr)
        R = oracle.update(batch, eps, updates, node_batch=node if updates % 10 == 0 else None)
        for r in range(2):
            vec_close(res[r]["c%d_ret" % ci], R["ret"], 1e-4, "rank %d ret (update %d)" % (r, updates))
            for name, osd in (("critic", oracle.critic), ("policy", oracle.policy), ("node", oracle.node)):
                ov = torch.cat([osd[k].detach().reshape(-1) for k in osd])
                params_close(res[r]["c%d_p_%s" % (ci, name)], ov, lr[name] * (ci + 1), "rank %d params %s (update %d)" % (r, name, updates))
    for k in res[0].files:                                            # replicas stay bit-identical
        np.testing.assert_array_equal(res[0][k], res[1][k])


@pytest.mark.gpu
@pytest.mark.parametrize("env_name,adjoint", [("Unicycle", False), ("Pvtol", True)], ids=["Unicycle", "Pvtol-adjoint"])
def test_two_rank_update_with_per_shard_step_control(tmp_path, env_name, adjoint):
    """``enable_data_parallel(step_control="shard")``: every rank controls the dopri5 steps of its own rows and no
    collective runs inside a solve.  What the two ranks compute is the single-device update whose solvers cut every
    problem into two row groups with step control of their own (``row_groups = 2``) — same solves row for row, the
    gradient sums differ by the summation order only (1e-5) — and it stays within the solver tolerance of the
    global-norm run (the oracle on the whole batch: 2e-3 on the returned losses, not the 1e-4 parity bar, which belongs
    to ``step_control="global"``)."""
    from oracle import nlbac_oracle as O
    from dp_worker import ORACLE_CASE, oracle_case_inputs
    from test_agent_parity_gpu import make_agent
    out = str(tmp_path / "dpshard")
    launch(2, ["--device", "cuda", "--out", out, "--solver", "dopri5", "--env", env_name, "--oracle-case",
               "--step-control", "shard"] + (["--adjoint"] if adjoint else []), timeout=900)
    res = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    c = ORACLE_CASE
    B, seed = c["B"], c["seed"]
    agent, env = make_agent(B, c["hidden"], seed, "dopri5", env_name, c["gamma_b"][env_name])
    agent.adjoint = bool(adjoint)
    for sv in agent.task.solvers:
        # (the parameter adjoint of a NODE fit is one vector with one step control per solve: the adjoint fit cannot be
        # cut into row groups on one device — for that case the fit's own shards are only held to the solver tolerance)
        if not (adjoint and sv is agent.task.fit_solver):
            sv.row_groups = 2
    strict = not adjoint
    torch.set_num_threads(4)
    oargs = O.Args(batch_size=B, hidden_size=c["hidden"], seed=seed)
    oargs.gamma_b = c["gamma_b"][env_name]
    kw = dict(adjoint=True) if adjoint else {}
    oracle = O.make_oracle(synth.fixture_env(env_name, seed), oargs, synth.agent_weights(env_name, c["hidden"], seed),
                           solver="dopri5", **kw)
    tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    fields = synth.fields(env_name)
    lr = dict(critic=4e-4, policy=3e-4, node=1e-3)
    for ci, updates in enumerate(c["updates"]):
        batch, eps, node = oracle_case_inputs(env_name, agent, env, B, ci, updates, tr)
        agent.set_noise(eps)
        node_np = tuple(t.numpy() for t in node) if updates % 10 == 0 else None
        ret = np.array(agent.update_from_host(tuple(batch[f].numpy() for f in fields), updates, node_np))
        torch.cuda.synchronize()
        R = oracle.update(batch, eps, updates, node_batch=node if updates % 10 == 0 else None)
        for r in range(2):
            vec_close(res[r]["c%d_ret" % ci], ret, (1e-5 if ci == 0 else 1e-4) if strict else 2e-3,
                      "rank %d ret vs row groups (update %d)" % (r, updates))
            vec_close(res[r]["c%d_ret" % ci], R["ret"], 2e-3, "rank %d ret vs global-norm oracle (update %d)" % (r, updates))
            for name, mod in (("critic", agent.critic), ("policy", agent.policy), ("node", agent.neural_ode_model)):
                v = torch.cat([q.detach().reshape(-1) for q in mod.parameters()]).cpu().numpy()
                # (Adam's first steps move every weight by ~lr whatever its gradient's size: a weight whose gradient is
                # the cancellation residue of the batch sum lands up to 2 lr apart when the summation order changes —
                # the policy's, at these first updates, in under 2 % of its entries; all others agree to 1e-5)
                err = np.abs(res[r]["c%d_p_%s" % (ci, name)].astype(np.float64) - v)
                # (first update; afterwards the few weights that did land apart feed the next update's every output)
                if ci == 0 and (strict or name != "node"):
                    assert (err > 1e-5 * np.abs(v).max()).mean() <= 2e-2, (name, updates, (err > 1e-5 * np.abs(v).max()).mean())
                if strict or name != "node":
                    assert np.linalg.norm(err) <= 1e-3 * np.linalg.norm(v.astype(np.float64)), (name, updates)
                assert err.max() <= 2.1 * lr[name] * (ci + 1), (name, updates, err.max())
    for k in res[0].files:                                            # replicas stay bit-identical
        np.testing.assert_array_equal(res[0][k], res[1][k])
    # exchanges per update under per-shard step control: NONE inside a solve.  What is left is what the dependencies
    # force — the critic gradient (+ loss sums) before its Adam step, the constraint / actor sums before the
    # augmented-Lagrangian scalars (nonlinear in the global value), the actors' gradient before theirs — plus, on a
    # NODE-fit update, the fit's gradient (+ its loss sum): a collective of its own because the rollout, which is queued
    # before the critic exchange exists, needs the stepped NODE.
    for ci, updates in enumerate(c["updates"]):
        n, nbytes = (int(x) for x in res[0]["c%d_collectives" % ci])
        print("update %d: %d all-reduces, %d bytes" % (updates, n, nbytes))
        n_fit = 1 if updates % 10 == 0 else 0
        assert n <= 3 + len(agent.actor_groups) - 1 + n_fit, (updates, n)


@pytest.mark.gpu
def test_rccl_branch_of_the_exchange_layer_runs_on_one_gpu():
    """``nccl`` (RCCL) initialised in a fresh process before any other GPU call, world size 1, the exchange layer's
    one-rank short-cut switched off: device tensors go through ``all_reduce_`` / ``broadcast_`` on the RCCL branch (the
    gloo tests stage through the host), and an agent in data-parallel mode broadcasts its arenas over it."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_worker.py"), str(port)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
