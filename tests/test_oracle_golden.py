"""CPU: the oracle restatement reproduces what the reference itself computed.

The fixtures in tests/golden/ were written by oracle/gen_golden.py from the
reference's own modules; this test needs neither the reference nor a GPU.
Euler fixtures pin the oracle to the reference's real behaviour; rk4/dopri5
fixtures pin everything except the solver semantics (torchdiffeq is absent:
"parity unpinned" for those two, see oracle/nlbac_oracle.py).
"""
import numpy as np
import pytest
import torch

from common import case_inputs, load_golden, vec_close
from nlbac_amd import synth
from nlbac_amd.envspec import make_env
from oracle import nlbac_oracle as O

TOL = 2e-5  # fp32, same op order up to reduction details


def flat_sd(sd):
    return torch.cat([v.detach().reshape(-1) for v in sd.values()])


@pytest.mark.parametrize("env_name", ["Unicycle", "SimulatedCars", "UnicycleBarrier", "Pvtol", "PvtolBarrier"])
@pytest.mark.parametrize("solver", ["euler", "rk4", "dopri5"])
@pytest.mark.parametrize("B", [8, 128])
def test_oracle_matches_reference_fixture(solver, B, env_name):
    torch.set_num_threads(1)
    g = load_golden(solver, B, env_name)
    seed, hidden = int(g["meta_seed"]), int(g["meta_hidden"])
    env = synth.fixture_env(env_name, seed)
    args = O.Args(batch_size=B, hidden_size=hidden, seed=seed)
    if "meta_gamma_b" in g.files:
        args.gamma_b = float(g["meta_gamma_b"])
    agent = O.make_oracle(env, args, synth.agent_weights(env_name, hidden, seed), solver=solver)
    tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    for ci in range(len(g["meta_calls"])):
        batch, eps, node, updates = case_inputs(g, ci, tr)
        R = agent.update(batch, eps, updates, node_batch=node if updates % 10 == 0 else None)
        p = "c%d_" % ci
        vec_close(R["ret"], g[p + "ret"], TOL, p + "ret")
        vec_close(R["required"], g[p + "required"], TOL, p + "required")
        vec_close(R["lambdas"], g[p + "lambdas"], TOL, p + "lambdas")
        assert abs(R["augmented_term"] - float(g[p + "augmented_term"])) < 1e-12
        vec_close(R["x_next"], g[p + "x_next"], TOL, p + "x_next")
        if p + "brequired" in g.files:           # variants with a backup controller
            vec_close(R["brequired"], g[p + "brequired"], TOL, p + "brequired")
            vec_close(R["backup_lambdas"], g[p + "backup_lambdas"], TOL, p + "backup_lambdas")
            vec_close(R["bx_next"], g[p + "bx_next"], TOL, p + "bx_next")
        for key in ("x_next2", "bx_next2", "x_next3", "bx_next3"):
            if p + key in g.files:
                vec_close(R[key], g[p + key], TOL, p + key)
        if p + "backup_augmented_term" in g.files:
            assert abs(agent.backup_augmented_term - float(g[p + "backup_augmented_term"])) < 1e-12
            vec_close(agent.backup_lambda_values, g[p + "backup_lambdas"], TOL, p + "backup_lambdas (every call)")
        if B <= 16:
            vec_close(R["matr"], g[p + "matr"], TOL, p + "matr")
            if p + "bmatr" in g.files:
                vec_close(R["bmatr"], g[p + "bmatr"], TOL, p + "bmatr")
        if solver == "dopri5":
            st, gs = np.array(R["ode_info"]["steps"], dtype=np.float64), g[p + "ode_steps"]
            assert st.shape == gs.shape
            np.testing.assert_allclose(st[:, 0], gs[:, 0], rtol=1e-5)   # step sizes
            np.testing.assert_allclose(st[:, 1], gs[:, 1], rtol=2e-2)   # error ratio: cancellation noise
            np.testing.assert_array_equal(st[:, 2], gs[:, 2])           # accept flags
        for name, key in (("critic", "g_critic"), ("lya", "g_lya"), ("policy", "g_policy"),
                          ("backup", "g_backup"), ("node", "g_node"), ("barrier", "g_barrier")):
            if p + "g_%s_norm" % name not in g.files or key not in R:
                continue
            v = R[key]
            assert abs(float(v.double().norm()) / float(g[p + "g_%s_norm" % name]) - 1) < TOL
            vec_close(v[:48], g[p + "g_%s_head" % name], 5 * TOL, p + key + "_head")
            vec_close(v[-48:], g[p + "g_%s_tail" % name], 5 * TOL, p + key + "_tail")
        nets = [("critic", agent.critic), ("lya", agent.lya), ("policy", agent.policy), ("node", agent.node),
                ("critic_target", agent.critic_target), ("lya_target", agent.lya_target)]
        nets += ([("barrier", agent.barrier), ("barrier_target", agent.barrier_target)] if hasattr(agent, "barrier")
                 else [("backup", agent.backup)])
        for name, sd in nets:
            v = flat_sd(sd)
            assert abs(float(v.double().norm()) / float(g[p + "p_%s_norm" % name]) - 1) < 1e-6
            vec_close(v[:48], g[p + "p_%s_head" % name], TOL, p + "p_" + name)
        assert abs(float(agent.log_alpha) - float(g[p + "log_alpha"])) < 1e-6
        if p + "backup_log_alpha" in g.files:
            assert abs(float(agent.backup_log_alpha) - float(g[p + "backup_log_alpha"])) < 1e-6
