"""BASELINE configs[4] — "Quadrotor (safe-control-gym) + neural barrier certificate".  The reference holds NO code for
it (empty submodule; prose at /root/reference/README.md:66-72, 190-192), so there is NO REFERENCE PARITY to test: the
task is this build's reading of the prose (envspec.QuadrotorLikeSpec, tasks.QuadrotorBarrierTask) and the device path
is checked against the oracle's restatement of the same reading (``OracleQuadrotorLikeAgent``) only.

CPU part: the oracle's normalised single-net NODE against the same net with the normalisation folded into its first
and last layer (exact algebra), and one oracle update end to end.  GPU part: whole updates, all three solvers, B = 8 /
128, and the BASELINE size B = 32768."""
import numpy as np
import pytest
import torch

from nlbac_amd import synth
from nlbac_amd.envspec import make_env

ENV = "QuadrotorLike"


def _oracle(B, hidden, solver, seed=0):
    from oracle import nlbac_oracle as O
    oargs = O.Args(batch_size=B, hidden_size=hidden, seed=seed)
    oargs.gamma_b = 1.0
    return O.make_oracle(make_env(ENV, seed), oargs, synth.agent_weights(ENV, hidden, seed), solver=solver)


def test_normalised_node_equals_the_net_with_folded_normalisation():
    from oracle import nlbac_oracle as O
    env = make_env(ENV, 0)
    W = synth.agent_weights(ENV, 64, 0)["node"]
    sd = {k: torch.tensor(v, dtype=torch.float64) for k, v in W.items()}
    im, isd, om, osd = (torch.tensor(np.asarray(v), dtype=torch.float64) for v in env.node_normalizer)
    node = O.ConcatNode(sd, n_s=6, n_carry=2, norm=None)
    node.norm = (im, 1.0 / isd, om, osd)
    folded = dict(sd)
    folded["net.0.weight"] = sd["net.0.weight"] / isd[None, :]
    folded["net.0.bias"] = sd["net.0.bias"] - sd["net.0.weight"] @ (im / isd)
    folded["net.6.weight"] = sd["net.6.weight"] * osd[:, None]
    folded["net.6.bias"] = sd["net.6.bias"] * osd + om
    plain = O.ConcatNode(folded, n_s=6, n_carry=2)
    tr = synth.transitions(ENV, 256, seed=2, env=env)
    s = torch.tensor(np.concatenate([tr["obs"], tr["action"]], 1))
    a, b = node(0.0, s), plain(0.0, s)
    assert float((a - b).abs().max()) <= 1e-10 * float(b.abs().max())
    assert float(a[:, 6:].abs().max()) == 0.0          # the action columns are carried


@pytest.mark.parametrize("solver", ["euler", "dopri5"])
def test_oracle_update_runs_and_trains_the_barrier(solver):
    B = 64
    env = make_env(ENV, 0)
    agent = _oracle(B, 64, solver)
    tr = synth.transitions(ENV, 1024, seed=1, env=env)
    assert set(np.unique(tr["barrier_signal"])) <= {0.0, -1.0, -10.0, -11.0}     # D1, D2 (README.md:190)
    idx = np.arange(B)
    batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in synth.fields(ENV)}
    eps = [torch.from_numpy(e) for e in synth.normal_eps(3, B, 2, seed=0)]
    node = tuple(torch.tensor(tr[f][:256], dtype=torch.float32) for f in ("obs", "action", "next_obs"))
    R = agent.update(batch, eps, 0, node_batch=node)
    assert all(np.isfinite(R["ret"])) and R["x_next"].shape == (B, 6) and np.isfinite(R["barrier_loss"])
    assert float(R["g_node"].abs().max()) > 0 and float(R["g_policy"].abs().max()) > 0


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["euler", "rk4", "dopri5"])
@pytest.mark.parametrize("B", [8, 128])
def test_update_matches_oracle(solver, B):
    from common import vec_close
    from test_agent_parity_gpu import make_agent, params_close, flat_params, flat_grad
    from nlbac_amd.sac_cbf_clf import _layout as SC
    torch.set_num_threads(4)
    hidden, seed, TOL = 64, 0, 1e-4
    agent, env = make_agent(B, hidden, seed, solver, ENV, 1.0)
    oracle = _oracle(B, hidden, solver, seed)
    tr = synth.transitions(ENV, 4096, seed=seed + 1, env=env)
    lr = dict(critic=4e-4, policy=3e-4, node=1e-3, barrier=4e-4)
    for ci, updates in enumerate((0, 1, 8)):
        rs = np.random.RandomState(10 + ci)
        idx, nidx = rs.choice(4096, B, replace=False), rs.choice(4096, 777, replace=False)
        batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in synth.fields(ENV)}
        eps = [torch.from_numpy(e) for e in synth.normal_eps(3, B, 2, seed=ci)]
        node = tuple(torch.tensor(tr[f][nidx], dtype=torch.float32) for f in ("obs", "action", "next_obs"))
        with_fit = updates % 10 == 0
        R = oracle.update(batch, eps, updates, node_batch=node if with_fit else None)
        agent.set_noise(eps)
        ret = agent.update_from_host(tuple(batch[f].numpy() for f in synth.fields(ENV)), updates,
                                     tuple(t.numpy() for t in node) if with_fit else None)
        torch.cuda.synchronize()
        p = "update %d: " % updates
        sc = agent.sc.cpu().numpy()
        vec_close(ret, R["ret"], TOL, p + "returned floats")
        vec_close(sc[SC.SC_REQ:SC.SC_REQ + 2], R["required"].numpy(), TOL, p + "required")
        vec_close(agent.lambda_values, R["lambdas"], TOL, p + "lambdas")
        assert abs(agent.augmented_term - R["augmented_term"]) < 1e-12
        vec_close(agent.node_solver.ctx["out"].cpu().numpy(), R["x_next"].numpy(), TOL, p + "x_next")
        vec_close(flat_grad(agent, agent.ar_a, agent.policy), R["g_policy"], TOL, p + "policy gradient")
        if with_fit:
            vec_close(flat_grad(agent, agent.ar_n, agent.neural_ode_model), R["g_node"], 1e-3 if solver == "dopri5" else TOL,
                      p + "NODE-fit gradient (normalised inputs / de-normalised outputs)")
        for name, mod, osd in (("critic", agent.critic, oracle.critic), ("policy", agent.policy, oracle.policy),
                               ("barrier", agent.BarrierNet, oracle.barrier), ("node", agent.neural_ode_model, oracle.node)):
            ov = torch.cat([osd[k].detach().reshape(-1) for k in osd])
            params_close(flat_params(mod), ov, lr[name] * (ci + 1), p + "params " + name)


@pytest.mark.gpu
def test_baseline_size_B32768():
    """configs[4]'s size on one GPU: batch 32768, hidden 256, dopri5, NODE fit on 32768 rows — the six returned floats,
    `required`, the NODE-fit and policy gradient norms against the oracle; two agents land on identical parameters."""
    from common import vec_close
    from test_agent_parity_gpu import make_agent, flat_grad
    from nlbac_amd.sac_cbf_clf import _layout as SC
    B, H, solver = 32768, 256, "dopri5"
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    oracle = _oracle(B, H, solver)
    agents = [make_agent(B, H, 0, solver, ENV, 1.0)[0] for _ in range(2)]
    env = make_env(ENV, 0)
    tr = synth.transitions(ENV, B, seed=3, env=env)
    for u in (0, 1):
        idx = np.random.RandomState(u).permutation(B)
        batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in synth.fields(ENV)}
        eps = [torch.from_numpy(e) for e in synth.normal_eps(3, B, 2, seed=u)]
        node = tuple(torch.tensor(tr[f], dtype=torch.float32) for f in ("obs", "action", "next_obs")) if u == 0 else None
        R = oracle.update(batch, eps, u, node_batch=node)
        rets = []
        for a in agents:
            a.set_noise(eps)
            rets.append(a.update_from_host(tuple(batch[f].numpy() for f in synth.fields(ENV)), u,
                                           tuple(x.numpy() for x in node) if node else None))
        worst = max(abs(a - b) / (abs(b) + 1e-3) for a, b in zip(rets[0], R["ret"]))
        assert worst < 1e-4, "update %d: max rel err vs oracle %.2e" % (u, worst)
        assert rets[0] == rets[1]
        a = agents[0]
        vec_close(a.sc.cpu().numpy()[SC.SC_REQ:SC.SC_REQ + 2], R["required"].numpy(), 1e-4, "required")
        gp = flat_grad(a, a.ar_a, a.policy)
        assert abs(float(gp.double().norm()) / float(R["g_policy"].double().norm()) - 1) < 1e-3
        if u == 0:
            gn = flat_grad(a, a.ar_n, a.neural_ode_model)
            assert abs(float(gn.double().norm()) / float(R["g_node"].double().norm()) - 1) < 1e-3
    torch.cuda.synchronize()
    for x, y in zip(agents[0].arenas, agents[1].arenas):
        assert torch.equal(x.theta, y.theta)
