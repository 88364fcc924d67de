"""GPU: the HIP update path vs (a) golden fixtures produced by the reference
itself and (b) the CPU oracle, on identical seeded minibatches, weights and
policy noise.  Tolerance 1e-4 relative to each tensor's scale (BASELINE.json
north_star); step sizes / error ratios of dopri5 get the looser bounds noted.
"""
import numpy as np
import pytest
import torch

from common import BATCH_FIELDS, case_inputs, load_golden, vec_close
from nlbac_amd import synth
from nlbac_amd.envspec import make_env

pytestmark = pytest.mark.gpu
TOL = 1e-4


def make_agent(B, hidden, seed, solver, env_name="Unicycle", gamma_b=None, env=None):
    from oracle.nlbac_oracle import Args
    if env_name.endswith("Barrier") or env_name == "QuadrotorLike":
        from nlbac_amd.neural_barrier_certificate.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
    else:
        from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
    env = env if env is not None else synth.fixture_env(env_name, seed)
    args = Args(batch_size=B, hidden_size=hidden, seed=seed, cuda=True)
    if gamma_b is not None:
        args.gamma_b = gamma_b
    agent = SAC_CBF_CLF(env.obs_dim, env.action_space, env, args)
    agent.solver = solver
    W = synth.agent_weights(env_name, hidden, seed)
    t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    agent.critic.load_state_dict(t(W["critic"]))
    agent.critic_target.load_state_dict(t(W["critic"]))
    agent.lyapunovNet.load_state_dict(t(W["lyapunov"]))
    agent.lyapunovNet_target.load_state_dict(t(W["lyapunov"]))
    agent.policy.load_state_dict(t(W["policy"]))
    if "backup_policy" in W:
        agent.backup_policy.load_state_dict(t(W["backup_policy"]))
    if "barrier" in W:
        agent.BarrierNet.load_state_dict(t(W["barrier"]))
        agent.BarrierNet_target.load_state_dict(t(W["barrier"]))
    agent.neural_ode_model.load_state_dict(t(W["node"]))
    agent.repack_all()
    return agent, env


GRAD_FLOOR = 1e-6


def params_close(v, ov, step_bound, name, gmin=None):
    """Post-Adam parameters.  Adam's update is lr * m_hat / (sqrt(v_hat) + 1e-8): where every gradient an entry has
    seen so far is of the order of that 1e-8, an ABSOLUTE gradient difference of 1e-10 — the cancellation residue of a
    batch sum whose terms are 1e-3 — moves the update by a sizeable fraction of lr.  Bar: every entry within TOL of the
    tensor's scale, except at most 0.5 % of the entries, which must still be within the accumulated Adam step bound
    (lr per update) — and, where the oracle's gradients are at hand (``gmin``: the smallest |g| the oracle saw for the
    entry over the updates so far), every such entry must be one whose gradient did fall below ``GRAD_FLOOR``."""
    v, ov = np.asarray(v, dtype=np.float64), np.asarray(ov, dtype=np.float64)
    err = np.abs(v - ov)
    scale = np.abs(ov).max()
    bad = err > TOL * scale
    assert bad.mean() <= 5e-3, "%s: %.4f %% of the entries off by more than %.0e" % (name, 100 * bad.mean(), TOL)
    assert err.max() <= step_bound, "%s: max abs err %.3e beyond the Adam step bound %.1e" % (name, err.max(), step_bound)
    if gmin is not None and bad.any():
        assert (np.asarray(gmin)[bad] < GRAD_FLOOR).all(), (
            "%s: %d entries beyond %.0e whose oracle gradient never fell below %.0e (largest such |g| %.3e)"
            % (name, int((np.asarray(gmin)[bad] >= GRAD_FLOOR).sum()), TOL, GRAD_FLOOR, np.asarray(gmin)[bad].max()))


def flat_params(module):
    return torch.cat([p.detach().reshape(-1) for p in module.parameters()]).cpu()


def flat_sd(sd, order):
    return torch.cat([sd[k].detach().reshape(-1) for k in order]).cpu()


def flat_grad(agent, arena, module, n_slabs=None):
    return torch.cat([arena.grad_view(p).reshape(-1) for p in module.parameters()]).cpu()


CASES = [("Unicycle", False), ("Unicycle", True), ("SimulatedCars", False), ("UnicycleBarrier", False),
         ("UnicycleBarrier", True), ("Pvtol", False), ("PvtolBarrier", False)]
IDS = ["unicycle-eager", "unicycle-hipgraph", "cars-eager", "nbc-unicycle-eager", "nbc-unicycle-hipgraph",
       "pvtol-eager", "nbc-pvtol-eager"]


@pytest.mark.parametrize("env_name,graphs", CASES, ids=IDS)
@pytest.mark.parametrize("solver", ["euler", "rk4", "dopri5"])
@pytest.mark.parametrize("B", [8, 128])
def test_update_matches_reference_fixture_and_oracle(solver, B, env_name, graphs):
    from oracle import nlbac_oracle as O
    torch.set_num_threads(4)
    g = load_golden(solver, B, env_name)
    seed, hidden = int(g["meta_seed"]), int(g["meta_hidden"])
    gamma_b = float(g["meta_gamma_b"]) if "meta_gamma_b" in g.files else 50.0
    agent, env = make_agent(B, hidden, seed, solver, env_name, gamma_b)
    agent.use_graphs = graphs      # call 0 warms up eagerly, calls 1 and 2 capture + replay hipGraphs
    oargs = O.Args(batch_size=B, hidden_size=hidden, seed=seed)
    oargs.gamma_b = gamma_b
    oracle = O.make_oracle(synth.fixture_env(env_name, seed), oargs, synth.agent_weights(env_name, hidden, seed), solver=solver)
    tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    n_cbf = agent.num_cbfs
    lr = dict(critic=4e-4, policy=3e-4, node=1e-3)
    from nlbac_amd.sac_cbf_clf import _layout as SC
    gmins = {}
    for ci in range(len(g["meta_calls"])):
        batch, eps, node, updates = case_inputs(g, ci, tr)
        with_fit = updates % 10 == 0
        R = oracle.update(batch, eps, updates, node_batch=node if with_fit else None)
        agent.set_noise(eps)
        host_batch = tuple(batch[f].numpy() for f in synth.fields(env_name))
        node_np = tuple(t.numpy() for t in node) if with_fit else None
        ret = agent.update_from_host(host_batch, updates, node_np)
        torch.cuda.synchronize()
        p = "c%d_" % ci
        sc = agent.sc.cpu().numpy()
        ws = agent._ws[B]
        # ---- against the reference-generated fixture
        vec_close(ret, g[p + "ret"], TOL, p + "ret vs golden")
        vec_close(sc[SC.SC_REQ:SC.SC_REQ + n_cbf + 1], g[p + "required"], TOL, p + "required vs golden")
        vec_close(agent.lambda_values, g[p + "lambdas"], TOL, p + "lambdas vs golden")
        backup = p + "brequired" in g.files
        if backup:
            vec_close(sc[SC.SC_BREQ:SC.SC_BREQ + n_cbf], g[p + "brequired"], TOL, p + "brequired vs golden")
            vec_close(agent.backup_lambda_values, g[p + "backup_lambdas"], TOL, p + "blambdas vs golden")
        assert abs(agent.augmented_term - float(g[p + "augmented_term"])) < 1e-12
        xn = agent.node_solver.ctx["out"].cpu().numpy()
        vec_close(xn[:B], g[p + "x_next"], TOL, p + "x_next vs golden")
        if backup:
            vec_close(xn[B:], g[p + "bx_next"], TOL, p + "bx_next vs golden")
        if env_name == "Pvtol":
            vec_close(ws.x2[:B].cpu().numpy(), g[p + "x_next2"], TOL, p + "x_next2 vs golden")
            vec_close(ws.x3[:B].cpu().numpy(), g[p + "x_next3"], TOL, p + "x_next3 vs golden")
            if backup:
                vec_close(ws.x3[B:].cpu().numpy(), g[p + "bx_next3"], TOL, p + "bx_next3 vs golden")
            assert abs(agent.backup_augmented_term - float(g[p + "backup_augmented_term"])) < 1e-12
            vec_close(agent.backup_lambda_values, g[p + "backup_lambdas"], TOL, p + "blambdas (every call)")
        elif p + "x_next2" in g.files:
            xn2 = agent.task.solver2.ctx["out"].cpu().numpy()
            vec_close(xn2[:B], g[p + "x_next2"], TOL, p + "x_next2 vs golden")
            vec_close(xn2[B:], g[p + "bx_next2"], TOL, p + "bx_next2 vs golden")
        if B <= 16:
            vec_close(ws.matr.cpu().numpy(), g[p + "matr"], TOL, p + "matr vs golden")
            if backup:
                vec_close(ws.bmatr.cpu().numpy(), g[p + "bmatr"], TOL, p + "bmatr vs golden")
        if solver == "dopri5" and "info" in agent.node_solver.ctx and not (graphs and ci > 0):
            info = agent.node_solver.ctx["info"]
            for prob, key in ((0, "ode_steps"), (1, "bode_steps"))[:2 if backup else 1]:
                st = np.array([a[prob] for a in info], dtype=np.float64)
                gs = g[p + key]
                assert st.shape == gs.shape
                np.testing.assert_allclose(st[:, 0], gs[:, 0], rtol=1e-4)
                # error ratio = |y5 - y4| / tol: a difference of nearly equal fp32 numbers (cancellation noise);
                # ratios << 1 (cars: 1e-5) are rounding residue, hence the absolute floor
                np.testing.assert_allclose(st[:, 1], gs[:, 1], rtol=5e-2, atol=1e-4)
                np.testing.assert_array_equal(st[:, 2], gs[:, 2])
        for name, ar, mod in (("critic", agent.ar_c, agent.critic), ("lya", agent.ar_c, agent.lyapunovNet),
                              ("policy", agent.ar_a, agent.policy), ("backup", agent.pol_arena[-1], agent.backup_policy),
                              ("barrier", agent.ar_c, agent.BarrierNet), ("node", agent.ar_n, agent.neural_ode_model)):
            if p + "g_%s_norm" % name not in g.files:
                continue
            v = flat_grad(agent, ar, mod)
            assert abs(float(v.double().norm()) / float(g[p + "g_%s_norm" % name]) - 1) < TOL, name
            okey = "g_" + name
            if okey in R:
                vec_close(v, R[okey], TOL, p + "grad %s vs oracle (full vector)" % name)
        order = None
        for name, mod in (("critic", agent.critic), ("lya", agent.lyapunovNet), ("policy", agent.policy),
                          ("backup", agent.backup_policy), ("barrier", agent.BarrierNet),
                          ("node", agent.neural_ode_model)):
            if mod is None:
                continue
            v = flat_params(mod)
            assert abs(float(v.double().norm()) / float(g[p + "p_%s_norm" % name]) - 1) < 1e-5
            vec_close(v[:48], g[p + "p_%s_head" % name], TOL, p + "params " + name)
            vec_close(v[-48:], g[p + "p_%s_tail" % name], TOL, p + "params tail " + name)
        targets = [("critic_target", agent.critic_target), ("lya_target", agent.lyapunovNet_target)]
        if agent.BarrierNet is not None:
            targets.append(("barrier_target", agent.BarrierNet_target))
        for name, tv in targets:
            sd = tv.state_dict()
            v = torch.cat([sd[k].reshape(-1) for k in sd]).cpu()
            assert abs(float(v.double().norm()) / float(g[p + "p_%s_norm" % name]) - 1) < 1e-5
            vec_close(v[:48], g[p + "p_%s_head" % name], TOL, p + "params " + name)
        assert abs(float(agent.log_alpha) - float(g[p + "log_alpha"])) < 1e-5
        if backup:
            assert abs(float(agent.backup_log_alpha) - float(g[p + "backup_log_alpha"])) < 1e-5
        # ---- against the oracle on the full tensors
        vec_close(ret, R["ret"], TOL, p + "ret vs oracle")
        vec_close(ws.pi2[:B].cpu().numpy(), R["pi"], TOL, p + "pi vs oracle")
        vec_close(ws.logp2[:B].cpu().numpy(), R["log_pi"].reshape(-1), TOL, p + "log_pi vs oracle")
        vec_close(ws.next_q.cpu().numpy(), R["next_q"].reshape(-1), TOL, p + "next_q vs oracle")
        for name, mod, osd in (("critic", agent.critic, oracle.critic), ("policy", agent.policy, oracle.policy),
                               ("node", agent.neural_ode_model, oracle.node)):
            ov = torch.cat([osd[k].detach().reshape(-1) for k in osd])
            if "g_" + name in R:       # smallest oracle |gradient| of every entry over the updates so far
                ga = np.abs(np.asarray(R["g_" + name], dtype=np.float64)).reshape(-1)
                gmins[name] = ga if name not in gmins else np.minimum(gmins[name], ga)
            params_close(flat_params(mod), ov, lr[name] * (ci + 1), p + "all params %s vs oracle" % name, gmins.get(name))


@pytest.mark.parametrize("solver", ["euler", "dopri5"])
def test_pvtol_with_the_reference_env_defaults(solver):
    """The fixtures run Pvtol with a tighter corridor / slower safety operator (synth.FIXTURE_ENV) so that every barrier
    family is active and well conditioned.  This is the env bench.py runs — the reference's own constants (|y| < 100,
    operator_dist 1.0, follow 0.7): the well-conditioned quantities (returned floats, `required`, the three predicted
    states, multipliers, augmented terms) against the oracle at 1e-4; the policy gradient, whose y / operator barrier
    part is a 1000:1 fp32 cancellation here, relative to its own scale with an absolute floor."""
    from oracle import nlbac_oracle as O
    from nlbac_amd.sac_cbf_clf import _layout as SC
    torch.set_num_threads(4)
    B, hidden, seed, env_name = 128, 64, 0, "Pvtol"
    env = make_env(env_name, seed)
    assert (env.y_max, env.operator_dist, env.safety_operator_follow) == (100.0, 1.0, 0.7)
    agent, _ = make_agent(B, hidden, seed, solver, env_name, 0.8, env=env)
    oargs = O.Args(batch_size=B, hidden_size=hidden, seed=seed)
    oargs.gamma_b = 0.8
    oracle = O.make_oracle(make_env(env_name, seed), oargs, synth.agent_weights(env_name, hidden, seed), solver=solver)
    tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    fields = synth.fields(env_name)
    n_cbf = agent.num_cbfs
    for ci, updates in enumerate((0, 1, 20)):
        rs = np.random.RandomState(40 + ci)
        idx, nidx = rs.choice(4096, B, replace=False), rs.choice(4096, 512, replace=False)
        batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in fields}
        eps = [torch.from_numpy(e) for e in synth.normal_eps(7, B, 2, seed=ci)]
        node = tuple(torch.tensor(tr[f][nidx], dtype=torch.float32) for f in ("obs", "action", "next_obs"))
        with_fit = updates % 10 == 0
        R = oracle.update(batch, eps, updates, node_batch=node if with_fit else None)
        agent.set_noise(eps)
        ret = agent.update_from_host(tuple(batch[f].numpy() for f in fields), updates,
                                     tuple(t.numpy() for t in node) if with_fit else None)
        torch.cuda.synchronize()
        p = "update %d: " % updates
        sc, ws = agent.sc.cpu().numpy(), agent._ws[B]
        vec_close(ret, R["ret"], TOL, p + "returned floats")
        vec_close(sc[SC.SC_REQ:SC.SC_REQ + n_cbf + 1], R["required"].numpy(), TOL, p + "required")
        vec_close(agent.lambda_values, R["lambdas"], TOL, p + "lambdas")
        assert abs(agent.augmented_term - R["augmented_term"]) < 1e-12
        for name, dev in (("x_next", ws.x1[:B]), ("x_next2", ws.x2[:B]), ("x_next3", ws.x3[:B])):
            vec_close(dev.cpu().numpy(), R[name].numpy(), TOL, p + name)
        if "brequired" in R:
            vec_close(sc[SC.SC_BREQ:SC.SC_BREQ + n_cbf], R["brequired"].numpy(), TOL, p + "brequired")
            vec_close(ws.x3[B:].cpu().numpy(), R["bx_next3"].numpy(), TOL, p + "bx_next3")
        g, go = flat_grad(agent, agent.ar_a, agent.policy).double().numpy(), R["g_policy"].double().numpy()
        floor = 1e-6 * max(1.0, float(np.abs(R["required"].numpy()).max()))
        err = np.abs(g - go)
        assert (err <= 1e-3 * np.abs(go).max() + floor).all(), p + "policy gradient: worst %.3e (scale %.3e)" % (
            err.max(), np.abs(go).max())
        assert np.linalg.norm(g - go) <= 1e-3 * np.linalg.norm(go) + floor, p + "policy gradient (L2)"


@pytest.mark.parametrize("solver", ["euler", "dopri5"])
def test_hipgraph_replay_is_bit_identical_to_eager(solver):
    """12 consecutive updates (NODE fit at updates 0 and 10, lambda updates at 0 and 8): the captured graphs
    replay the same kernels with the same arguments, so every parameter must match bit for bit."""
    B, hidden, seed = 256, 256, 0
    tr = synth.unicycle_transitions(4096, seed=3)
    fields = ("obs", "action", "reward", "constraint", "center", "next_center", "next_obs", "mask")
    agents = []
    for graphs in (False, True):
        agent, env = make_agent(B, hidden, seed, solver)
        agent.use_graphs = graphs
        rs = np.random.RandomState(5)
        rets = []
        for updates in range(12):
            idx = rs.choice(4096, B, replace=False)
            nidx = rs.choice(4096, 1024, replace=False)
            agent.set_noise(synth.normal_eps(3, B, 2, seed=updates))
            host = tuple(tr[f][idx] for f in fields)
            node = tuple(tr[f][nidx] for f in ("obs", "action", "next_obs")) if updates % 10 == 0 else None
            rets.append(agent.update_from_host(host, updates, node))
        torch.cuda.synchronize()
        agents.append((agent, rets))
    (a0, r0), (a1, r1) = agents
    assert len(a1._ws[B].graphs) >= 3, "graphs were not captured"
    np.testing.assert_array_equal(np.array(r0), np.array(r1))
    for ar0, ar1 in ((a0.ar_c, a1.ar_c), (a0.ar_a, a1.ar_a), (a0.ar_n, a1.ar_n)):
        assert torch.equal(ar0.theta, ar1.theta) and torch.equal(ar0.m, ar1.m) and torch.equal(ar0.v, ar1.v)
    assert torch.equal(a0.ar_c.target, a1.ar_c.target)
    assert torch.equal(a0.sc, a1.sc)


@pytest.mark.parametrize("env_name,B", [("Unicycle", 256), ("Unicycle", 8), ("UnicycleBarrier", 128), ("Pvtol", 128)])
def test_deferred_head_sums_match_the_elections(env_name, B):
    """nlbac_dy_head::sums_defer / finish (the td head's and the actor-q head's batch sums finished by two workgroups of
    the actors' data backward instead of by elections at the end of their own launches): the same tile partials summed
    in the same order, so the returned losses, the temperatures' gradient and every parameter match bit for bit —
    B = 8: one 16-row tile, every job on the launch's only tile; UnicycleBarrier: four critic nets (out_x).  Unicycle: the
    constraint head the same way (nlbac_gauss_head::cf_defer: tile column sums out; nlbac_dy_head::cb_defer: the constraint
    backward's workgroups run the augmented-Lagrangian step privately; job kind 4 commits the staged block) against its
    election + in-launch step: multipliers, rho and coefficients in the scalars block bit for bit."""
    hidden, seed = 256, 0
    gamma_b = {"Unicycle": 50.0, "Pvtol": 0.8, "UnicycleBarrier": 5.0}[env_name]
    fields = synth.fields(env_name)
    runs = []
    for defer in (True, False):
        agent, env = make_agent(B, hidden, seed, "euler", env_name, gamma_b)
        agent.sums_defer = defer
        tr = synth.transitions(env_name, 4096, seed=3, env=env)
        rs = np.random.RandomState(5)
        rets = []
        for updates in (7, 8, 9, 16):          # (8, 16: the multipliers are stepped — the staged block carries new lambdas)
            idx = rs.choice(4096, B, replace=False)
            agent.set_noise(synth.normal_eps(agent.task.n_eps, B, env.n_u, seed=updates))
            rets.append(agent.update_from_host(tuple(tr[f][idx] for f in fields), updates, None))
        torch.cuda.synchronize()
        runs.append((agent, rets))
    (a0, r0), (a1, r1) = runs
    np.testing.assert_array_equal(np.array(r0), np.array(r1))
    for x, y in zip(a0.arenas, a1.arenas):
        assert torch.equal(x.theta, y.theta) and torch.equal(x.m, y.m)
    assert torch.equal(a0.sc, a1.sc)
    assert int(a0._ws[B].sums_tiles[0]) > 0 and int(a1._ws[B].sums_tiles[0]) == 0, "the deferred form was not the one that ran"


@pytest.mark.parametrize("env_name,solver", [("Unicycle", "dopri5"), ("Unicycle", "euler"), ("Pvtol", "dopri5"),
                                             ("UnicycleBarrier", "dopri5")])
def test_folded_launches_match_the_launches_they_replace(env_name, solver):
    """The per-row steps evaluated inside the MLP / solver launches (Gaussian head, dy heads, in- / out-map, results
    written to pinned memory by the kernels) against the same update with every step as the launch of its own
    (``fold_launches = False``): identical row arithmetic, so the per-row outputs left in memory are bit-identical on the
    first update (up to the contraction of the look-ahead map's multiply-adds); the batch sums are taken in a different order (per 32-row tile instead of per 256-row block), so the
    losses, the temperatures' gradient and — through it — the parameters agree to rounding.  With the folds off the dopri5
    solves also run their norms the classical way (fused with an election for f0 / the probe, a launch over the error
    rows for an attempt: ``solver.norm_defer = False``, tasks.reserve) and their interpolation as launches."""
    B, hidden, seed = 256, 256, 0
    gamma_b = {"Unicycle": 50.0, "Pvtol": 0.8, "UnicycleBarrier": 5.0}[env_name]
    fields = synth.fields(env_name)
    node_fields = ("obs", "action", "next_obs")
    runs = []
    for fold in (True, False):
        agent, env = make_agent(B, hidden, seed, solver, env_name, gamma_b)
        agent.fold_launches = fold
        tr = synth.transitions(env_name, 4096, seed=3, env=env)
        rs = np.random.RandomState(5)
        rets, first = [], None
        for updates in range(4):
            idx = rs.choice(4096, B, replace=False)
            nidx = rs.choice(4096, 1024, replace=False)
            agent.set_noise(synth.normal_eps(agent.task.n_eps, B, env.n_u, seed=updates))
            host = tuple(tr[f][idx] for f in fields)
            node = tuple(tr[f][nidx] for f in node_fields) if updates == 0 else None
            rets.append(agent.update_from_host(host, updates, node))
            if updates == 0:
                ws = agent._ws[B]
                first = {k: getattr(ws, k).clone() for k in ("act3", "logp3", "dq3", "dq_pi", "dheads2", "next_q", "next_l")}
        torch.cuda.synchronize()
        runs.append((agent, rets, first))
    (a0, r0, f0), (a1, r1, f1) = runs
    for k in f0:
        if k == "dheads2":      # (downstream of the rollout's backward, where the look-ahead map's multiply-adds are
            #                      contracted differently inside the interpolation launch: rounding, not bit-identical)
            assert float((f0[k] - f1[k]).abs().max()) <= 1e-5 * float(f1[k].abs().max()), k
        else:
            assert torch.equal(f0[k], f1[k]), "%s differs between the folded and the separate launches" % k
    np.testing.assert_allclose(np.array(r0), np.array(r1), rtol=2e-6, atol=1e-7)
    for i, (x, y) in enumerate(zip(a0.arenas, a1.arenas)):     # (Adam turns gradient rounding at |g| ~ 1e-8 into lr steps)
        params_close(x.theta.cpu().numpy(), y.theta.cpu().numpy(), 4 * 1e-3, "arena %d" % i)


@pytest.mark.parametrize("solver", ["euler", "rk4", "dopri5"])
@pytest.mark.parametrize("rows", [96, 77])
def test_fused_rk_step_kernel_matches_per_stage_launches(solver, rows):
    """nlbac_node_rk_fwd (one launch per RK step) against the un-fused stage-by-stage path: same rollout,
    same saved stage buffers, same action gradient (2 problems x rows, ragged last tile at rows=77)."""
    from nlbac_amd.odeint import AffineNodeSolver
    agent, env = make_agent(128, 256, 0, solver)
    node = agent.neural_ode_model
    g = torch.Generator().manual_seed(rows)
    y0 = torch.cat([torch.rand(2 * rows, 2, generator=g) * 4 - 2, torch.rand(2 * rows, 1, generator=g) * 6 - 3], 1).cuda()
    u = (torch.rand(2 * rows, 2, generator=g) * 2 - 1).cuda() * torch.tensor([3.5, 12.0]).cuda()
    dout = torch.randn(2 * rows, 3, generator=g).cuda()
    res = []
    for fused in (False, True):
        sol = AffineNodeSolver(node, "cuda")
        sol.fused = fused
        out = sol.forward(y0, u, 2, rows, solver, 0.02).clone()
        du, dy0 = sol.backward(dout, need_du=True, need_dy0=True)
        ws = sol.ctx["steps"][-1]["ws"]
        res.append((out, du.clone(), dy0.clone(), ws.K.clone(), ws.Y.clone(), ws.gout.clone(), ws.acts_f.clone()))
    for a, b, name in zip(res[0], res[1], ("out", "du", "dy0", "K", "Y", "g(x)", "acts_f")):
        vec_close(b.cpu().numpy(), a.cpu().numpy(), 2e-6, "fused vs staged: " + name)
    # bit-packed ReLU masks instead of saved activations (rollouts that need no weight gradients): same arithmetic
    sol = AffineNodeSolver(node, "cuda")
    sol.keep_acts = False
    out = sol.forward(y0, u, 2, rows, solver, 0.02).clone()
    du, dy0 = sol.backward(dout, need_du=True, need_dy0=True)
    assert sol.ctx["steps"][-1]["ws"].bits and sol.ctx["steps"][-1]["ws"].acts_f.dtype == torch.int32
    # (not bit for bit since round 4: in mask mode the g_net wave of a half tile takes over part of f_net's last layer and
    #  f_net's output layer is summed in two parts — node_rr_kernels.hip, SPLIT —, in activation mode the forward is not
    #  split: the same fp32 arithmetic in another summation order)
    for a, b, name in zip(res[1][:3], (out, du, dy0), ("out", "du", "dy0")):
        vec_close(b.cpu().numpy(), a.cpu().numpy(), 2e-6, "mask mode vs activation mode: " + name)


def test_hipgraphs_with_alternating_batch_sizes_stay_correct():
    """Captured graphs bake in the addresses of the solvers' buffers and belong to the solve they recorded.  Updates that
    alternate between three batch sizes (each with its own workspaces, graphs and dopri5 control blocks, and a NODE fit
    whose row count changes) must each reproduce the oracle — a replay must neither write into buffers another size
    has taken over nor read another size's accept decision."""
    from oracle import nlbac_oracle as O
    torch.set_num_threads(4)
    hidden, seed, env_name, solver = 64, 0, "Unicycle", "dopri5"
    agent, env = make_agent(128, hidden, seed, solver, env_name)
    agent.use_graphs = True
    oargs = O.Args(batch_size=128, hidden_size=hidden, seed=seed)
    oracle = O.make_oracle(synth.fixture_env(env_name, seed), oargs, synth.agent_weights(env_name, hidden, seed), solver=solver)
    tr = synth.transitions(env_name, 8192, seed=seed + 1, env=env)
    fields = synth.fields(env_name)
    sizes = [128, 256, 64, 128, 256, 64, 128, 256, 64, 256, 128]
    fit_rows = {0: 512, 10: 5000}
    for u, B in enumerate(sizes):
        rs = np.random.RandomState(100 + u)
        idx = rs.choice(8192, B, replace=False)
        batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in fields}
        eps = [torch.from_numpy(e) for e in synth.normal_eps(3, B, 2, seed=u)]
        node = None
        if u in fit_rows:
            nidx = rs.choice(8192, fit_rows[u], replace=False)
            node = tuple(torch.tensor(tr[f][nidx], dtype=torch.float32) for f in ("obs", "action", "next_obs"))
        R = oracle.update(batch, eps, u, node_batch=node)
        agent.set_noise(eps)
        ret = agent.update_from_host(tuple(batch[f].numpy() for f in fields), u,
                                     tuple(t.numpy() for t in node) if node else None)
        torch.cuda.synchronize()
        vec_close(ret, R["ret"], TOL, "update %d (B=%d): returned floats" % (u, B))
        vec_close(agent.node_solver.ctx["out"][:B].cpu().numpy(), R["x_next"].numpy(), TOL, "update %d: x_next" % u)
        vec_close(flat_grad(agent, agent.ar_a, agent.policy), R["g_policy"], TOL, "update %d: policy gradient" % u)
    assert sum(len(w.graphs) for w in agent._ws.values()) >= 6, "graphs were not captured for the three sizes"
    for name, mod, osd in (("critic", agent.critic, oracle.critic), ("policy", agent.policy, oracle.policy),
                           ("node", agent.neural_ode_model, oracle.node)):
        ov = torch.cat([osd[k].detach().reshape(-1) for k in osd])
        params_close(flat_params(mod), ov, 1e-3 * len(sizes), "params %s after %d updates" % (name, len(sizes)))
