"""GPU: the vectorised driver (``nlbac_amd.train.train_vectorized``): N device environments, policy,
replay and updates without a host copy of any transition.  The simulators themselves are pinned lane by lane against
the reference's own traces in test_device_envs_gpu.py; here: the rows that reach the replay are the transitions the
environments produced, in the agent's minibatch layout, with the driver's mask / time conventions (U/main.py:146-156),
and a run is reproducible bit for bit."""
import types

import numpy as np
import pytest
import torch

from nlbac_amd.envs import device as denv
from nlbac_amd.train import train_vectorized
from test_agent_parity_gpu import make_agent

pytestmark = pytest.mark.gpu


def run(env_name, n_envs, iters, seed=0):
    agent, spec = make_agent(64, 64, seed, "euler", env_name, {"Unicycle": 50.0, "Pvtol": 0.8}[env_name])
    env = denv.make(env_name, n_envs, seed=seed)
    env.max_episode_steps = 25                      # (short episodes: resets and time-limit masks occur in the run)
    args = types.SimpleNamespace(replay_size=4096, seed=seed, start_steps=4 * n_envs, batch_size=64, updates_per_step=1,
                                 NODE_model_update_interval=10)
    seen = []
    res = train_vectorized(agent, env, args, iters * n_envs, log=lambda *a: None,
                           check=lambda it, rows: seen.append(rows.clone()) if it < 40 else None)
    torch.cuda.synchronize()
    return agent, env, res, seen


@pytest.mark.parametrize("env_name", ["Unicycle", "Pvtol"])
def test_rows_are_the_transitions_and_runs_repeat(env_name):
    N, iters = 16, 60
    a0, env, r0, seen = run(env_name, N, iters)
    lay = a0.lay
    assert r0["steps"] == N * iters and r0["updates"] == iters - (64 // N) and r0["episodes"] >= N
    dt = float(env.dt)
    n_cont = 0
    for k in range(len(seen) - 1):
        cur, nxt = seen[k], seen[k + 1]
        mask, t, nt = cur[:, lay.mask], cur[:, lay.t], cur[:, lay.nt]
        assert bool(((mask == 0) | (mask == 1)).all())
        np.testing.assert_allclose((nt - t).cpu().numpy(), dt, rtol=1e-5)
        step_idx = torch.round(t / dt)
        ended = step_idx >= 25                                     # time limit: the episode is over but the mask stays 1
        assert bool((mask[ended] == 1).all())
        cont = ~ended & (mask == 1)                                # lanes that go on: next row starts where this one ended
        if bool(cont.any()):      # (at the time limit every lane of this lock-step run ends at once)
            n_cont += int(cont.sum())
            assert torch.equal(cur[cont][:, lay.nobs:lay.nobs + lay.obs_dim], nxt[cont][:, lay.obs:lay.obs + lay.obs_dim])
            np.testing.assert_allclose(torch.round(nxt[cont][:, lay.t] / dt).cpu().numpy(), (step_idx[cont] + 1).cpu().numpy())
        fresh = ~cont                                              # lanes that were reset: their next row is step 1
        if bool(fresh.any()):
            np.testing.assert_allclose(torch.round(nxt[fresh][:, lay.t] / dt).cpu().numpy(), 1.0)
        lo, hi = env.action_space.low, env.action_space.high
        act = cur[:, lay.act:lay.act + lay.act_dim].cpu().numpy()
        assert (act >= lo - 1e-5).all() and (act <= hi + 1e-5).all()
    assert n_cont > 20 * N
    a1, _, r1, _ = run(env_name, N, iters)
    assert r0 == r1
    for x, y in zip(a0.arenas, a1.arenas):
        assert torch.equal(x.theta, y.theta), "two vectorised runs from the same seed differ"


def test_pvtol_fit_stops_after_100_episodes_per_lane_and_a_wrapped_replay_still_fits():
    """The driver hands ``update_parameters`` the reference's trailing ``i_episode`` (P/main.py; the Pvtol copy fits its
    NODE up to episode 100, P/sac_cbf_clf.py:205) as finished episodes per lane; and a device replay whose ring has just
    wrapped (``position`` back at 0) fits on its filled size instead of on zero rows."""
    N = 16
    agent, spec = make_agent(64, 64, 0, "euler", "Pvtol", 0.8)
    env = denv.make("Pvtol", N, seed=0)
    env.max_episode_steps = 2                       # 2-step episodes: 100 episodes per lane pass in 200 vector steps
    args = types.SimpleNamespace(replay_size=1040, seed=0, start_steps=1 << 30, batch_size=64, updates_per_step=1,
                                 NODE_model_update_interval=10)
    seen = []
    orig = agent.fit_node_rows
    agent.fit_node_rows = lambda rows: (seen.append((len(seen), rows.shape[0])), orig(rows))[1]
    episodes = []
    upd = agent.update_parameters
    agent.update_parameters = lambda *a: (episodes.append(a[-1]), upd(*a))[1]
    # 1040-row ring, 16 rows per step: position is back at 0 after step 65 — whose update is number 60, a fit update
    res = train_vectorized(agent, env, args, 260 * N, log=lambda *a: None)
    assert res["episodes"] >= 125 * N
    # the episode counter is read back (a host sync) only for the updates whose number makes a NODE fit possible; the
    # others pass None (the agent looks at it on fit updates only)
    assert all((e is not None) == (k % 10 == 0) for k, e in enumerate(episodes))
    n_updates = len(episodes)
    episodes = [e for e in episodes if e is not None]      # (one per fit-capable update: every 10th)
    assert episodes[0] <= 4 and episodes == sorted(episodes) and episodes[-1] > 100       # (the first update comes at step 5)
    n_fit_updates = sum(1 for e in episodes if e <= 100)
    assert len(seen) == n_fit_updates and 0 < n_fit_updates < (n_updates + 9) // 10
    assert all(n > 0 for _, n in seen)
    assert any(n == 1040 for _, n in seen), "no fit ran on a just-wrapped ring (position 0): pick other sizes"
