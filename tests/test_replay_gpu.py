"""GPU: the HBM-resident replay selects the same transitions as the reference-shaped host replay for
the same seed, and ``update_parameters`` driven from it lands on bit-identical parameters."""
import random

import numpy as np
import pytest
import torch

from nlbac_amd import synth
from test_agent_parity_gpu import make_agent

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_name", ["Unicycle", "UnicycleBarrier"])
def test_device_replay_matches_host_replay(env_name):
    if env_name.endswith("Barrier"):
        from nlbac_amd.neural_barrier_certificate.sac_cbf_clf.replay_memory import DeviceReplayMemory, ReplayMemory
    else:
        from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory, ReplayMemory
    B, n = 64, 700
    fields = synth.fields(env_name)
    tr = synth.transitions(env_name, n, seed=4)
    results = []
    for kind in ("host", "device"):
        agent, env = make_agent(B, 64, 0, "rk4", env_name)
        mem = ReplayMemory(1000, 7) if kind == "host" else DeviceReplayMemory(1000, 7, agent, chunk=256)
        for i in range(n):
            vals = [tr[f][i] for f in fields]
            mem.push(*vals[:-2], t=vals[-2], next_t=vals[-1])
        assert len(mem) == n and mem.position == n
        random.seed(11)
        first = mem.sample(5)
        rets = []
        for updates in range(3):
            agent.set_noise(synth.normal_eps(agent.task.n_eps, B, env.n_u, seed=updates))
            rets.append(agent.update_parameters(mem, B, updates, None, mem, 10))
        torch.cuda.synchronize()
        results.append((first, rets, agent.ar_c.theta.clone(), agent.ar_a.theta.clone(), agent.ar_n.theta.clone()))
    (f0, r0, *th0), (f1, r1, *th1) = results
    for a, b in zip(f0, f1):
        np.testing.assert_allclose(np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32), rtol=0, atol=0)
    np.testing.assert_array_equal(np.array(r0), np.array(r1))
    for a, b in zip(th0, th1):
        assert torch.equal(a, b)


def test_device_replay_ring_wraps_and_bulk_insert():
    from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
    agent, env = make_agent(8, 64, 0, "euler")
    mem = DeviceReplayMemory(100, 0, agent, chunk=16)
    tr = synth.transitions("Unicycle", 130, seed=2)
    for i in range(130):
        mem.push(*[tr[f][i] for f in synth.FIELDS])
    mem.flush()
    assert len(mem) == 100 and mem.position == 30
    rows = agent._rows_from_host(tuple(tr[f] for f in synth.FIELDS))
    torch.testing.assert_close(mem.rows[:30].cpu(), rows[100:130])
    torch.testing.assert_close(mem.rows[30:].cpu(), rows[30:100])
    mem2 = DeviceReplayMemory(100, 0, agent)
    mem2.push_rows(rows[:130][:90])
    mem2.push_rows(rows[90:130].to(agent.device))
    torch.testing.assert_close(mem2.rows.cpu(), mem.rows.cpu())
    out = mem.sample_rows(8)
    assert out.shape == (8, agent.lay.LD)


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """save_model / load_weights round trip (reference files) and the full-state checkpoint: resuming continues bit
    for bit."""
    B = 64
    tr = synth.transitions("Unicycle", 512, seed=9)

    def run(agent, env, updates):
        rs = np.random.RandomState(updates[0])
        rets = []
        for u in updates:
            idx = rs.choice(512, B, replace=False)
            agent.set_noise(synth.normal_eps(3, B, 2, seed=u))
            host = tuple(tr[f][idx] for f in synth.FIELDS)
            node = tuple(tr[f][:256] for f in ("obs", "action", "next_obs")) if u % 10 == 0 else None
            rets.append(agent.update_from_host(host, u, node))
        return rets
    a0, env = make_agent(B, 64, 0, "rk4")
    run(a0, env, [0, 1, 2])
    a0.save_checkpoint(str(tmp_path / "full.pt"))
    a0.save_model(str(tmp_path))
    tail0 = run(a0, env, [10, 11])
    a1, _ = make_agent(B, 64, 1, "euler")          # different seed / solver: everything must come from the file
    a1.load_checkpoint(str(tmp_path / "full.pt"))
    tail1 = run(a1, env, [10, 11])
    np.testing.assert_array_equal(np.array(tail0), np.array(tail1))
    for x, y in ((a0.ar_c, a1.ar_c), (a0.ar_a, a1.ar_a), (a0.ar_n, a1.ar_n)):
        assert torch.equal(x.theta, y.theta) and torch.equal(x.m, y.m)
    assert torch.equal(a0.ar_c.target, a1.ar_c.target) and torch.equal(a0.sc, a1.sc)
    # reference-format files: policy / critic / Lyapunov / NODE weights round-trip through the .pkl files
    a2, _ = make_agent(B, 64, 2, "rk4")
    a2.load_weights(str(tmp_path))
    obs = tr["obs"][:5]
    a3, _ = make_agent(B, 64, 3, "rk4")
    a3.load_checkpoint(str(tmp_path / "full.pt"))
    np.testing.assert_array_equal(a2.select_action(obs, evaluate=True), a3.select_action(obs, evaluate=True))


def test_device_drawn_minibatch_and_noise():
    """``device_rng`` mode: one launch draws the row indices (uniform, with replacement), gathers the rows and fills the
    update's N(0,1) noise.  Checked: every output row is a whole source row with an index in range, the draw is a
    deterministic function of (seed, draw number), indices and noise have the right first moments."""
    from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
    agent, env = make_agent(8, 64, 0, "euler")
    LD, n_src, B = agent.lay.LD, 3000, 4096
    src = torch.randn(n_src, LD)
    src[:, 0] = torch.arange(n_src, dtype=torch.float32)          # column 0 = the row's own index
    outs = []
    for seed in (5, 5, 6):
        mem = DeviceReplayMemory(n_src + 10, seed, agent, device_rng=True)      # (capacity > rows held: len matters)
        mem.push_rows(src)
        eps = torch.full((3, B, 2), float("nan"), device="cuda")
        a = mem.sample_rows(B, eps_out=eps).cpu()
        b = mem.sample_rows(B).cpu()
        outs.append((a, b, eps.cpu()))
        idx = a[:, 0].long()
        assert int(idx.min()) >= 0 and int(idx.max()) < n_src
        assert torch.equal(a, src[idx])                           # whole rows, bit-exact
        assert not torch.equal(a[:, 0], b[:, 0])                  # the next draw differs
    (a0, b0, e0), (a1, b1, e1), (a2, b2, e2) = outs
    assert torch.equal(a0, a1) and torch.equal(b0, b1) and torch.equal(e0, e1)      # same seed: same stream
    assert not torch.equal(a0[:, 0], a2[:, 0]) and not torch.equal(e0, e2)          # another seed: another stream
    # first moments: indices uniform on [0, n_src) (mean n/2, sd n/sqrt(12), 4096 draws), noise standard normal
    idx = a0[:, 0].double()
    assert abs(float(idx.mean()) - (n_src - 1) / 2) < 5 * n_src / np.sqrt(12 * B)
    assert len(torch.unique(idx)) > 0.6 * min(n_src, B)
    e = e0.double().flatten()
    assert torch.isfinite(e).all() and len(torch.unique(e)) > 0.999 * e.numel()
    assert abs(float(e.mean())) < 5 / np.sqrt(e.numel()) and abs(float(e.std()) - 1) < 0.03
    assert abs(float((e ** 3).mean())) < 0.1 and abs(float((e ** 4).mean()) - 3) < 0.25


def test_lagged_loss_readback_returns_the_previous_update():
    """``update_on_device(..., sync="lagged")`` hands back the 6 floats of the previous call (None first) and leaves
    the updates themselves untouched."""
    from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory
    B = 64
    got = []
    for mode in (True, "lagged"):
        agent, env = make_agent(B, 64, 0, "dopri5")
        mem = DeviceReplayMemory(512, 3, agent, device_rng=True)
        mem.push_rows(agent._rows_from_host(tuple(synth.transitions("Unicycle", 512, seed=9)[f] for f in synth.FIELDS)))
        ws = agent._workspace(B)
        rets = []
        for i in range(4):
            mem.sample_rows(B, out=ws.mb, eps_out=ws.eps)
            rets.append(agent.update_on_device(ws, i, sync=mode, eps_ready=True))
        torch.cuda.synchronize()
        got.append((rets, agent.ar_c.theta.clone(), agent.ar_a.theta.clone()))
    (r_sync, c0, a0), (r_lag, c1, a1) = got
    assert r_lag[0] is None
    assert r_lag[1:] == r_sync[:3]
    assert torch.equal(c0, c1) and torch.equal(a0, a1)


def test_select_action_latency_path_matches_the_policy_formula():
    """``select_action`` on one observation (pinned staging, persistent buffers) against the oracle's policy head:
    the deterministic action is tanh(mean)·scale + bias, the stochastic one is the squashed Gaussian at the draw the
    call made; a batch of observations takes the tensor path and agrees with the single-observation path."""
    from oracle import nlbac_oracle as O
    agent, env = make_agent(8, 256, 0, "euler")
    sd = {k: v.detach().cpu().clone() for k, v in agent.policy.state_dict().items()}
    scale, bias = agent.policy.action_scale.cpu(), agent.policy.action_bias.cpu()
    tr = synth.transitions("Unicycle", 16, seed=5, env=env)
    for i in range(4):
        obs = tr["obs"][i]
        o32 = torch.tensor(obs, dtype=torch.float32)[None]
        a_det = agent.select_action(obs, evaluate=True)
        ref_det = O.policy_sample(sd, o32, torch.zeros(1, 2), scale, bias)[2][0].numpy()
        np.testing.assert_allclose(a_det, ref_det, rtol=1e-5, atol=1e-6)
        a = agent.select_action(obs)
        eps = agent.policy._act_ws.eps.cpu()
        ref = O.policy_sample(sd, o32, eps, scale, bias)[0][0].numpy()
        np.testing.assert_allclose(a, ref, rtol=1e-5, atol=1e-6)
        assert a.shape == (2,) and a.dtype == np.float32
    batch = agent.select_action(tr["obs"][:4], evaluate=True)
    singles = np.stack([agent.select_action(tr["obs"][i], evaluate=True) for i in range(4)])
    np.testing.assert_allclose(batch, singles, rtol=1e-6, atol=1e-7)
    with pytest.raises(AttributeError):
        make_agent(8, 64, 0, "euler", "UnicycleBarrier")[0].select_action_backup(tr["obs"][0])
