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
