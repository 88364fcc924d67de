"""GPU, row f3: the batched device simulators (``nlbac_amd.envs.device``, ``csrc/env_kernels.hip``) against the traces
recorded from the REFERENCE's own env classes (tests/golden/driver_<env>.npz, oracle/gen_driver_golden.py): the same
scripted actions, episode by episode, must reproduce every observation, reward, constraint, barrier signal, Lyapunov
input, done flag and safety counter — for every one of n_envs copies advanced by the same launches, and with copies
that are reset at different times staying independent."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_driver_golden import ScriptedAgent

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["Unicycle", "UnicycleBarrier", "Pvtol", "PvtolBarrier", "SimulatedCars"])
def test_device_env_reproduces_the_reference_trace(name):
    from nlbac_amd.envs import device as D
    g = np.load(os.path.join(GOLD, "driver_%s.npz" % name))
    N = 3
    env = D.make(name, N, 0)
    env.max_episode_steps = int(g["meta_max_steps"])
    agent = ScriptedAgent(name, env.action_space, pattern=int(g["meta_pattern"]))
    barrier = name.endswith("Barrier")
    n = len(g["reward"])
    backup = g["backup"]
    tol = dict(rtol=1e-9, atol=1e-11)
    cars = name == "SimulatedCars"

    def reset():
        # the reference draws one N(0, 0.5) velocity offset per reset from numpy's global generator (seeded in the
        # env's constructor, whose own reset takes the first draw): every copy gets the trace's draw
        env.reset(noise=[np.random.normal(0, 0.5)] * N) if cars else env.reset()
    if cars:
        np.random.seed(0)
        np.random.normal(0, 0.5)
    reset()
    for k in range(n):
        a = agent.select_action_backup(None) if backup[k] else agent.select_action(None)
        out = env.step(np.tile(a, (N, 1)))
        obs, reward, constraint = out[:3]
        lya, nlya, done, info = out[-4:]
        for e in range(N):      # every copy, advanced by the same launch
            np.testing.assert_allclose(obs[e].cpu().numpy(), g["obs"][k], **tol)
        np.testing.assert_allclose(reward[0].item(), g["reward"][k], **tol)
        np.testing.assert_allclose(constraint[0].item(), g["constraint"][k], **tol)
        if barrier:
            np.testing.assert_allclose(out[3][0].item(), g["extra"][k, 0], **tol)
        np.testing.assert_allclose(lya[1].cpu().numpy(), g["lya"][k], **tol)
        np.testing.assert_allclose(nlya[2].cpu().numpy(), g["next_lya"][k], **tol)
        assert bool(done[0].item()) == bool(g["done"][k])
        assert float(info["num_safety_violation"][0]) == g["n_violation"][k]
        np.testing.assert_allclose(float(info["safety_cost"][0]), g["safety_cost"][k], rtol=1e-8, atol=1e-11)
        first = "reached" if name == "SimulatedCars" else "goal_met"
        assert float(info[first][0]) == (g["reached"][k] if name == "SimulatedCars" else g["goal_met"][k])
        if g["done"][k]:
            reset()


def test_copies_reset_at_different_times_stay_independent():
    from nlbac_amd import envs as H
    from nlbac_amd.envs import device as D
    N = 4
    env = D.make("Unicycle", N, 0)
    hosts = [H.make("Unicycle", 0) for _ in range(N)]
    rs = np.random.RandomState(3)
    for k in range(120):
        a = rs.uniform(env.action_space.low, env.action_space.high, size=(N, 2))
        obs = env.step(a)[0].cpu().numpy()
        for e in range(N):
            np.testing.assert_allclose(obs[e], hosts[e].step(a[e])[0], rtol=1e-9, atol=1e-11)
        if k % 37 == 36:      # reset one copy only
            e = (k // 37) % N
            mask = np.zeros(N, dtype=bool)
            mask[e] = True
            env.reset(mask)
            hosts[e].reset()
