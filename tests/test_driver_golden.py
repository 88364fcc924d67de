"""Rows f2 / f3 against the REFERENCE's own code: ``tests/golden/driver_<env>.npz`` hold what the reference's
``main.py::train`` loop did on the reference's simulators when driven by a scripted stand-in agent
(``oracle/gen_driver_golden.py``, run in the build container) — every env step's outputs, which controller acted,
which transitions were kept out of the controller replay, the time stamps both replays received.  The same script
through ``nlbac_amd.train.train`` on ``nlbac_amd.envs`` must reproduce the trace: the simulators to 1e-12, the
driver's decisions exactly.  (SimulatedCars: the reference's hand-over condition — 4th car within 2.5 of the 5th while
its distance to the 3rd is in range — is not reachable with the 5th car's own braking rule, so that trace pins the
simulator, the time stamps and the absence of hand-overs; the hand-over itself is pinned by
``driver_SimulatedCarsHandover.npz``: the reference's loop on a scripted stand-in ENVIRONMENT whose gaps make every
branch of C/main.py:102-112 fire.)"""
import os
import types

import numpy as np
import pytest

from nlbac_amd import envs, train
from oracle.gen_driver_golden import Recorder, ScriptedAgent, ScriptedCarsEnv

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["Unicycle", "SimulatedCars", "Pvtol", "UnicycleBarrier", "PvtolBarrier"])
def test_driver_and_simulator_reproduce_the_reference_trace(name):
    g = np.load(os.path.join(GOLD, "driver_%s.npz" % name))
    env = envs.make(name, 0)
    env.max_episode_steps = int(g["meta_max_steps"])
    agent = ScriptedAgent(name, env.action_space, pattern=int(g["meta_pattern"]))
    args = types.SimpleNamespace(env=name, batch_size=int(g["meta_batch_size"]), updates_per_step=1,
                                 start_steps=int(g["meta_start_steps"]), max_episodes=int(g["meta_episodes"]),
                                 NODE_model_update_interval=10, output=None, max_steps=0)
    steps = []
    orig = env.step

    def step(a):
        out = orig(a)
        steps.append(out)
        return out
    env.step = step
    mem, node, trace = Recorder(), Recorder(), []
    train.train(agent, env, None, args, mem, node, log=lambda *a: None, trace=trace)
    n = len(g["reward"])
    assert len(steps) == n, "episode lengths differ: %d steps vs the reference's %d" % (len(steps), n)
    # ---- f3: the simulator
    tol = dict(rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.stack([s[0] for s in steps]), g["obs"], **tol)
    np.testing.assert_allclose([s[1] for s in steps], g["reward"], **tol)
    np.testing.assert_allclose([s[2] for s in steps], g["constraint"], **tol)
    if name.endswith("Barrier"):
        np.testing.assert_allclose([s[3] for s in steps], g["extra"][:, 0], **tol)
    np.testing.assert_allclose(np.stack([np.asarray(s[-4], dtype=np.float64) for s in steps]), g["lya"], **tol)
    np.testing.assert_allclose(np.stack([np.asarray(s[-3], dtype=np.float64) for s in steps]), g["next_lya"], **tol)
    np.testing.assert_array_equal([bool(s[-2]) for s in steps], g["done"])
    viol = [float(sum(v for k, v in s[-1].items() if k.startswith("num_safety_violation"))) for s in steps]
    np.testing.assert_array_equal(viol, g["n_violation"])
    cost = [float(sum(v for k, v in s[-1].items() if k.startswith("safety_cost"))) for s in steps]
    np.testing.assert_allclose(cost, g["safety_cost"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal([float(bool(s[-1].get("goal_met", False))) for s in steps], g["goal_met"])
    # ---- f2: the driver
    np.testing.assert_array_equal(agent.calls, g["backup"])
    np.testing.assert_array_equal([int(p) for _, p in trace], g["pushed"])
    np.testing.assert_array_equal([int(b) for b, _ in trace], g["backup"])
    assert agent.updates == int(g["updates"])
    np.testing.assert_allclose([r[1] for r in mem.rows], g["mem_t"], rtol=0, atol=1e-12)
    np.testing.assert_allclose([r[2] for r in mem.rows], g["mem_next_t"], rtol=0, atol=1e-12)
    np.testing.assert_allclose([r[1] for r in node.rows], g["node_t"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal([float(r[0][-1]) for r in node.rows], g["mask"])
    if name in ("Unicycle", "Pvtol"):
        assert g["backup"].sum() > 0 and (g["pushed"] == 0).sum() == g["backup"].sum()


def test_cars_handover_reproduces_the_reference_loop():
    """``_CarsHandover`` against the reference's own ``main.py::train`` (C/main.py:41-112) on ``ScriptedCarsEnv``: which
    controller acted at every step (hand-overs ended by the 15-step cap, by both gaps re-opening after 5 steps, by the
    end of the episode; a close approach without ``reached`` that must NOT hand over), which transitions were kept out
    of the controller replay, the time stamps of both replays, the number of updates."""
    g = np.load(os.path.join(GOLD, "driver_SimulatedCarsHandover.npz"))
    env = ScriptedCarsEnv(int(g["meta_max_steps"]))
    agent = ScriptedAgent("SimulatedCars", env.action_space, pattern=int(g["meta_pattern"]))
    args = types.SimpleNamespace(env="SimulatedCars", batch_size=int(g["meta_batch_size"]), updates_per_step=1,
                                 start_steps=int(g["meta_start_steps"]), max_episodes=int(g["meta_episodes"]),
                                 NODE_model_update_interval=10, output=None, max_steps=0)
    mem, node, trace = Recorder(), Recorder(), []
    train.train(agent, env, None, args, mem, node, log=lambda *a: None, trace=trace)
    assert g["backup"].sum() > 100 and len(trace) == len(g["backup"])
    np.testing.assert_array_equal(agent.calls, g["backup"])
    np.testing.assert_array_equal([int(b) for b, _ in trace], g["backup"])
    np.testing.assert_array_equal([int(p) for _, p in trace], g["pushed"])
    assert (g["pushed"] == 0).sum() == g["backup"].sum()
    # every kind of ending occurs in the trace: runs of 15 backup steps (the cap) and shorter ones (gaps re-opened / episode end)
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], g["backup"], [0]]))).reshape(-1, 2), axis=1).reshape(-1)
    assert (runs == 15).any() and (runs == 5).any(), runs
    assert agent.updates == int(g["updates"])
    np.testing.assert_allclose([r[1] for r in mem.rows], g["mem_t"], rtol=0, atol=1e-12)
    np.testing.assert_allclose([r[2] for r in mem.rows], g["mem_next_t"], rtol=0, atol=1e-12)
    np.testing.assert_allclose([r[1] for r in node.rows], g["node_t"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal([float(r[0][-1]) for r in node.rows], g["mask"])
