"""The gym-free simulators reproduce the synthetic-transition formulas (which follow the reference env files), and —
on a GPU — the reference-shaped training driver runs episodes end to end on them."""
import numpy as np
import pytest

from nlbac_amd import envs, synth
from nlbac_amd.envspec import make_env


@pytest.mark.parametrize("name", ["Unicycle", "SimulatedCars", "Pvtol", "UnicycleBarrier", "PvtolBarrier"])
def test_env_step_matches_synthetic_transition_formulas(name):
    spec = make_env(name, 0)
    tr = synth.transitions(name, 40, seed=7, env=spec)
    env = envs.make(name, 0)
    barrier = name.endswith("Barrier")
    for i in range(40):
        obs, act = tr["obs"][i], tr["action"][i]
        if name.startswith("Unicycle"):
            env.state = np.array([obs[0], obs[1], np.arctan2(obs[3], obs[2])])
            env.last_goal_dist = np.linalg.norm(env.goal_pos - tr["center"][i])
        elif name == "SimulatedCars":
            env.state = obs.copy()
            env.state[::2] *= 100.0
            env.state[1::2] *= 30.0
            env.t = tr["t"][i]
        else:
            env.state = np.array([obs[0], obs[1], np.arctan2(obs[3], obs[2]), obs[4], obs[5], obs[6], obs[7]])
        env.episode_step = 0
        out = env.step(act)
        np.testing.assert_allclose(out[0], tr["next_obs"][i], rtol=1e-9, atol=1e-12)
        bonus = env.reward_goal if out[-1].get("goal_met") else 0.0      # (the synthetic rows carry no goal bonus)
        np.testing.assert_allclose(out[1] - bonus, tr["reward"][i], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(out[2], tr["constraint"][i], rtol=1e-9, atol=1e-12)
        if barrier:
            np.testing.assert_allclose(out[3], tr["barrier_signal"][i], atol=1e-12)
        np.testing.assert_allclose(out[-4], tr["center"][i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(out[-3], tr["next_center"][i], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name,device_replay", [("Unicycle", False), ("Unicycle", True), ("UnicycleBarrier", True),
                                                ("SimulatedCars", False), ("Pvtol", True)])
def test_training_driver_runs_end_to_end(name, device_replay):
    from nlbac_amd import train
    argv = ["--env", name, "--cuda", "--batch_size", "64", "--start_steps", "100", "--max_episodes", "2",
            "--max_steps", "260", "--seed", "0", "--updates_per_step", "1", "--solver", "rk4"]
    if device_replay:
        argv.append("--device_replay")
    hist = train.main(argv)
    assert hist and hist[-1]["total_steps"] >= 260 and hist[-1]["updates"] >= 150
    assert all(np.isfinite(h["reward"]) for h in hist)


@pytest.mark.gpu
def test_training_driver_is_reproducible_and_graph_replay_changes_nothing():
    """Same seed, same run: the driver seeds every generator before the agent builds its networks (U/main.py:253-258),
    the update has no order-dependent reductions, so two runs agree to the last bit — and a third run that replays
    each update as hipGraphs does too."""
    import torch
    from nlbac_amd import train
    argv = ["--env", "Unicycle", "--cuda", "--batch_size", "64", "--start_steps", "120", "--max_episodes", "2",
            "--max_steps", "320", "--seed", "3", "--updates_per_step", "2", "--solver", "dopri5", "--device_replay"]
    runs = []
    for extra in ([], [], ["--hipgraphs"]):
        agents = []
        orig = train.train

        def spy(agent, *a, **k):
            agents.append(agent)
            return orig(agent, *a, **k)
        train.train = spy
        try:
            hist = train.main(argv + extra)
        finally:
            train.train = orig
        torch.cuda.synchronize()
        runs.append(([h["reward"] for h in hist], [a.theta.clone() for a in agents[0].arenas]))
    for rewards, thetas in runs[1:]:
        assert rewards == runs[0][0]
        assert all(torch.equal(x, y) for x, y in zip(thetas, runs[0][1]))
