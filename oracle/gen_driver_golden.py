"""Driver / simulator golden traces — runs ONLY in the build container (needs ``/root/reference``).  Test
infrastructure, not product code.

Imports the reference's OWN training loop (``<copy>/main.py::train``) and simulators (``<copy>/envs/*.py``) and runs
them with a scripted stand-in agent (deterministic actions from seeded streams, a no-op ``update_parameters``), then
records what the loop did: every env step's observation, reward, constraint, Lyapunov inputs, done flag and safety
counters (rows f3), which controller acted and whether the transition reached the controller replay, and the time
stamps both replays received (row f2).  ``tests/test_driver_golden.py`` replays the same script through
``nlbac_amd.train.train`` on ``nlbac_amd.envs`` and must reproduce the trace.

What has to be faked to import the reference's ``main.py`` / ``envs`` here (none of it is on the traced path):
  * ``gym`` (not installed): a module with ``Env`` and ``spaces.Box`` (shape / low / high / sample / seed);
  * ``wandb``, ``tensorflow``, ``mpi4py``, ``joblib``: empty modules; ``utils.logx.EpochLogger`` -> a no-op class
    (spinup logging, out of scope); ``main.writer`` (a module-level wandb run in the reference) -> a no-op;
  * ``torchdiffeq`` -> this repo's oracle ``odeint`` (imported by the agent module, never called here).

Usage: python oracle/gen_driver_golden.py --env Unicycle|SimulatedCars|Pvtol|UnicycleBarrier|PvtolBarrier
(one env per process: the copies share module names).  Output: tests/golden/driver_<env>.npz (numbers only).
"""
import argparse
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle.gen_golden import REFS  # noqa: E402

# episodes / steps per episode of the trace, and the scripted agent's behaviour per env: the primary controller's
# actions are chosen so that every hand-over rule fires (standing still -> "stuck", braking into the car behind, ...)
SCRIPT = {
    "SimulatedCarsHandover": dict(episodes=6, max_steps=90, start_steps=20),      # ScriptedCarsEnv below, not the simulator
    "Unicycle": dict(episodes=7, max_steps=160, start_steps=40),
    "UnicycleBarrier": dict(episodes=3, max_steps=120, start_steps=40),
    "SimulatedCars": dict(episodes=6, max_steps=300, start_steps=30),
    "Pvtol": dict(episodes=6, max_steps=200, start_steps=40),
    "PvtolBarrier": dict(episodes=3, max_steps=120, start_steps=40),
}


PATTERN = int(os.environ.get("NLBAC_DRIVER_PATTERN", "0"))


class ScriptedAgent:
    """Same calls as SAC_CBF_CLF from the driver's point of view.  Primary: mostly (near-)idle actions so that the
    vehicle gets stuck / drifts, with bursts; backup: a steady push.  Everything comes from two seeded streams, so the
    device build's driver can be fed the identical script."""

    def __init__(self, env_name, action_space, seed=0, pattern=0):
        self.env_name, self.pattern = env_name, pattern
        self.lo, self.hi = np.asarray(action_space.low, dtype=np.float64), np.asarray(action_space.high, dtype=np.float64)
        self.rs = np.random.RandomState(seed)
        self.rs_b = np.random.RandomState(seed + 1)
        self.calls = []               # per env step: 0 primary, 1 backup
        self.updates = 0
        self.backup_policy = object() if not env_name.endswith("Barrier") else None
        self.k = 0

    def _act(self, rs, backup):
        self.k += 1
        u = rs.uniform(0.0, 1.0, size=self.lo.shape)
        if self.env_name.startswith("Unicycle"):
            a = np.array([0.02 * u[0], (u[1] - 0.5) * 0.4]) if not backup else np.array([2.5 + u[0], (u[1] - 0.5)])
            if not backup and (self.k // 90) % 3 == 2:
                a = np.array([3.0, 0.5 * (u[1] - 0.5)])
        elif self.env_name == "SimulatedCars":
            # piecewise-constant acceleration levels drawn per 25-step block (the hand-over needs the 4th car to drop
            # back onto the 5th while its distance to the 3rd is in range: found by trying pattern seeds)
            lvl = np.random.RandomState(1000 + self.pattern + self.k // 25).uniform(-3.0, 3.0)
            a = np.array([lvl + 0.1 * (u[0] - 0.5)])
            if backup:
                a = np.array([2.5 + 0.5 * u[0]])
        else:
            a = np.array([(u[0] - 0.5) * 0.2, (u[1] - 0.5) * 0.2]) if not backup else np.array([1.0 + u[0], (u[1] - 0.5)])
            if not backup and (self.k // 70) % 3 == 1:
                a = np.array([2.0, -1.0 + 2 * u[1]])
        return np.clip(a, self.lo, self.hi)

    def select_action(self, obs, evaluate=False, warmup=False):
        self.calls.append(0)
        return self._act(self.rs, False)

    def select_action_backup(self, obs, evaluate=False, warmup=False):
        self.calls.append(1)
        return self._act(self.rs_b, True)

    def update_parameters(self, *a, **k):
        self.updates += 1
        return (0.0,) * 6

    def save_model(self, output):
        pass


class ScriptedCarsEnv:
    """A stand-in for the SimulatedCars ENVIRONMENT whose observations follow a script instead of the car dynamics: the
    reference simulator's 5th car keeps its own distance, so the hand-over condition of C/main.py:102-112 — 4th car
    within 2.5 of the 5th WHILE the following distance to the 3rd is met (``info['reached']``) — never fires on it
    (tests/golden/driver_SimulatedCars.npz: 0 backup steps).  Here the gaps are scripted so that every branch fires: a
    hand-over ended by the 15-step cap, one ended after 5 steps by both gaps re-opening, a close approach WITHOUT
    ``reached`` (no hand-over), and one that is cut off by the end of the episode.  Same 7-tuple as the reference's
    ``SimulatedCarsEnv.step`` (next_obs, reward, constraint, cur_pos_vel_info, next_pos_vel_info, done, info)."""
    dt = 0.02

    def __init__(self, max_episode_steps=90):
        self.max_episode_steps = max_episode_steps
        lo, hi = np.array([-10.0]), np.array([10.0])
        self.action_space = type("Box", (), dict(low=lo, high=hi, shape=(1,)))()
        self.k = 0
        self.episode = -1

    @staticmethod
    def gaps(k):
        """(d34, d45, reached) after step k of an episode, in the units the driver compares with 2.5"""
        if k == 10: return 9.5, 2.0, 1            # -> hand-over ...
        if 10 < k < 40: return 2.0, 2.0, 1        # ... gaps never re-open: ended by the 15-step cap (and taken again)
        if k == 45: return 9.5, 2.0, 0            # close, but the following distance is not met: no hand-over
        if k == 50: return 9.5, 1.0, 1            # -> hand-over ...
        if 50 < k < 53: return 2.0, 2.0, 1
        if 53 <= k < 70: return 5.0, 5.0, 1       # ... both gaps open again: ended once 5 backup steps have passed
        if k >= 85: return 9.5, 2.0, 1            # -> hand-over cut off by the end of the episode
        return 9.5, 6.0, 0

    def _obs(self, k):
        d34, d45, _ = self.gaps(k)
        p3 = 0.6 + 0.001 * k + 0.01 * self.episode
        p4, p5 = p3 - d34 / 100.0, p3 - d34 / 100.0 - d45 / 100.0
        return np.array([p3 + 0.2, 0.3, p3 + 0.1, 0.3, p3, 0.3 + 0.001 * k, p4, 0.29, p5, 0.28])

    def reset(self):
        self.k = 0
        self.episode += 1
        return self._obs(0)

    def step(self, action):
        cur = self._obs(self.k)
        self.k += 1
        nxt = self._obs(self.k)
        _, _, reached = self.gaps(self.k)
        done = self.k >= self.max_episode_steps
        info = {"reached": reached, "num_safety_violation": 0, "safety_cost": 0.0}
        return (nxt, -0.01 * self.k + float(action[0]) * 1e-3, 0.0, cur[4:8].copy(), nxt[4:8].copy(), done, info)


class Recorder:
    """Replay stand-in that records what the driver pushes (and counts like ``len(ReplayMemory)``)."""

    def __init__(self, *a, **k):
        self.rows = []
        self.position = 0

    def push(self, *row, t=None, next_t=None):
        self.rows.append((row, t, next_t))
        self.position += 1

    def __len__(self):
        return len(self.rows)


def stub_modules():
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            low, high = np.asarray(low, dtype=dtype), np.asarray(high, dtype=dtype)
            if shape is not None and low.shape != tuple(shape):
                low, high = np.full(shape, low, dtype=dtype), np.full(shape, high, dtype=dtype)
            self.low, self.high, self.shape, self.dtype = low, high, low.shape, dtype
            self._rng = np.random.RandomState()

        def seed(self, s=None):
            self._rng = np.random.RandomState(s)
            return [s]

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    gym.Env, gym.spaces = Env, spaces
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    for name in ("wandb", "tensorflow", "mpi4py", "joblib"):
        sys.modules[name] = types.ModuleType(name)
    import torch  # noqa: F401
    from oracle import nlbac_oracle as O
    td = types.ModuleType("torchdiffeq")
    td.odeint = O.odeint
    sys.modules["torchdiffeq"] = td


def run(env_name):
    scripted_env = env_name == "SimulatedCarsHandover"
    out_name = env_name
    if scripted_env:
        env_name = "SimulatedCars"
    ref = REFS[env_name]
    stub_modules()
    sys.path.insert(0, ref)
    logx = types.ModuleType("utils.logx")

    class EpochLogger(object):
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    logx.EpochLogger = EpochLogger
    import utils  # noqa: F401  (the reference's package; logx inside it is replaced before main imports it)
    sys.modules["utils.logx"] = logx
    import main as M
    import sac_cbf_clf.model as Mo
    import torch
    Mo.device = torch.device("cpu")

    class W(object):
        def log(self, *a, **k):
            pass
    M.writer = W()
    M.prYellow = M.prGreen = lambda *a, **k: None
    cfg = SCRIPT[out_name]
    recs = {"memory": None, "node": None}

    def make_memory(*a, **k):
        r = Recorder()
        recs["memory" if recs["memory"] is None else "node"] = r
        return r
    M.ReplayMemory = make_memory
    args = types.SimpleNamespace(env=env_name.replace("Barrier", ""), env_name=env_name.replace("Barrier", ""), seed=0, replay_size=1000, batch_size=16,
                                 updates_per_step=1, start_steps=cfg["start_steps"], max_episodes=cfg["episodes"],
                                 NODE_model_update_interval=10, output="/tmp", cuda=False)
    env = ScriptedCarsEnv(cfg["max_steps"]) if scripted_env else M.build_env(args)
    env.max_episode_steps = cfg["max_steps"]
    agent = ScriptedAgent(env_name, env.action_space, pattern=PATTERN)
    steps = []
    orig_step = env.step

    def step(action):
        out = orig_step(action)
        info = out[-1]
        steps.append((np.asarray(out[0], dtype=np.float64).copy(), float(out[1]), float(out[2]),
                      [float(x) for x in out[3:-4]], np.asarray(out[-4], dtype=np.float64).copy(),
                      np.asarray(out[-3], dtype=np.float64).copy(), bool(out[-2]),
                      float(sum(v for k, v in info.items() if k.startswith("num_safety_violation"))),
                      float(sum(v for k, v in info.items() if k.startswith("safety_cost"))),
                      float(bool(info.get("goal_met", False))), float(info.get("reached", 0))))
        return out
    env.step = step

    class Dyn(object):
        def get_state(self, obs):
            return obs
    M.train(agent, env, Dyn(), args)
    mem, node = recs["memory"], recs["node"]
    n = len(steps)
    assert len(agent.calls) == n == len(node.rows)
    pushed = np.zeros(n, dtype=np.int64)
    # the controller replay holds a subset of the NODE replay's rows, in order: match them up by identity of next_obs
    j = 0
    for i, (row, t, nt) in enumerate(node.rows):
        if j < len(mem.rows) and mem.rows[j][0][-2] is row[-2]:
            pushed[i] = 1
            j += 1
    assert j == len(mem.rows)
    out = dict(meta_pattern=PATTERN, meta_env=out_name, meta_episodes=cfg["episodes"], meta_max_steps=cfg["max_steps"],
               meta_start_steps=cfg["start_steps"], meta_batch_size=args.batch_size,
               obs=np.stack([s[0] for s in steps]), reward=np.array([s[1] for s in steps]),
               constraint=np.array([s[2] for s in steps]), extra=np.array([s[3] for s in steps], dtype=np.float64),
               lya=np.stack([s[4] for s in steps]), next_lya=np.stack([s[5] for s in steps]),
               done=np.array([s[6] for s in steps]), n_violation=np.array([s[7] for s in steps]),
               safety_cost=np.array([s[8] for s in steps]), goal_met=np.array([s[9] for s in steps]),
               reached=np.array([s[10] for s in steps]),
               backup=np.array(agent.calls, dtype=np.int64), pushed=pushed,
               mem_t=np.array([r[1] for r in mem.rows]), mem_next_t=np.array([r[2] for r in mem.rows]),
               node_t=np.array([r[1] for r in node.rows]), mask=np.array([float(r[0][-1]) for r in node.rows]),
               updates=agent.updates)
    path = os.path.join(ROOT, "tests", "golden", "driver_%s.npz" % out_name)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", n, "steps;", int(out["backup"].sum()), "backup steps;",
          int(n - pushed.sum()), "kept out of memory; updates", agent.updates)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", required=True, choices=sorted(SCRIPT))
    run(ap.parse_args().env)
