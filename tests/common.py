"""Shared helpers: rebuild the seeded inputs a golden fixture was made from."""
import os

import numpy as np
import torch

import nlbac_amd  # noqa: F401  (registers the package alias)
from nlbac_amd import synth
from nlbac_amd.envspec import make_env

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_FIELDS = ("obs", "action", "reward", "constraint", "center", "next_center", "next_obs", "mask", "t", "next_t")
PREFIX = {"QuadrotorLike": "quadrotor_like", "Unicycle": "unicycle", "SimulatedCars": "cars", "UnicycleBarrier": "nbc_unicycle", "Pvtol": "pvtol",
          "PvtolBarrier": "nbc_pvtol"}
N_EPS = {"QuadrotorLike": 3, "Unicycle": 3, "SimulatedCars": 5, "UnicycleBarrier": 3, "Pvtol": 7, "PvtolBarrier": 3}
NODE_FIELDS = {"QuadrotorLike": ("obs", "action", "next_obs"), "Unicycle": ("obs", "action", "next_obs"), "SimulatedCars": ("obs", "action", "next_obs", "t"),
               "UnicycleBarrier": ("obs", "action", "next_obs"), "Pvtol": ("obs", "action", "next_obs"),
               "PvtolBarrier": ("obs", "action", "next_obs")}


def load_golden(solver, B, env="Unicycle"):
    return np.load(os.path.join(GOLD, "%s_%s_B%d.npz" % (PREFIX[env], solver, B)))


def golden_env(g):
    return str(g["meta_env"]) if "meta_env" in g.files else "Unicycle"


def case_inputs(g, ci, transitions=None):
    """(batch dict of float32 tensors, eps list, node_batch tuple, updates)."""
    seed, env_name = int(g["meta_seed"]), golden_env(g)
    env = synth.fixture_env(env_name, seed)
    tr = transitions if transitions is not None else synth.transitions(env_name, 4096, seed=seed + 1, env=env)
    idx, nidx = g["c%d_idx" % ci], g["c%d_nidx" % ci]
    B = int(g["meta_B"])
    batch = {f: torch.tensor(tr[f][idx], dtype=torch.float32) for f in synth.fields(env_name)}
    eps = [torch.from_numpy(e) for e in synth.normal_eps(N_EPS[env_name], B, env.n_u, seed=100 * seed + ci)]
    node = tuple(torch.tensor(tr[f][nidx], dtype=torch.float32) for f in NODE_FIELDS[env_name])
    return batch, eps, node, int(g["c%d_updates" % ci])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-6 * (np.abs(b).max() + 1e-30) + 1e-30))) if a.size else 0.0


def vec_close(a, b, rtol, name=""):
    """Relative to the vector's own scale (max |b|): robust to tiny entries."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "%s: max err %.3e (scale %.3e) > %.1e" % (name, err, scale, rtol)
    return err
