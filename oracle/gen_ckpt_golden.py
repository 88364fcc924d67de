"""Checkpoint fixtures (row f4) — runs ONLY in the build container (needs ``/root/reference``).  Test infrastructure.

The REFERENCE agent (imported as in ``oracle/gen_golden.py``) writes its checkpoint files with its own
``save_model`` — ``actor.pkl``, ``critic.pkl``, ``lyapunov.pkl``, ``node_model.pkl`` (+ ``barrier.pkl`` in the
learned-barrier copy) — into ``tests/golden/ckpt_<env>/`` (``torch.save`` of ``state_dict``s: tensors only).  A second
reference agent (another seed) then restores them with the reference's ``load_weights`` (+ the NODE file, which the
reference's loader leaves out) and runs one ``update_parameters``; its outputs go to ``ckpt_<env>/expected.npz``.  The
build must load the same files (weights-only loader) and reproduce that update (tests/test_checkpoint_golden_gpu.py).

Usage: python oracle/gen_ckpt_golden.py --env Unicycle|UnicycleBarrier|SimulatedCars|Pvtol|PvtolBarrier
(one env per process: the five copies share module names)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import gen_golden as G  # noqa: E402
from oracle import nlbac_oracle as O  # noqa: E402
from nlbac_amd import synth  # noqa: E402

HIDDEN, B, SEED_A, SEED_B = 64, 64, 3, 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Unicycle", choices=["Unicycle", "UnicycleBarrier", "SimulatedCars", "Pvtol", "PvtolBarrier"])
    env_name = ap.parse_args().env
    M, S = G.import_reference(env_name)
    torch.set_num_threads(1)
    cfg = G.CFG[env_name]
    out_dir = os.path.join(ROOT, "tests", "golden", "ckpt_%s" % env_name)
    os.makedirs(out_dir, exist_ok=True)
    barrier = env_name.endswith("Barrier")

    def build(seed):
        import random
        # the reference driver seeds every generator BEFORE the agent exists (U/main.py:253-263): the critic, the
        # Lyapunov net and their targets are created ahead of the constructor's own manual_seed (sac_cbf_clf.py:47-70)
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        env = synth.fixture_env(env_name, seed)
        args = O.Args(batch_size=B, hidden_size=HIDDEN, seed=seed)
        args.gamma_b = cfg["gamma_b"]
        return S.SAC_CBF_CLF(cfg["obs"], env.action_space, env, args), env, args
    a, _, _ = build(SEED_A)
    # make the saved nets distinguishable from any fresh initialisation (Xavier weights, ZERO biases)
    with torch.no_grad():
        for mod in (a.policy, a.critic, a.lyapunovNet) + ((a.BarrierNet,) if barrier else ()):
            for p in mod.parameters():
                p.add_(0.01 * torch.randn_like(p))
    a.save_model(out_dir)
    b, env, args = build(SEED_B)
    b.load_weights(out_dir)
    b.neural_ode_model.load_state_dict(torch.load(os.path.join(out_dir, "node_model.pkl")))
    from sac_cbf_clf.dynamics import DynamicsModel
    dyn = DynamicsModel(env, args)
    tr = synth.transitions(env_name, 4096, seed=SEED_B + 1, env=env)
    fields = synth.fields(env_name)
    rs = np.random.RandomState(77)
    idx, nidx = rs.choice(4096, B, replace=False), rs.choice(4096, 256, replace=False)
    eps = synth.normal_eps(cfg["n_eps"], B, cfg["act"], seed=9)
    queue = [torch.from_numpy(e) for e in eps]
    orig = torch.distributions.Normal.rsample
    torch.distributions.Normal.rsample = lambda self, sample_shape=torch.Size(): self.loc + queue.pop(0) * self.scale
    out = dict(meta_env=env_name, meta_hidden=HIDDEN, meta_B=B, meta_seed_b=SEED_B, idx=idx, nidx=nidx,
               gamma_b=cfg["gamma_b"])
    G.summarize("pre_critic_target", G.flat_params(b.critic_target), out)     # the loader leaves the targets alone
    G.summarize("pre_critic", G.flat_params(b.critic), out)
    try:
        extra = (0,) if env_name.startswith("Pvtol") else ()          # P / NP: trailing i_episode argument
        ret = b.update_parameters(G.FakeMemory(tr, idx, fields), B, 0, dyn, G.FakeMemory(tr, nidx, fields), 10, *extra)
    finally:
        torch.distributions.Normal.rsample = orig
    out["ret"] = np.array(ret, dtype=np.float64)
    out["lambdas"] = np.array([float(x) for x in b.lambda_values])
    mods = [("critic", b.critic), ("lya", b.lyapunovNet), ("policy", b.policy), ("node", b.neural_ode_model),
            ("critic_target", b.critic_target)] + ([("barrier", b.BarrierNet)] if barrier else [("backup", b.backup_policy)])
    for name, mod in mods:
        G.summarize("p_" + name, G.flat_params(mod), out)
    np.savez_compressed(os.path.join(out_dir, "expected.npz"), **out)
    for f in sorted(os.listdir(out_dir)):
        print(f, os.path.getsize(os.path.join(out_dir, f)))
    print("ret", out["ret"])


if __name__ == "__main__":
    main()
