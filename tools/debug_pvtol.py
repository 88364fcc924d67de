"""Debug: constraint-loss gradient w.r.t. the actions, HIP path vs oracle autograd (Pvtol)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import nlbac_amd
from nlbac_amd import synth
from nlbac_amd.envspec import make_env
from oracle import nlbac_oracle as O
from common import case_inputs, load_golden
from test_agent_parity_gpu import make_agent

solver, B, env_name = sys.argv[1], int(sys.argv[2]), "Pvtol"
g = load_golden(solver, B, env_name)
seed, hidden, gamma_b = 0, 256, float(g["meta_gamma_b"])
agent, env = make_agent(B, hidden, seed, solver, env_name, gamma_b)
oargs = O.Args(batch_size=B, hidden_size=hidden, seed=seed); oargs.gamma_b = gamma_b
oracle = O.make_oracle(synth.fixture_env(env_name, seed), oargs, synth.agent_weights(env_name, hidden, seed), solver=solver)
tr = synth.transitions(env_name, 4096, seed=seed + 1, env=env)
saved = {}
orig_pt, orig_al = oracle._primary_terms, oracle._auglag
def pt(batch, pi, eps):
    saved["pi"] = pi
    m, e = orig_pt(batch, pi, eps)
    saved["matr"] = m
    return m, e
def al(required, lambdas, updates, with_clf, backup=False):
    out = orig_al(required, lambdas, updates, with_clf, backup)
    if with_clf:
        saved["dpi"] = torch.autograd.grad(out["loss"], saved["pi"], retain_graph=True)[0]
    return out
oracle._primary_terms, oracle._auglag = pt, al
batch, eps, node, updates = case_inputs(g, 0, tr)
R = oracle.update(batch, eps, updates, node_batch=node)
agent.set_noise(eps)
host = tuple(batch[f].numpy() for f in synth.FIELDS)
agent.update_from_host(host, updates, tuple(t.numpy() for t in node))
torch.cuda.synchronize()
ws = agent._ws[B]
du = agent.task.steps[0]._buf("du", 2 * B, 2).cpu().numpy()[:B]
ref = saved["dpi"].numpy()
err = np.abs(du - ref).max(1)
act = (saved["matr"].detach().numpy() > 0)
print("scale", np.abs(ref).max(), "max err", err.max())
for i in np.argsort(-err)[:12]:
    print(i, "err %.3e" % err[i], "du", du[i], "ref", ref[i], "active", np.nonzero(act[i])[0])
dx1 = ws.dx1.cpu().numpy()[:B]
print("rows with big err: obs y", batch["obs"][np.argsort(-err)[:6], 1].numpy(), "op-x", (batch["obs"][:, 7] - batch["obs"][:, 0])[np.argsort(-err)[:6]].numpy())

# ---- per-tensor policy gradient comparison
gp = R["g_policy"].numpy()
off = 0
for name, prm in agent.policy.named_parameters():
    n = prm.numel()
    v = agent.ar_a.grad_view(prm).reshape(-1).cpu().numpy()
    o = gp[off:off + n]
    off += n
    print("%-24s |o| %.3e  max abs err %.3e  rel-to-max %.3e" % (name, np.abs(o).max(), np.abs(v - o).max(), np.abs(v - o).max() / (np.abs(o).max() + 1e-30)))
print("pi err", np.abs(ws.pi2[:B].cpu().numpy() - R["pi"].numpy()).max(), "logpi err", np.abs(ws.logp2[:B].cpu().numpy() - R["log_pi"].reshape(-1).numpy()).max())
heads = ws.heads2[:B].cpu().numpy()
print("heads mean range", heads[:, :2].min(), heads[:, :2].max(), "log_std range", heads[:, 2:].min(), heads[:, 2:].max())
dh = ws.dheads2[:B].cpu().numpy()
big = np.argsort(-np.abs(dh).max(1))[:8]
for i in big:
    print(i, "heads", heads[i], "dheads", dh[i], "eps", eps[1][i].numpy(), "obs y", float(batch["obs"][i, 1]), "op", float(batch["obs"][i, 7]))
