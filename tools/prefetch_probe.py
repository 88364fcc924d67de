"""One-off check: the update sequence with the next minibatch + policy forward queued behind each update's last launch
(update_on_device(prefetch=...)) returns bit-identical losses to drawing at the start of every update."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from nlbac_amd import synth
from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory


def run(prefetch, n=45, B=4096):
    torch.manual_seed(0)
    env = Bn.make_env("Unicycle", 0)
    args = Bn.Args(B)
    args.gamma_b = Bn.GAMMA_B["Unicycle"]
    agent = SAC_CBF_CLF(env.obs_dim, env.action_space, env, args)
    agent.solver = "dopri5"
    replay = DeviceReplayMemory(Bn.REPLAY_ROWS, 1234, agent, device_rng=True)
    replay.push_rows(Bn.replay_rows(agent, synth.transitions("Unicycle", Bn.REPLAY_ROWS, seed=1, env=env)))
    ws = agent._workspace(B)
    fit_rows = torch.empty(Bn.NODE_FIT_ROWS, agent.lay.LD, device=agent.device)
    draw = lambda: replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)
    out = []
    for i in range(n):
        if prefetch:
            if i % 10 == 0:
                if ws.__dict__.get("_prefetched") is None:
                    agent.update_prefetch(ws, i, draw)
                agent.fit_node_rows(replay.sample_rows(Bn.NODE_FIT_ROWS, out=fit_rows))
            out.append(agent.update_on_device(ws, i, prefetch=draw))
        else:
            draw()
            if i % 10 == 0:
                agent.fit_node_rows(replay.sample_rows(Bn.NODE_FIT_ROWS, out=fit_rows))
            out.append(agent.update_on_device(ws, i, eps_ready=True))
    return out


a, b, c = run(False), run(True), run(False)
same_ab = sum(x == y for x, y in zip(a, b))
same_ac = sum(x == y for x, y in zip(a, c))
print("classic vs prefetch: %d / %d updates bit-identical; classic vs classic: %d / %d" % (same_ab, len(a), same_ac, len(a)))
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("first difference at update", i, x, y)
        break
