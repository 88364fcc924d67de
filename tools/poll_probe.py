"""How long does the host wait for a dopri5 accept decision, and how: event behind the controller launch (NLBAC_CTL_POLL=0)
against polling the stamped control block?  Headline update loop (bench.py's lean form), per-wait wall time of
AffineNodeSolver._ctl_read and per-update time.   python tools/poll_probe.py [env] [batch] [updates]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from nlbac_amd import synth, odeint
from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
from nlbac_amd.sac_cbf_clf.replay_memory import DeviceReplayMemory

name = sys.argv[1] if len(sys.argv) > 1 else "Unicycle"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 120
SYNC = {"0": False, "1": True, "lagged": "lagged"}[sys.argv[4]] if len(sys.argv) > 4 else False
FIT = len(sys.argv) > 5 and sys.argv[5] == "fit"


def run(poll):
    odeint.CTL_POLL = poll
    torch.manual_seed(0)
    env = Bn.make_env(name, 0)
    args = Bn.Args(B)
    args.gamma_b = Bn.GAMMA_B[name]
    agent = SAC_CBF_CLF(env.obs_dim, env.action_space, env, args)
    agent.solver = "dopri5"
    replay = DeviceReplayMemory(Bn.REPLAY_ROWS, 1234, agent, device_rng=True)
    replay.push_rows(Bn.replay_rows(agent, synth.transitions(name, Bn.REPLAY_ROWS, seed=1, env=env)))
    ws = agent._workspace(B)
    draw = lambda: replay.sample_rows(B, out=ws.mb, eps_out=ws.eps)
    fit_rows = torch.empty(Bn.NODE_FIT_ROWS, agent.lay.LD, device=agent.device)
    waits = []
    orig = odeint.AffineNodeSolver._ctl_read

    def timed(self, P):
        t = time.perf_counter()
        c = orig(self, P)
        waits.append(time.perf_counter() - t)
        return c
    odeint.AffineNodeSolver._ctl_read = timed
    try:
        t_upd = []
        for i in range(N):
            if i == 20:
                torch.cuda.synchronize(); waits.clear(); t0 = time.perf_counter()
            if FIT and i % 10 == 0:
                if ws.__dict__.get("_prefetched") is None:
                    agent.update_prefetch(ws, i, draw)
                agent.fit_node_rows(replay.sample_rows(Bn.NODE_FIT_ROWS, out=fit_rows))
            agent.update_on_device(ws, i if FIT else i + 1, sync=SYNC, prefetch=draw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (N - 20)
    finally:
        odeint.AffineNodeSolver._ctl_read = orig
    w = sorted(waits)
    print("sync=%s fit=%s " % (SYNC, FIT) + "poll=%d  %.4f ms/update  waits per update %.2f  wait: median %.1f us  p90 %.1f  max %.1f  sum/update %.1f us"
          % (poll, dt * 1e3, len(w) / (N - 20), w[len(w) // 2] * 1e6, w[int(len(w) * 0.9)] * 1e6, w[-1] * 1e6,
             sum(w) / (N - 20) * 1e6), flush=True)


for poll in (False, True, False, True):
    run(poll)
