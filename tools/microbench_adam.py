"""GPU micro-benchmark of nlbac_adam_fused on the agent's own arenas (critic + Lyapunov with targets; policies): the
full call, and the call without its parts (fragment scatter, Polyak targets, slab sum, host mirror), back to back."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import nlbac_amd
from nlbac_amd import _lib
from nlbac_amd.arena import stream_ptr
from test_agent_parity_gpu import make_agent

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
agent, env = make_agent(B, 256, 0, "dopri5")
pin = torch.zeros(128, dtype=torch.float32).pin_memory()


def run(a, target, scatter, slabs, mirror):
    scat, scat_t = a.scatter_tables()
    tgt = a.target.data_ptr() if (target and getattr(a, "target", None) is not None) else None
    _lib.call("nlbac_adam_fused", a.theta.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), a.grad.data_ptr(),
              a.n_slabs if slabs else 1, a.n, a.n, a.state.data_ptr(), 3e-4, tgt, 0.005 if tgt else -1.0,
              scat.data_ptr() if scatter else None,
              scat_t.data_ptr() if (scatter and tgt and scat_t is not None) else None, a.scatter_slots, 0, None, None,
              agent.sc.data_ptr() if mirror else None, pin.data_ptr() if mirror else None, 128 if mirror else 0, stream_ptr())


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, a in (("critic+Lyapunov", agent.ar_c), ("policies", agent.ar_a)):
    print("%s: n = %d, %d slabs, %d scatter slots, target %s" % (name, a.n, a.n_slabs, a.scatter_slots, getattr(a, "target", None) is not None))
    for label, kw in (("full", dict(target=True, scatter=True, slabs=True, mirror=False)),
                      ("full + host mirror", dict(target=True, scatter=True, slabs=True, mirror=True)),
                      ("no scatter", dict(target=True, scatter=False, slabs=True, mirror=False)),
                      ("no target", dict(target=False, scatter=True, slabs=True, mirror=False)),
                      ("one slab", dict(target=True, scatter=True, slabs=False, mirror=False)),
                      ("bare (one slab, no scatter, no target)", dict(target=False, scatter=False, slabs=False, mirror=False))):
        ts = sorted(timeit(lambda: run(a, **kw)) for _ in range(3))
        print("   %-42s %.1f us" % (label, ts[1]))
