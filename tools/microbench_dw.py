"""GPU micro-benchmark of nlbac_mlp_bwd_weights on the NODE fit's shape: f_net (3 -> 100 x4 -> 3) and g_net
(3 -> 100 x3 -> 6) over 6 x 32768 rows, 48 gradient slabs.   python tools/microbench_dw.py [rows] [slabs] [hid]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
import nlbac_amd
from nlbac_amd import _lib, arena as A

B = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
slabs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
hid = int(sys.argv[3]) if len(sys.argv) > 3 else 100
shapes = [(3, 3, 5), (3, 6, 4)]
mods = [[nn.Linear(i, hid)] + [nn.Linear(hid, hid) for _ in range(nl - 2)] + [nn.Linear(hid, o)] for i, o, nl in shapes]
ar = A.Arena("cuda", n_slabs=slabs)
hs = [A.MlpHandle(ar, [(l.weight, l.bias) for l in m]) for m in mods]
ar.finalize()
for h in hs:
    h.bind()
A.pack(hs)
nets = A.mlp_array([h.desc for h in hs])
io = A.io_array(2)
keep = []
for i, (idim, odim, nl) in enumerate(shapes):
    x, dy = torch.randn(B, idim, device="cuda"), torch.randn(B, odim, device="cuda")
    acts, dz = torch.randn(nl - 1, B, hid, device="cuda"), torch.randn(nl - 1, B, hid, device="cuda")
    keep += [x, dy, acts, dz]
    io[i].x0, io[i].x0_dim, io[i].x0_ld = x.data_ptr(), idim, idim
    io[i].acts, io[i].dz = acts.data_ptr(), dz.data_ptr()
    io[i].dy, io[i].dy_ld = dy.data_ptr(), odim
    io[i].grad = ar.grad.data_ptr()
ws = torch.empty(_lib.load().nlbac_mlp_bwd_weights_ws_floats(nets, 2, B), device="cuda")


def run():
    _lib.call("nlbac_mlp_bwd_weights", nets, io, 2, B, slabs, ar.n, ws.data_ptr(), ws.numel(), A.stream_ptr())


for _ in range(3):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
gb = sum(2 * (nl - 1) * B * hid * 4 for _, _, nl in shapes) / 1e9
print("bwd_weights rows %d slabs %d hid %d: median %.1f us  min %.1f us   (%.2f GB of dz + activations: %.2f TB/s)"
      % (B, slabs, hid, ts[len(ts) // 2], ts[0], gb, gb / ts[len(ts) // 2] * 1e3))
