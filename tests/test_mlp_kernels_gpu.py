"""GPU: fused MLP kernels (forward / backward-data / backward-weights, Adam)
through the C ABI vs a plain PyTorch fp32 reference of the same op.

Tolerance: 1e-4 relative to each tensor's scale (BASELINE.json north_star:
"within 1e-4 rel fp32"); the kernels use exact-fp32 MFMA so they typically
land at ~1e-6.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn as nn

from common import vec_close

pytestmark = pytest.mark.gpu

TOL = 1e-4


def build(in_dim, hid, out_dim, n_layers, seed, n_slabs=4):
    from nlbac_amd import arena as A
    torch.manual_seed(seed)
    dims = [in_dim] + [hid] * (n_layers - 1) + [out_dim]
    lins = [nn.Linear(dims[i], dims[i + 1]) for i in range(n_layers)]
    for l in lins:
        nn.init.uniform_(l.bias, -0.3, 0.3)
    ref = [(l.weight.detach().clone(), l.bias.detach().clone()) for l in lins]
    ar = A.Arena("cuda", n_slabs=n_slabs)
    h = A.MlpHandle(ar, [(l.weight, l.bias) for l in lins])
    ar.finalize()
    h.bind()
    A.pack([h])
    return ar, h, lins, ref


def torch_ref(ref, x, dy):
    ws = [(w.clone().requires_grad_(), b.clone().requires_grad_()) for w, b in ref]
    x = x.clone().requires_grad_()
    h = x
    acts = []
    for i, (w, b) in enumerate(ws):
        h = torch.nn.functional.linear(h, w, b)
        if i < len(ws) - 1:
            h = torch.relu(h)
            acts.append(h)
    h.backward(dy)
    return h.detach(), [a.detach() for a in acts], x.grad, [(w.grad, b.grad) for w, b in ws]


@pytest.mark.parametrize("in_dim,hid,out_dim,n_layers,B,split", [
    (9, 256, 1, 3, 100, 7),      # Q-net: obs(7) || action(2)
    (2, 256, 1, 3, 33, 0),       # Lyapunov net
    (7, 256, 4, 3, 4096, 0),     # policy heads at the headline batch
    (3, 100, 3, 5, 77, 0),       # f_net (hidden 100 -> padded tiles)
    (3, 100, 6, 4, 32, 0),       # g_net
    (12, 64, 10, 4, 50, 10),     # SimulatedCars NODE shape
    (13, 256, 1, 3, 70, 11),     # Pvtol Q-net: obs(11) || action(2)  (first layer spans two K chunks)
    (6, 100, 12, 4, 45, 0),      # Pvtol g_net
    (16, 64, 16, 3, 40, 0),      # widest skinny layers the weight-gradient kernel takes
    (3, 8, 2, 2, 5, 0),          # smallest legal net
    (5, 80, 3, 4, 130, 3),       # 6-tile form of the one-launch weight gradients (dwordx4 + dwordx2 columns), split input
    (4, 96, 5, 3, 61, 0),        # the widest 6-tile net, no hid x hid... one hid x hid layer, ragged rows (61 = 15 k-steps + 1)
    (6, 112, 4, 4, 203, 0),      # the widest net the one-launch kernel takes (7 full tiles)
    (3, 100, 3, 5, 1000, 0),     # f_net on more rows than slabs x waves x ring depth
])
def test_mlp_fwd_bwd_matches_torch(in_dim, hid, out_dim, n_layers, B, split):
    from nlbac_amd import _lib, arena as A
    ar, h, lins, ref = build(in_dim, hid, out_dim, n_layers, seed=B)
    dev = "cuda"
    g = torch.Generator().manual_seed(B + 1)
    x = torch.randn(B, in_dim, generator=g)
    dy = torch.randn(B, out_dim, generator=g)
    y_ref, acts_ref, dx_ref, grads_ref = torch_ref(ref, x, dy)

    xd, dyd = x.to(dev), dy.to(dev)
    if split:
        x0 = xd[:, :split].contiguous()
        x1 = xd[:, split:].contiguous()
    nw = n_layers - 1
    y = torch.full((B, out_dim), float("nan"), device=dev)
    acts = torch.full((nw, B, hid), float("nan"), device=dev)
    dz = torch.full((nw, B, hid), float("nan"), device=dev)
    dx = torch.full((B, in_dim), float("nan"), device=dev)
    io = A.io_array(1)
    if split:
        io[0].x0, io[0].x0_dim, io[0].x0_ld = x0.data_ptr(), split, split
        io[0].x1, io[0].x1_dim, io[0].x1_ld = x1.data_ptr(), in_dim - split, in_dim - split
    else:
        io[0].x0, io[0].x0_dim, io[0].x0_ld = xd.data_ptr(), in_dim, in_dim
    io[0].y, io[0].y_ld = y.data_ptr(), out_dim
    io[0].acts = acts.data_ptr()
    io[0].dy, io[0].dy_ld = dyd.data_ptr(), out_dim
    io[0].dz = dz.data_ptr()
    io[0].dx, io[0].dx_ld = dx.data_ptr(), in_dim
    io[0].grad = ar.grad.data_ptr()
    nets = A.mlp_array([h.desc])
    s = A.stream_ptr()
    _lib.call("nlbac_mlp_fwd", nets, io, 1, B, s)
    _lib.call("nlbac_mlp_bwd_data", nets, io, 1, B, s)
    A.bwd_weights(nets, io, 1, B, ar.n_slabs, ar.n, "cuda")
    torch.cuda.synchronize()

    vec_close(y.cpu(), y_ref, TOL, "y")
    for l in range(nw):
        vec_close(acts[l].cpu(), acts_ref[l], TOL, "acts%d" % l)
    vec_close(dx.cpu(), dx_ref, TOL, "dx")
    for l, lin in enumerate(lins):
        gw = ar.grad_view(lin.weight).cpu()
        gb = ar.grad_view(lin.bias).cpu()
        vec_close(gw, grads_ref[l][0], TOL, "dW%d" % l)
        vec_close(gb, grads_ref[l][1], TOL, "db%d" % l)


@pytest.mark.parametrize("fused", [False, True], ids=["prepare+step+pack", "adam_fused"])
def test_multi_net_launch_and_adam_soft_update(fused):
    """Three nets in one launch (grid.y) + Adam with slab reduction + fused Polyak update
    vs torch.optim.Adam / the reference's soft_update arithmetic.  ``fused``: the one-launch optimiser step that
    also refreshes the MFMA-fragment weight copies through the arena's scatter tables."""
    from nlbac_amd import _lib, arena as A
    torch.manual_seed(0)
    B, hid = 96, 256
    mods = [[nn.Linear(9, hid), nn.Linear(hid, hid), nn.Linear(hid, 1)] for _ in range(2)]
    mods.append([nn.Linear(2, hid), nn.Linear(hid, hid), nn.Linear(hid, 1)])
    ref = [[(l.weight.detach().clone(), l.bias.detach().clone()) for l in m] for m in mods]
    ar = A.Arena("cuda", n_slabs=3, with_target=True)
    hs = [A.MlpHandle(ar, [(l.weight, l.bias) for l in m]) for m in mods]
    ar.finalize()
    ar.hard_update_target()
    for h in hs:
        h.bind()
    A.pack(hs)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(B, 9, generator=g), torch.randn(B, 9, generator=g), torch.randn(B, 2, generator=g)]
    dys = [torch.randn(B, 1, generator=g) for _ in range(3)]
    io = A.io_array(3)
    keep = []
    for i in range(3):
        xd, dyd = xs[i].cuda(), dys[i].cuda()
        y = torch.empty(B, 1, device="cuda")
        acts = torch.empty(2, B, hid, device="cuda")
        dz = torch.empty(2, B, hid, device="cuda")
        keep += [xd, dyd, y, acts, dz]
        io[i].x0, io[i].x0_dim, io[i].x0_ld = xd.data_ptr(), xs[i].shape[1], xs[i].shape[1]
        io[i].y, io[i].y_ld = y.data_ptr(), 1
        io[i].acts, io[i].dz = acts.data_ptr(), dz.data_ptr()
        io[i].dy, io[i].dy_ld = dyd.data_ptr(), 1
        io[i].grad = ar.grad.data_ptr()
    nets = A.mlp_array([h.desc for h in hs])
    s = A.stream_ptr()
    tau, lr = 0.005, 4e-4
    # torch reference: two Adam steps + soft updates
    tparams = [[(w.clone().requires_grad_(), b.clone().requires_grad_()) for w, b in m] for m in ref]
    flat = [t for m in tparams for wb in m for t in wb]
    opt = torch.optim.Adam(flat, lr=lr)
    targ = [t.detach().clone() for t in flat]
    for it in range(2):
        _lib.call("nlbac_mlp_fwd", nets, io, 3, B, s)
        _lib.call("nlbac_mlp_bwd_data", nets, io, 3, B, s)
        A.bwd_weights(nets, io, 3, B, ar.n_slabs, ar.n, "cuda")
        if fused:
            scat, scat_t = ar.scatter_tables()
            # (mirror: the step's last workgroup also delivers a block of device floats to pinned host memory)
            msrc = torch.arange(37, dtype=torch.float32, device="cuda") + it
            mdst = torch.zeros(37, dtype=torch.float32).pin_memory()
            _lib.call("nlbac_adam_fused", ar.theta.data_ptr(), ar.m.data_ptr(), ar.v.data_ptr(), ar.grad.data_ptr(),
                      ar.n_slabs, ar.n, ar.n, ar.state.data_ptr(), lr, ar.target.data_ptr(), tau, scat.data_ptr(),
                      scat_t.data_ptr(), ar.scatter_slots, 0, None, None, msrc.data_ptr(), mdst.data_ptr(), 37, s)
            torch.cuda.synchronize()
            assert torch.equal(mdst, msrc.cpu()), "adam_fused did not mirror the scalars block to pinned memory"

        else:
            _lib.call("nlbac_adam_prepare", ar.state.data_ptr(), lr, s)
            _lib.call("nlbac_adam_step", ar.theta.data_ptr(), ar.m.data_ptr(), ar.v.data_ptr(), ar.grad.data_ptr(),
                      ar.n_slabs, ar.n, ar.n, ar.state.data_ptr(), ar.target.data_ptr(), tau, s)
            A.pack(hs)
        opt.zero_grad()
        for i in range(3):
            h = xs[i]
            for j, (w, b) in enumerate(tparams[i]):
                h = torch.nn.functional.linear(h, w, b)
                if j < 2:
                    h = torch.relu(h)
            h.backward(dys[i])
        opt.step()
        with torch.no_grad():
            for t, p in zip(targ, flat):
                t.copy_(t * (1.0 - tau) + p * tau)
    torch.cuda.synchronize()
    st = ar.state.cpu()
    assert int(st[0]) == 2 and int(st[3]) == 0, "step counter / ticket after two optimiser steps: %s" % st
    if fused:       # the scattered fragment copies are bit-identical to a fresh pack of the stepped parameters
        got = [(h.packed.clone(), h.packed_target.clone()) for h in hs]
        A.pack(hs)
        A.pack(hs, target=True)
        torch.cuda.synchronize()
        for h, (pk, pkt) in zip(hs, got):
            assert torch.equal(pk, h.packed) and torch.equal(pkt, h.packed_target)
    k = 0
    for i, m in enumerate(mods):
        for l in m:
            for p in (l.weight, l.bias):
                # (1e-5: Adam's m / (sqrt(v) + eps) turns the rounding of a near-zero gradient entry into a visible fraction
                #  of a step; the weight-gradient kernels sum the rows in another order than torch — 2.6e-6 observed)
                vec_close(p.detach().cpu(), flat[k].detach(), 1e-5, "param %d" % k)
                off = ar.offset_of[id(p)]
                vec_close(ar.target[off:off + p.numel()].cpu().view(p.shape), targ[k], 1e-5, "target %d" % k)
                k += 1


@pytest.mark.parametrize("B,hid,dims", [(4096, 256, (9, 1)), (100, 256, (7, 4)), (1000, 100, (3, 6)), (77, 160, (13, 2))])
def test_skinny_gradient_partials_from_the_data_backward_are_bit_identical(B, hid, dims):
    """With ``nlbac_mlp_io::skinny_ws`` set, ``nlbac_mlp_bwd_data`` leaves the per-32-row partial sums of the bias,
    first- and last-layer gradients (it holds every dz tile in LDS) and ``nlbac_mlp_bwd_weights`` only reduces them: the
    same sums in the same order as the separate partial pass, so every gradient must match bit for bit (widths above 112;
    narrower nets: see below)."""
    from nlbac_amd import _lib, arena as A
    in_dim, out_dim = dims
    torch.manual_seed(B + hid)
    mods = [[nn.Linear(in_dim, hid), nn.Linear(hid, hid), nn.Linear(hid, hid), nn.Linear(hid, out_dim)] for _ in range(2)]
    ar = A.Arena("cuda", n_slabs=4)
    hs = [A.MlpHandle(ar, [(l.weight, l.bias) for l in m]) for m in mods]
    ar.finalize()
    for h in hs:
        h.bind()
    A.pack(hs)
    g = torch.Generator().manual_seed(1)
    keep, grads = [], []
    nets = A.mlp_array([h.desc for h in hs])
    s = A.stream_ptr()
    for fused in (False, True):
        io = A.io_array(2)
        for i in range(2):
            x, dy = torch.randn(B, in_dim, generator=torch.Generator().manual_seed(10 + i)).cuda(), \
                torch.randn(B, out_dim, generator=torch.Generator().manual_seed(20 + i)).cuda()
            y, acts, dz = torch.empty(B, out_dim, device="cuda"), torch.empty(3, B, hid, device="cuda"), \
                torch.empty(3, B, hid, device="cuda")
            keep += [x, dy, y, acts, dz]
            io[i].x0, io[i].x0_dim, io[i].x0_ld = x.data_ptr(), in_dim, in_dim
            io[i].y, io[i].y_ld = y.data_ptr(), out_dim
            io[i].acts, io[i].dz = acts.data_ptr(), dz.data_ptr()
            io[i].dy, io[i].dy_ld = dy.data_ptr(), out_dim
            io[i].grad = ar.grad.data_ptr()
        ws = A.skinny_partials_ws(nets, (io,), 2, B, "cuda") if fused else None
        assert (ws is not None) == fused
        ar.grad.fill_(float("nan"))
        _lib.call("nlbac_mlp_fwd", nets, io, 2, B, s)
        _lib.call("nlbac_mlp_bwd_data", nets, io, 2, B, s)
        A.bwd_weights(nets, io, 2, B, ar.n_slabs, ar.n, "cuda", ws=ws)
        torch.cuda.synchronize()
        keep.append(ws)
        grads.append(ar.grad.clone())
    a, b = grads
    for m in mods:
        for l in m:
            for p in (l.weight, l.bias):
                off = ar.offset_of[id(p)]
                ga, gb = a[:, off:off + p.numel()], b[:, off:off + p.numel()]
                if hid <= 112:
                    # nets this narrow take the one-launch kernel (mlp_dw16_kernels.hip) on the separate route: every
                    # slab holds a partial of every gradient, summed in another order than the fused route's — the
                    # slab sums agree to fp32 summation error (1e-5 of the largest entry; the parity bar is 1e-4)
                    sa, sb = ga.sum(0), torch.nan_to_num(gb).sum(0)
                    assert torch.isfinite(sa).all()
                    assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max()) + 1e-6
                    continue
                assert torch.equal(torch.nan_to_num(ga), torch.nan_to_num(gb)) and \
                    torch.equal(torch.isnan(ga), torch.isnan(gb))
                assert torch.isfinite(ga[0]).all()


def test_one_launch_weight_gradients_of_nets_of_different_width_and_depth():
    """``mlp_dw16_kernels.hip`` picks its tile count from the widest net of a launch and its layer slots from the
    deepest: a 100-wide 5-layer net and a 64-wide 3-layer net in ONE ``nlbac_mlp_bwd_weights`` call (the narrower net's
    lanes past its width read out of range = zeros; its missing layer slots return at once), against torch autograd."""
    from nlbac_amd import _lib, arena as A
    shapes = [(3, 100, 3, 5), (12, 64, 10, 3)]
    B = 333
    torch.manual_seed(5)
    mods = [[nn.Linear(i, h)] + [nn.Linear(h, h) for _ in range(nl - 2)] + [nn.Linear(h, o)] for i, h, o, nl in shapes]
    host = [[(l.weight.detach().clone(), l.bias.detach().clone()) for l in m] for m in mods]     # (bind() moves the parameters)
    ar = A.Arena("cuda", n_slabs=6)
    hs = [A.MlpHandle(ar, [(l.weight, l.bias) for l in m]) for m in mods]
    ar.finalize()
    for h in hs:
        h.bind()
    A.pack(hs)
    nets = A.mlp_array([h.desc for h in hs])
    io = A.io_array(2)
    keep, refs = [], []
    for k, ((idim, hid, odim, nl), m) in enumerate(zip(shapes, mods)):
        x = torch.randn(B, idim, generator=torch.Generator().manual_seed(10 + k))
        dy = torch.randn(B, odim, generator=torch.Generator().manual_seed(20 + k))
        xd, dyd = x.cuda(), dy.cuda()
        y = torch.empty(B, odim, device="cuda")
        acts, dz = torch.empty(nl - 1, B, hid, device="cuda"), torch.empty(nl - 1, B, hid, device="cuda")
        keep += [xd, dyd, y, acts, dz]
        io[k].x0, io[k].x0_dim, io[k].x0_ld = xd.data_ptr(), idim, idim
        io[k].y, io[k].y_ld = y.data_ptr(), odim
        io[k].acts, io[k].dz = acts.data_ptr(), dz.data_ptr()
        io[k].dy, io[k].dy_ld = dyd.data_ptr(), odim
        io[k].grad = ar.grad.data_ptr()
        refs.append(torch_ref(host[k], x, dy)[3])
    ar.grad.fill_(float("nan"))
    s = A.stream_ptr()
    # (the two nets differ in depth, so the forward / data backward go net by net; the weight gradients in one call)
    for k in range(2):
        one_n, one_io = A.mlp_array([hs[k].desc]), A.io_array(1)
        C.memmove(C.byref(one_io[0]), C.byref(io[k]), C.sizeof(io[k]))
        _lib.call("nlbac_mlp_fwd", one_n, one_io, 1, B, s)
        _lib.call("nlbac_mlp_bwd_data", one_n, one_io, 1, B, s)
    A.bwd_weights(nets, io, 2, B, ar.n_slabs, ar.n, "cuda")
    torch.cuda.synchronize()
    for m, ref in zip(mods, refs):
        for l, (gw, gb) in zip(m, ref):
            vec_close(ar.grad_view(l.weight).cpu(), gw, TOL, "dW")
            vec_close(ar.grad_view(l.bias).cpu(), gb, TOL, "db")


def test_dx_first_limits_the_input_gradient_to_the_wanted_columns():
    """``nlbac_mlp_io::dx_first``: the data backward writes dx for input columns >= dx_first only (the Q(s, pi) nets'
    gradient is consumed for the action columns) — those columns equal the full gradient's, the others stay untouched."""
    from nlbac_amd import _lib, arena as A
    in_dim, hid, out_dim, B = 9, 256, 1, 200
    ar, h, lins, ref = build(in_dim, hid, out_dim, 3, seed=11)
    x = torch.randn(B, in_dim, generator=torch.Generator().manual_seed(1))
    dy = torch.randn(B, out_dim, generator=torch.Generator().manual_seed(2))
    _, _, dx_ref, _ = torch_ref(ref, x, dy)
    xd, dyd = x.cuda(), dy.cuda()
    y = torch.empty(B, out_dim, device="cuda")
    acts, dz = torch.empty(2, B, hid, device="cuda"), torch.empty(2, B, hid, device="cuda")
    nets, s = A.mlp_array([h.desc]), A.stream_ptr()
    for first in (0, 4, 7, 8):
        dx = torch.full((B, in_dim), float("nan"), device="cuda")
        io = A.io_array(1)
        io[0].x0, io[0].x0_dim, io[0].x0_ld = xd.data_ptr(), in_dim, in_dim
        io[0].y, io[0].y_ld = y.data_ptr(), out_dim
        io[0].acts, io[0].dz = acts.data_ptr(), dz.data_ptr()
        io[0].dy, io[0].dy_ld = dyd.data_ptr(), out_dim
        io[0].dx, io[0].dx_ld, io[0].dx_first = dx.data_ptr(), in_dim, first
        _lib.call("nlbac_mlp_fwd", nets, io, 1, B, s)
        _lib.call("nlbac_mlp_bwd_data", nets, io, 1, B, s)
        torch.cuda.synchronize()
        got = dx.cpu()
        assert torch.isnan(got[:, :first]).all(), "columns below dx_first were written"
        vec_close(got[:, first:], dx_ref[:, first:], TOL, "dx[:, %d:]" % first)
    io[0].dx_first = in_dim
    with pytest.raises(_lib.NlbacError):
        _lib.call("nlbac_mlp_bwd_data", nets, io, 1, B, s)


def test_one_launch_weight_gradients_on_random_narrow_shapes():
    """Seeded sweep over what ``mlp_dw16_kernels.hip`` has to cope with: widths 8 .. 112 (every tile form: 4 / 6 / 7 tiles,
    partly filled last tiles), depths 2 .. 6, input / output widths 1 .. 16, split inputs, row counts that are not
    multiples of 4, more slabs than k-steps — weight and bias gradients of every layer against torch autograd."""
    from nlbac_amd import _lib, arena as A
    rng = np.random.RandomState(123)
    for case in range(24):
        hid = int(rng.choice([8, 20, 36, 48, 64, 68, 80, 96, 100, 104, 112]))
        n_layers = int(rng.randint(2, 7))
        in_dim, out_dim = int(rng.randint(1, 17)), int(rng.randint(1, 17))
        B = int(rng.choice([1, 3, 5, 31, 64, 77, 130, 257, 1001]))
        n_slabs = int(rng.choice([1, 3, 8, 51]))
        split = int(rng.randint(1, in_dim)) if in_dim > 1 and rng.rand() < 0.5 else 0
        ar, h, lins, ref = build(in_dim, hid, out_dim, n_layers, seed=1000 + case, n_slabs=n_slabs)
        g = torch.Generator().manual_seed(case)
        x, dy = torch.randn(B, in_dim, generator=g), torch.randn(B, out_dim, generator=g)
        _, _, _, grads_ref = torch_ref(ref, x, dy)
        xd, dyd = x.cuda(), dy.cuda()
        nw = n_layers - 1
        y = torch.empty(B, out_dim, device="cuda")
        acts, dz = torch.empty(nw, B, hid, device="cuda"), torch.empty(nw, B, hid, device="cuda")
        io = A.io_array(1)
        keep = [xd, dyd, y, acts, dz]
        if split:
            x0, x1 = xd[:, :split].contiguous(), xd[:, split:].contiguous()
            keep += [x0, x1]
            io[0].x0, io[0].x0_dim, io[0].x0_ld = x0.data_ptr(), split, split
            io[0].x1, io[0].x1_dim, io[0].x1_ld = x1.data_ptr(), in_dim - split, in_dim - split
        else:
            io[0].x0, io[0].x0_dim, io[0].x0_ld = xd.data_ptr(), in_dim, in_dim
        io[0].y, io[0].y_ld = y.data_ptr(), out_dim
        io[0].acts, io[0].dz = acts.data_ptr(), dz.data_ptr()
        io[0].dy, io[0].dy_ld = dyd.data_ptr(), out_dim
        io[0].grad = ar.grad.data_ptr()
        nets, s = A.mlp_array([h.desc]), A.stream_ptr()
        ar.grad.fill_(float("nan"))
        _lib.call("nlbac_mlp_fwd", nets, io, 1, B, s)
        _lib.call("nlbac_mlp_bwd_data", nets, io, 1, B, s)
        A.bwd_weights(nets, io, 1, B, ar.n_slabs, ar.n, "cuda")
        torch.cuda.synchronize()
        tag = "case %d: %d -> %d x%d -> %d, B %d, %d slabs, split %d" % (case, in_dim, hid, nw, out_dim, B, n_slabs, split)
        for l, lin in enumerate(lins):
            vec_close(ar.grad_view(lin.weight).cpu(), grads_ref[l][0], TOL, tag + " dW%d" % l)
            vec_close(ar.grad_view(lin.bias).cpu(), grads_ref[l][1], TOL, tag + " db%d" % l)


@pytest.mark.parametrize("in_dim,hid,out_dim,B,keep_rows", [
    (9, 256, 1, 4096, False),     # Q(s, pi): differentiated w.r.t. its inputs only
    (2, 256, 1, 77, False),       # V(p(x')), ragged rows
    (7, 256, 4, 100, True),       # mask words AND activation rows (a net whose weight gradients are wanted too)
    (13, 128, 2, 333, False),     # width 128: two blocks per quarter
    (5, 64, 3, 50, True),
    (11, 256, 12, 65, False),     # out_dim > 4: the four-k-step form of the top product
])
def test_relu_mask_words_replace_the_saved_activation_rows(in_dim, hid, out_dim, B, keep_rows):
    """``nlbac_mlp_io::masks``: the register-resident forward leaves ReLU mask words (64 B per row), the data backward
    gates with them — with ``acts == NULL`` for nets that want dx only, or next to the rows.  dx (and dz, where rows are
    kept) against torch autograd; the words themselves against the activations' signs."""
    from nlbac_amd import _lib, arena as A
    ar, h, lins, ref = build(in_dim, hid, out_dim, 3, seed=B)
    nets, s = A.mlp_array([h.desc]), A.stream_ptr()
    assert _lib.load().nlbac_mlp_masks_ok(nets, 1) == 1
    g = torch.Generator().manual_seed(B + 1)
    x, dy = torch.randn(B, in_dim, generator=g), torch.randn(B, out_dim, generator=g)
    x[B // 2] = 0.0                                   # a row of zeros: relu'(0) = 0 must hold bit for bit (zero biases apart)
    y_ref, acts_ref, dx_ref, _ = torch_ref(ref, x, dy)
    xd, dyd = x.cuda(), dy.cuda()
    y = torch.full((B, out_dim), float("nan"), device="cuda")
    masks = torch.full((2, B, 8), -1, dtype=torch.int32, device="cuda")
    acts = torch.full((2, B, hid), float("nan"), device="cuda")
    dz = torch.full((2, B, hid), float("nan"), device="cuda")
    dx = torch.full((B, in_dim), float("nan"), device="cuda")
    io = A.io_array(1)
    io[0].x0, io[0].x0_dim, io[0].x0_ld = xd.data_ptr(), in_dim, in_dim
    io[0].y, io[0].y_ld = y.data_ptr(), out_dim
    io[0].masks = masks.data_ptr()
    io[0].dy, io[0].dy_ld = dyd.data_ptr(), out_dim
    io[0].dx, io[0].dx_ld = dx.data_ptr(), in_dim
    if keep_rows:
        io[0].acts, io[0].dz = acts.data_ptr(), dz.data_ptr()
    _lib.call("nlbac_mlp_fwd", nets, io, 1, B, s)
    _lib.call("nlbac_mlp_bwd_data", nets, io, 1, B, s)
    torch.cuda.synchronize()
    vec_close(y.cpu(), y_ref, TOL, "y")
    vec_close(dx.cpu(), dx_ref, TOL, "dx")
    # the bits.  Width 64 (half-panel kernels): unit 16 (NBH w + j) + 4 q + r of row b, layer l <-> bit 4 NBH - 1 - (4 j + r)
    # of the uint32 masks[l, b, 2 q + w]; widths 128 / 256 (quarter-panel kernels): unit 16 (NBQ cq + j) + 4 q + r <-> bit
    # 4 NBQ - 1 - (4 j + r) of the uint16 at [l, b, q, cq]
    u = np.arange(hid)
    blk, q, r = u // 16, (u % 16) // 4, u % 4
    if hid == 64:
        nbh = hid // 32
        m = masks.cpu().numpy().astype(np.uint32)
        w, j = blk // nbh, blk % nbh
        slot, bit = 2 * q + w, 4 * nbh - 1 - (4 * j + r)
    else:
        nbq = hid // 64
        m = masks.cpu().numpy().view(np.uint16).reshape(2, B, 16).astype(np.uint32)
        cq, j = blk // nbq, blk % nbq
        slot, bit = 4 * q + cq, 4 * nbq - 1 - (4 * j + r)
    for l in range(2):
        got = (m[l][:, slot] >> bit) & 1
        want = (acts_ref[l].numpy() > 0).astype(np.uint32)
        margin = np.abs(acts_ref[l].numpy())
        bad = (got != want) & (margin > 1e-6)         # (a pre-activation within rounding of zero may land on either side)
        assert not bad.any(), "layer %d: %d mask bits differ from the activations' signs" % (l, bad.sum())
    if keep_rows:
        for l in range(2):
            vec_close(acts[l].cpu(), acts_ref[l], TOL, "acts%d" % l)
        assert torch.isfinite(dz).all()
    else:
        assert torch.isnan(acts).all() and torch.isnan(dz).all(), "rows were written although acts / dz were not passed"


def test_mask_words_need_the_register_resident_kernels():
    """A net the register-resident kernels do not take (100 wide) reports so, and a launch that passes mask words for it
    fails with an error instead of running ungated."""
    from nlbac_amd import _lib, arena as A
    ar, h, lins, ref = build(3, 100, 3, 3, seed=1)
    nets, s = A.mlp_array([h.desc]), A.stream_ptr()
    assert _lib.load().nlbac_mlp_masks_ok(nets, 1) == 0
    B = 40
    x, y = torch.randn(B, 3, device="cuda"), torch.empty(B, 3, device="cuda")
    masks = torch.zeros(2, B, 8, dtype=torch.int32, device="cuda")
    io = A.io_array(1)
    io[0].x0, io[0].x0_dim, io[0].x0_ld = x.data_ptr(), 3, 3
    io[0].y, io[0].y_ld = y.data_ptr(), 3
    io[0].masks = masks.data_ptr()
    with pytest.raises(_lib.NlbacError):
        _lib.call("nlbac_mlp_fwd", nets, io, 1, B, s)


def test_weight_gradient_paths_can_alternate_on_one_gradient_buffer():
    """``nlbac_mlp_bwd_weights`` has two slab layouts (the one-launch kernel of narrow nets leaves a partial of every
    gradient in every slab; the other path leaves the bias / skinny-layer gradients in slab 0).  Every call defines all
    slabs' entries of its nets, so alternating the two on one ``grad`` buffer — stale NaNs in it to start with — gives
    the autograd gradient each time."""
    from nlbac_amd import _lib, arena as A
    in_dim, hid, out_dim, n_layers, B = 5, 96, 3, 4, 200
    ar, h, lins, ref = build(in_dim, hid, out_dim, n_layers, seed=3, n_slabs=5)
    nets, s = A.mlp_array([h.desc]), A.stream_ptr()
    g = torch.Generator().manual_seed(9)
    ar.grad.fill_(float("nan"))
    for it, fused in enumerate((False, True, False, True, False)):
        x, dy = torch.randn(B, in_dim, generator=g), torch.randn(B, out_dim, generator=g)
        _, _, _, grads_ref = torch_ref(ref, x, dy)
        xd, dyd = x.cuda(), dy.cuda()
        y = torch.empty(B, out_dim, device="cuda")
        acts, dz = torch.empty(n_layers - 1, B, hid, device="cuda"), torch.empty(n_layers - 1, B, hid, device="cuda")
        io = A.io_array(1)
        io[0].x0, io[0].x0_dim, io[0].x0_ld = xd.data_ptr(), in_dim, in_dim
        io[0].y, io[0].y_ld = y.data_ptr(), out_dim
        io[0].acts, io[0].dz = acts.data_ptr(), dz.data_ptr()
        io[0].dy, io[0].dy_ld = dyd.data_ptr(), out_dim
        io[0].grad = ar.grad.data_ptr()
        # fused: the data backward leaves the skinny partials -> the slab-0 layout; else the one-launch kernel (hid <= 112)
        ws = A.skinny_partials_ws(nets, (io,), 1, B, "cuda") if fused else None
        _lib.call("nlbac_mlp_fwd", nets, io, 1, B, s)
        _lib.call("nlbac_mlp_bwd_data", nets, io, 1, B, s)
        A.bwd_weights(nets, io, 1, B, ar.n_slabs, ar.n, "cuda", ws=ws)
        torch.cuda.synchronize()
        for lin in lins:        # (every slab's entries of every parameter; the arena's alignment padding stays as it was)
            for prm in (lin.weight, lin.bias):
                off = ar.offset_of[id(prm)]
                assert torch.isfinite(ar.grad[:, off:off + prm.numel()]).all(), "call %d left stale entries in some slab" % it
        for l, lin in enumerate(lins):
            vec_close(ar.grad_view(lin.weight).cpu(), grads_ref[l][0], TOL, "call %d dW%d" % (it, l))
            vec_close(ar.grad_view(lin.bias).cpu(), grads_ref[l][1], TOL, "call %d db%d" % (it, l))


@pytest.mark.parametrize("B,hid", [(4096, 256), (100, 256), (1000, 128)])
def test_layer0_dz_rows_are_not_stored_when_their_gradients_travel_as_partial_sums(B, hid):
    """``nlbac_mlp_io::dz_first = 1`` with ``skinny_ws`` (3-layer heads on the quarter-panel kernels): the data backward
    leaves layer 0's weight / bias gradients as partial sums and does not write its dz rows — the weight backward reads
    dz from layer 1 on — so every gradient is bit-identical to the run that stores them, and the rows stay untouched."""
    from nlbac_amd import _lib, arena as A
    in_dim, out_dim = 9, 1
    torch.manual_seed(B + hid)
    mods = [[nn.Linear(in_dim, hid), nn.Linear(hid, hid), nn.Linear(hid, out_dim)] for _ in range(2)]
    ar = A.Arena("cuda", n_slabs=4)
    hs = [A.MlpHandle(ar, [(l.weight, l.bias) for l in m]) for m in mods]
    ar.finalize()
    for h in hs:
        h.bind()
    A.pack(hs)
    nets = A.mlp_array([h.desc for h in hs])
    s = A.stream_ptr()
    keep, out = [], []
    for first in (0, 1):
        io = A.io_array(2)
        dzs = []
        for i in range(2):
            x = torch.randn(B, in_dim, generator=torch.Generator().manual_seed(10 + i)).cuda()
            dy = torch.randn(B, out_dim, generator=torch.Generator().manual_seed(20 + i)).cuda()
            y, acts = torch.empty(B, out_dim, device="cuda"), torch.empty(2, B, hid, device="cuda")
            dz = torch.full((2, B, hid), float("nan"), device="cuda")
            dx = torch.empty(B, in_dim, device="cuda")
            keep += [x, dy, y, acts, dz, dx]
            dzs.append(dz)
            io[i].x0, io[i].x0_dim, io[i].x0_ld = x.data_ptr(), in_dim, in_dim
            io[i].y, io[i].y_ld = y.data_ptr(), out_dim
            io[i].acts, io[i].dz = acts.data_ptr(), dz.data_ptr()
            io[i].dy, io[i].dy_ld = dy.data_ptr(), out_dim
            io[i].dx, io[i].dx_ld = dx.data_ptr(), in_dim
            io[i].grad = ar.grad.data_ptr()
        ws = A.skinny_partials_ws(nets, (io,), 2, B, "cuda")
        assert ws is not None
        for i in range(2):
            io[i].dz_first = first
        ar.grad.fill_(float("nan"))
        _lib.call("nlbac_mlp_fwd", nets, io, 2, B, s)
        _lib.call("nlbac_mlp_bwd_data", nets, io, 2, B, s)
        A.bwd_weights(nets, io, 2, B, ar.n_slabs, ar.n, "cuda", ws=ws)
        torch.cuda.synchronize()
        keep.append(ws)
        out.append((ar.grad.clone(), [d.clone() for d in dzs], dx.clone()))
    (g0, dz0, dx0), (g1, dz1, dx1) = out
    assert torch.equal(torch.nan_to_num(g0), torch.nan_to_num(g1)) and torch.equal(torch.isnan(g0), torch.isnan(g1))
    assert torch.equal(dx0, dx1)
    for a, b in zip(dz0, dz1):
        assert torch.isfinite(a).all() and torch.equal(a[1], b[1])
        assert torch.isnan(b[0]).all(), "layer 0's dz rows were written although dz_first = 1"
    # without the partial sums the weight backward needs those rows: it must refuse
    io[0].skinny_ws = None
    with pytest.raises(_lib.NlbacError):
        A.bwd_weights(nets, io, 2, B, ar.n_slabs, ar.n, "cuda")
