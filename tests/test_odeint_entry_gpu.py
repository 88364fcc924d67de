"""GPU: the torchdiffeq-shaped entry points of SURVEY.md §8(b) — ``odeint(func, y0, t, method=, atol=, rtol=)``,
``NeuralODEModel.forward(t, s)`` and ``train_step(...)`` with the reference's positional signatures — against the CPU
oracle (``oracle.nlbac_oracle.odeint`` on the same weights, plain autograd).  Tolerance 1e-4 relative (fp32 bar)."""
import numpy as np
import pytest
import torch

from oracle import nlbac_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def close(a, b, what, tol=TOL):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))
    assert err < tol, "%s: relative error %.3g" % (what, err)


def make(kind, seed=0):
    from nlbac_amd.sac_cbf_clf.model import NeuralODEModel
    torch.manual_seed(seed)
    if kind == "affine":
        m = NeuralODEModel(3, 3, 6)
        sd = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
        return m, O.AffineNode(sd), sd, 3, 2
    m = NeuralODEModel(12, 10)
    sd = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    return m, O.ConcatNode(sd), sd, 10, 2


@pytest.mark.parametrize("kind", ["affine", "concat"])
@pytest.mark.parametrize("method", ["euler", "rk4", "dopri5"])
def test_odeint_matches_oracle_with_gradients(kind, method):
    from nlbac_amd.odeint import odeint
    m, ref, sd, ns, nc = make(kind)
    g = torch.Generator().manual_seed(3)
    B = 96
    y0 = torch.cat([torch.rand(B, ns, generator=g) * 2 - 1, torch.rand(B, nc, generator=g) * 2 - 1], 1)
    w = torch.randn(2, B, ns + nc, generator=g)
    t = torch.tensor([0.0, 0.05])
    # oracle
    y0r = y0.clone().requires_grad_()
    out_r = O.odeint(ref, y0r, t, method=method)
    (out_r * w).sum().backward()
    # device
    y0d = y0.cuda().requires_grad_()
    out_d = odeint(m, y0d, t.cuda(), method=method, atol=1e-7, rtol=1e-5)
    assert out_d.shape == (2, B, ns + nc)
    (out_d * w.cuda()).sum().backward()
    close(out_d, out_r, "y(t)")
    close(y0d.grad, y0r.grad, "d/dy0")
    for k, p in m.named_parameters():
        if method == "dopri5":
            # 7 stages x 4 ReLU layers: a unit whose pre-activation sits within rounding of zero in one stage takes the
            # other branch on the device than in the oracle for that row (tests/test_solver_gpu.py counts such rows);
            # a weight gradient sums all rows, so it is compared in norm
            a, b = p.grad.detach().cpu().double(), sd[k].grad.double()
            err = float((a - b).norm()) / max(1e-9, float(b.norm()))
            assert err < 1e-3, "d/d%s: relative error %.3g (norm)" % (k, err)
        else:
            close(p.grad, sd[k].grad, "d/d" + k, tol=2e-4)
    # the bare field evaluation, reference-shaped
    close(m.forward(0.0, y0.cuda()), ref(0.0, y0), "forward(t, s)")


def test_train_step_reference_signature():
    """Three ``train_step`` calls with a torch optimizer (the reference's usage, U/sac_cbf_clf/sac_cbf_clf.py:205-219)
    follow the oracle's losses: the parameter gradients, the in-place optimizer step on the arena views and the
    re-packing of the device weights all have to be right for losses 2 and 3 to agree."""
    from nlbac_amd.sac_cbf_clf.model import train_step
    m, ref, sd, ns, nc = make("affine", seed=1)
    g = torch.Generator().manual_seed(5)
    N = 256
    st, ac = torch.rand(N, 3, generator=g) * 2 - 1, torch.rand(N, 2, generator=g) * 2 - 1
    nx = st + 0.02 * torch.randn(N, 3, generator=g)
    opt_d = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt_r = torch.optim.Adam(list(sd.values()), lr=1e-3)
    loss_fn = torch.nn.MSELoss()
    for it in range(3):
        ld = train_step(m, st.cuda(), ac.cuda(), nx.cuda(), opt_d, loss_fn, 1, "dopri5", 0.02)
        opt_r.zero_grad()
        pred = O.odeint(ref, torch.cat([st, ac], 1), torch.tensor([0.0, 0.02]), method="dopri5")[-1][:, :3]
        lr_ = loss_fn(pred, nx)
        lr_.backward()
        opt_r.step()
        assert abs(ld - float(lr_.detach())) < 1e-4 * max(1e-6, abs(float(lr_.detach()))) + 1e-9, (it, ld, float(lr_.detach()))


def test_train_step_cars_signature_with_time_batch():
    from nlbac_amd.sac_cbf_clf.model import train_step
    m, ref, sd, ns, nc = make("concat", seed=2)
    g = torch.Generator().manual_seed(6)
    N = 128
    st, ac, tb = torch.rand(N, 10, generator=g), torch.rand(N, 1, generator=g) * 2 - 1, torch.rand(N, 1, generator=g)
    nx = st + 0.02 * torch.randn(N, 10, generator=g)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ld = train_step(m, st.cuda(), ac.cuda(), nx.cuda(), tb.cuda(), opt, torch.nn.MSELoss(), 1, "rk4", 0.02)
    pred = O.odeint(ref, torch.cat([st, ac, tb], 1), torch.tensor([0.0, 0.02]), method="rk4")[-1][:, :10]
    lr_ = float(torch.nn.functional.mse_loss(pred, nx))
    assert abs(ld - lr_) < 1e-4 * abs(lr_) + 1e-9


def test_odeint_rejects_foreign_modules_and_grids():
    from nlbac_amd.odeint import odeint
    from nlbac_amd.sac_cbf_clf.model import NeuralODEModel
    with pytest.raises(TypeError):
        odeint(torch.nn.Linear(5, 5), torch.zeros(4, 5).cuda(), torch.tensor([0.0, 0.1]), method="euler")
    m = NeuralODEModel(3, 3, 6)
    with pytest.raises(NotImplementedError):
        odeint(m, torch.zeros(4, 5).cuda(), torch.tensor([0.0, 0.1, 0.2]), method="euler")


@pytest.mark.parametrize("hid", [64, 160, 232])
@pytest.mark.parametrize("masks", [False, True], ids=["acts", "relu-bit-masks"])
def test_other_node_widths_run_the_other_kernel_modes(hid, masks):
    """The fused RK kernels pick a tiling mode by hidden width (<= 128: one column tile per wave; 225-256: two; anything
    else: mixed).  The reference's NODE is 100 wide; 64 / 160 / 232 put the other instantiations — and their
    per-group barriers — under the same oracle check, with activations kept (parameter gradients) and with bit masks
    (gradient to the controls only)."""
    from nlbac_amd.odeint import odeint, _solver_of
    from nlbac_amd.sac_cbf_clf.model import NeuralODEModel
    torch.manual_seed(hid)
    m = NeuralODEModel(3, 3, 6, hidden_dim=hid)
    sd = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    ref = O.AffineNode(sd)
    g = torch.Generator().manual_seed(4)
    B = 200
    y0 = torch.cat([torch.rand(B, 3, generator=g) * 2 - 1, torch.rand(B, 2, generator=g) * 2 - 1], 1)
    w = torch.randn(B, 3, generator=g)
    t = torch.tensor([0.0, 0.05])
    y0r = y0.clone().requires_grad_()
    out_r = O.odeint(ref, y0r, t, method="dopri5")[-1][:, :3]
    (out_r * w).sum().backward()
    if not masks:
        y0d = y0.cuda().requires_grad_()
        out_d = odeint(m, y0d, t, method="dopri5")[-1][:, :3]
        (out_d * w.cuda()).sum().backward()
        close(out_d, out_r, "x(dt)")
        close(y0d.grad, y0r.grad, "d/dy0")
        for k, p in m.named_parameters():
            a, b = p.grad.detach().cpu().double(), sd[k].grad.double()
            assert float((a - b).norm()) / max(1e-9, float(b.norm())) < 1e-3, k
    else:
        sv = _solver_of(m)
        sv.keep_acts = False
        sv._ws.clear()                       # (workspaces are laid out for the mode they were created in)
        m.refresh_device_weights()
        yd = y0.cuda()
        x1 = sv.forward(yd[:, :3].contiguous(), yd[:, 3:].contiguous(), 1, B, "dopri5", 0.05, 1e-7, 1e-5)
        close(x1, out_r, "x(dt)")
        du, dy0 = sv.backward(w.cuda(), need_du=True, need_dy0=True)
        close(torch.cat([dy0, du], 1), y0r.grad, "d/dy0")
