"""CPU: the C-ABI library builds, loads and exports every symbol the header
declares; host-side mirrors (replay, dynamics glue, arena layout, model
containers) behave like the reference's.  No kernel is launched here."""
import collections
import os
import random
import re

import numpy as np
import pytest
import torch

import nlbac_amd  # noqa: F401
from nlbac_amd import _lib, synth
from nlbac_amd.envspec import make_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def header_functions():
    txt = open(os.path.join(ROOT, "include", "nlbac_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nlbac_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libnlbac_hip.so does not export %s" % n
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert lib.nlbac_abi_version() == _lib.ABI_VERSION == 16


def test_ctypes_structs_match_header_sizes(lib):
    # nlbac_mlp: 4 ints, ptr, 12 ints, ptr, 12 ints, 4 ints ; nlbac_mlp_io per header
    assert _lib.C.sizeof(_lib.Mlp) == 16 + 8 + 48 + 8 + 48 + 16
    io = _lib.MlpIO()
    assert _lib.C.sizeof(io) % 8 == 0
    net = _lib.Mlp()
    net.n_layers, net.in_dim, net.hid, net.out_dim = 3, 9, 256, 1
    n = lib.nlbac_mlp_pack_layout(_lib.C.byref(net))
    # ONE hid x hid layer of width 64 / 128 / 256, <= 15 inputs, <= 16 outputs: every launch of the net runs on the
    # register-resident kernels, so it has NO 32x32x2 packs (pf_off / pb_off = -1; the optimiser refreshes two fragment
    # slots per weight instead of four) — the two panel packs of the hid x hid layer (hid^2 floats each) and three
    # fragment blocks behind them: layer 0 forward, last layer and layer 0 transposed for the data backward
    assert n == 2 * 256 * 256 + 3 * 4 * 16 * 64
    assert all(net.pf_off[l] == -1 and net.pb_off[l] == -1 for l in range(3))
    assert net.rr_kind == 2 and net.rr_fwd_off == 0 and net.rr_bwd_off == 256 * 256 and net.packed_floats == n
    # a 3-layer net the register-resident kernels do not take (20 inputs) keeps the tile packs:
    # layer0 fwd: 8 tiles x 3 chunks ; layer1 fwd: 8 x 32 ; layer1 bwd: 8 x 32  (x256 floats), the panels behind them
    net = _lib.Mlp()
    net.n_layers, net.in_dim, net.hid, net.out_dim = 3, 20, 256, 1
    n = lib.nlbac_mlp_pack_layout(_lib.C.byref(net))
    old = (8 * 3 + 8 * 32 + 8 * 32) * 256
    assert net.pf_off[0] == 0 and net.pb_off[0] == -1 and net.pb_off[1] > net.pf_off[1] > 0
    assert net.rr_kind == 2 and net.rr_fwd_off == old and net.rr_bwd_off == old + 256 * 256 and net.packed_floats == n
    # a NODE-sized net (3 -> 100 -> 100 -> 100 -> 100 -> 3) also gets the RR packs of its three 100 x 100 layers:
    # 7 output blocks x 25 k-steps = 175 MFMAs = 44 float4 per lane and layer, forward and backward
    net = _lib.Mlp()
    net.n_layers, net.in_dim, net.hid, net.out_dim = 5, 3, 100, 3
    n = lib.nlbac_mlp_pack_layout(_lib.C.byref(net))
    old = (4 * 1 + 3 * (4 * 13 + 4 * 13)) * 256
    assert net.rr_kind == 1 and net.rr_fwd_off == old and net.rr_bwd_off == old + 3 * 44 * 256 and n == old + 6 * 44 * 256 == net.packed_floats


def test_integration_snippet_structs_match_the_compiled_header(tmp_path):
    """INTEGRATION.md shows a maintainer the ctypes structs to bind: their sizes must equal sizeof() of the C structs
    of include/nlbac_hip.h as gcc lays them out (a snippet one member short makes the library read past its end), and
    so must the build's own ctypes structs."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "nlbac_hip.h"\n'
                   'int main(void){printf("%zu %zu\\n", sizeof(nlbac_mlp), sizeof(nlbac_mlp_io));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    sz_mlp, sz_io = (int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert (C.sizeof(_lib.Mlp), C.sizeof(_lib.MlpIO)) == (sz_mlp, sz_io)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    classes = re.findall(r"^class (Mlp|MlpIO)\(C\.Structure\):.*?(?=^\S)", doc, flags=re.S | re.M)
    code = "".join(m.group(0) for m in re.finditer(r"^class (?:Mlp|MlpIO)\(C\.Structure\):.*?(?=^\S)", doc, flags=re.S | re.M))
    assert len(classes) == 2, "INTEGRATION.md no longer shows the two ctypes structs"
    ns = {"C": C}
    exec(code, ns)
    assert (C.sizeof(ns["Mlp"]), C.sizeof(ns["MlpIO"])) == (sz_mlp, sz_io), "INTEGRATION.md's structs drifted from the header"


def test_every_ctypes_struct_of_the_binding_matches_the_header(tmp_path):
    """The by-pointer argument structs of the C ABI, as ``_lib`` declares them to ctypes, against sizeof() and the offset
    of the last member in include/nlbac_hip.h (a field missing or out of order shifts everything behind it)."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("nlbac_mlp", _lib.Mlp, None), ("nlbac_mlp_io", _lib.MlpIO, "masks"),
             ("nlbac_auglag_args", _lib.AuglagArgs, "lam_hi"), ("nlbac_actor_scalar_args", _lib.ActorScalarArgs, "sc"),
             ("nlbac_rk_chain", _lib.RkChain, "ctl_host"), ("nlbac_in_map", _lib.InMap, "ps"),
             ("nlbac_out_map", _lib.OutMap, "x"), ("nlbac_gauss_head", _lib.GaussHead, "logp"),
             ("nlbac_dy_head", _lib.DyHead, "out")]
    body = "".join('printf("%%zu %%zu\\n", sizeof(%s), %s);' % (c, "offsetof(%s, %s)" % (c, last) if last else "(size_t)0")
                   for c, _, last in pairs)
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nlbac_hip.h"\nint main(void){%s return 0;}\n' % body)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    for k, (cname, cls, last) in enumerate(pairs):
        size, off = int(out[2 * k]), int(out[2 * k + 1])
        assert C.sizeof(cls) == size, "%s: ctypes %d bytes, header %d" % (cname, C.sizeof(cls), size)
        if last:
            assert getattr(cls, last).offset == off, "%s.%s: ctypes offset %d, header %d" % (cname, last, getattr(cls, last).offset, off)


def test_error_reporting_is_loud(lib):
    net = _lib.Mlp()
    net.n_layers, net.in_dim, net.hid, net.out_dim = 3, 99, 256, 1     # in_dim too large
    rc = lib.nlbac_mlp_pack(_lib.C.byref(net), 1, None)
    assert rc != 0 and b"in_dim" in lib.nlbac_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnlbac_hip.so")
    with pytest.raises(_lib.NlbacError, match="no CPU fallback"):
        _lib.load()


def test_agent_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle.nlbac_oracle import Args
    from nlbac_amd.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
    env = make_env("Unicycle", 0)
    with pytest.raises(RuntimeError, match="MI355X"):
        SAC_CBF_CLF(7, env.action_space, env, Args(cuda=True))
    with pytest.raises(RuntimeError, match="MI355X"):
        SAC_CBF_CLF(7, env.action_space, env, Args(cuda=False))


def test_replay_memory_matches_list_semantics():
    """Same seed -> same transitions as the reference's list + random.sample + np.stack."""
    from nlbac_amd.sac_cbf_clf.replay_memory import ReplayMemory
    tr = synth.unicycle_transitions(300, seed=3)
    mem = ReplayMemory(1000, seed=5, initial_rows=64)          # forces two grow steps
    ref = []
    for i in range(300):
        row = tuple(tr[f][i] for f in synth.FIELDS)
        mem.push(*row[:8], t=row[8], next_t=row[9])
        ref.append(row)
    assert len(mem) == 300 and mem.position == 300
    random.seed(11)
    got = mem.sample(32)
    random.seed(11)
    want = tuple(map(np.stack, zip(*random.sample(ref, 32))))
    assert len(got) == 10
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    # ring overwrite
    small = ReplayMemory(4, seed=0)
    for i in range(6):
        small.push(np.full(7, i), np.zeros(2), i, 0., np.zeros(2), np.zeros(2), np.zeros(7), 1.)
    assert len(small) == 4 and small.position == 2
    assert sorted(small.sample(4)[2].tolist()) == [2., 3., 4., 5.]


def test_dynamics_model_get_state_conventions():
    from nlbac_amd.sac_cbf_clf.dynamics import DynamicsModel
    from oracle.nlbac_oracle import Args
    env = make_env("Unicycle", 0)
    dyn = DynamicsModel(env, Args(cuda=False))
    tr = synth.unicycle_transitions(16, seed=2)
    st = dyn.get_state(tr["obs"])
    assert st.dtype == np.float64 and st.shape == (16, 3)
    np.testing.assert_allclose(st[:, 2], np.arctan2(tr["obs"][:, 3], tr["obs"][:, 2]))
    assert dyn.get_state(tr["obs"][0]).shape == (3,)
    t = dyn.get_state(torch.tensor(tr["obs"], dtype=torch.float32))
    assert torch.is_tensor(t) and t.dtype == torch.float32 and not t.requires_grad
    # tensor path = float32 obs -> float64 atan2 on the host -> cast back (dynamics.py:45-69)
    want = dyn.get_state(tr["obs"].astype(np.float32).astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(t.numpy(), want)


def test_arena_layout_and_state_dict_keys():
    """Parameters become views into one flat buffer; key names/shapes are the reference's."""
    from nlbac_amd.arena import Arena
    from nlbac_amd.sac_cbf_clf.model import GaussianPolicy, LyaNetwork, NeuralODEModel, QNetwork
    env = make_env("Unicycle", 0)
    torch.manual_seed(0)
    q, l = QNetwork(7, 2, 256), LyaNetwork(2, 256)
    p, node = GaussianPolicy(7, 2, 256, env.action_space), NeuralODEModel(3, 3, 6)
    before = {k: v.clone() for k, v in q.state_dict().items()}
    ar = Arena("cpu", n_slabs=2, with_target=True)
    hs = q.attach(ar) + l.attach(ar) + p.attach(ar) + node.attach(ar)
    ar.finalize()
    for h in hs:
        h.bind()
    W = synth.unicycle_agent_weights(256, 0)
    for mod, key in ((q, "critic"), (l, "lyapunov"), (p, "policy"), (node, "node")):
        assert list(mod.state_dict().keys()) == list(W[key].keys())
        for k, v in mod.state_dict().items():
            assert tuple(v.shape) == W[key][k].shape
    for k, v in q.state_dict().items():          # values survived the move into the arena
        assert torch.equal(v, before[k])
    base = ar.theta.data_ptr()
    for mod in (q, l, p, node):
        for prm in mod.parameters():
            off = (prm.data_ptr() - base) // 4
            assert 0 <= off and off + prm.numel() <= ar.n
    for h in hs:                                   # float4 alignment of every weight matrix
        for lidx in range(h.n_layers):
            assert h.desc.w_off[lidx] % 4 == 0
    # policy heads are one contiguous (2*n_u, hid) block / (2*n_u) bias block
    assert p.log_std_linear.weight.data_ptr() - p.mean_linear.weight.data_ptr() == 2 * 256 * 4
    assert p.log_std_linear.bias.data_ptr() - p.mean_linear.bias.data_ptr() == 2 * 4
    # in-place load keeps the storage
    q.load_state_dict({k: torch.from_numpy(v) for k, v in W["critic"].items()})
    assert q.linear1.weight.data_ptr() - base == ar.offset_of[id(q.linear1.weight)] * 4
    np.testing.assert_array_equal(q.linear5.weight.detach().numpy(), W["critic"]["linear5.weight"])


def test_model_init_follows_reference_rules():
    from nlbac_amd.sac_cbf_clf.model import LyaNetwork, NeuralODEModel
    torch.manual_seed(1)
    l = LyaNetwork(2, 256)
    assert float(l.linear2.bias.abs().max()) == 0.0
    bound = np.sqrt(6.0 / (256 + 256))
    assert float(l.linear2.weight.abs().max()) <= bound and float(l.linear2.weight.abs().max()) > 0.9 * bound
    n = NeuralODEModel(3, 3, 6)
    lin = n.f_net[2]
    assert float(lin.weight.abs().max()) <= 1 / np.sqrt(100) + 1e-7 and float(lin.bias.abs().max()) > 0
    assert sum(p.numel() for p in n.parameters()) == 52209


def test_oracle_dopri5_agrees_with_scipy_rk45():
    """Independent sanity of the from-memory dopri5 restatement: same field, tight-tolerance RK45."""
    from scipy.integrate import solve_ivp
    from oracle import nlbac_oracle as O
    W = {k: torch.from_numpy(v) for k, v in synth.unicycle_agent_weights(64, 0)["node"].items()}
    node = O.AffineNode(W)
    rs = np.random.RandomState(0)
    y0 = torch.tensor(np.concatenate([rs.uniform(-2, 2, (6, 3)), rs.uniform(-3, 3, (6, 2))], 1), dtype=torch.float32)
    info = {}
    with torch.no_grad():
        y = O.odeint(node, y0, torch.tensor([0., 0.02]), method="dopri5", atol=1e-7, rtol=1e-5, info=info)[-1]
        yr = O.odeint(node, y0, torch.tensor([0., 0.02]), method="rk4")[-1]
    for i in range(6):
        f = lambda t, s: node(t, torch.tensor(s[None], dtype=torch.float32)).double().numpy()[0]
        with torch.no_grad():
            sol = solve_ivp(f, (0, 0.02), y0[i].double().numpy(), method="RK45", rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(y[i].numpy(), sol.y[:, -1], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(yr[i].numpy(), sol.y[:, -1], rtol=2e-5, atol=2e-6)
    assert all(s[2] for s in info["steps"][-1:])


def test_reference_import_lines_resolve_to_this_build():
    """U/main.py:6-10 (and the same lines of the other drivers) after ``install_reference_names``."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import nlbac_amd\n"
        "nlbac_amd.install_reference_names(barrier=%s)\n"
        "from sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF\n"
        "from sac_cbf_clf.replay_memory import ReplayMemory\n"
        "from sac_cbf_clf.dynamics import DynamicsModel\n"
        "from sac_cbf_clf.utils import prGreen, get_output_folder, prYellow\n"
        "from sac_cbf_clf.model import BarrierNetwork, NeuralODEModel\n"
        "import inspect\n"
        "print(SAC_CBF_CLF.variant or 'base', len(inspect.signature(ReplayMemory.push).parameters))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for barrier, expect in ((False, "base 11"), (True, "Barrier 12")):
        out = subprocess.run([sys.executable, "-c", code % (root, barrier)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip() == expect


def test_dynamics_model_pvtol_returns_state_pair():
    from nlbac_amd.sac_cbf_clf.dynamics import DynamicsModel
    env = make_env("Pvtol", 0)
    dm = DynamicsModel(env, type("A", (), {"cuda": False})())
    tr = synth.transitions("Pvtol", 5, seed=2, env=env)
    st, dyn = dm.get_state(tr["obs"])
    assert st.shape == (5, 7) and dyn.shape == (5, 6)
    np.testing.assert_allclose(st[:, 6], tr["obs"][:, 7])
    np.testing.assert_allclose(np.cos(st[:, 2]), tr["obs"][:, 2], atol=1e-12)
    st_t, dyn_t = dm.get_state(torch.tensor(tr["obs"][0], dtype=torch.float32))
    assert st_t.shape == (7,) and dyn_t.shape == (6,) and st_t.dtype == torch.float32


def test_graft_entry_build_is_the_drivers_build_check():
    """``__graft_entry__.build()`` is what the driver runs as its "does it build" check: it must compile (incrementally
    here), load the library and agree with the header's ABI version (it once compared against a literal that the
    header had moved past)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_pmc_summary_differences_two_run_lengths(tmp_path):
    """tools/pmc_summary.py: per_update_bytes is the DIFFERENCE of two lean processes' counter totals over the difference
    of their update counts — what a process does once (zero fills at allocation) must cancel and show up as
    one_time_bytes; the read side is doubled (gfx950 tallies 128-byte requests at 64 bytes), KiB -> bytes."""
    import csv
    import json
    import subprocess
    import sys

    def write(d, counter, rows):
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "run_counter_collection.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for name, n, kib in rows:
                for _ in range(n):
                    w.writerow([name, counter, kib])

    k, fill = "void node_rr_fwd_kernel<7, 1, 1>(NodeRkLaunch)", "void at::native::fill_kernel<float>(float*)"
    # per update: 2 launches of k reading 100 KiB (tallied as 50) and writing 10 KiB each; once per process: a 4000 KiB fill
    for n_upd in (30, 50):
        write(str(tmp_path / ("f%d" % n_upd)), "FETCH_SIZE", [(k, 2 * n_upd, 50.0), (fill, 1, 0.0)])
        write(str(tmp_path / ("w%d" % n_upd)), "WRITE_SIZE", [(k, 2 * n_upd, 10.0), (fill, 1, 4000.0)])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path / "f50"), str(tmp_path / "w50"),
                        "50", str(tmp_path / "f30"), str(tmp_path / "w30"), "30"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    per_update = 2 * (2 * 50.0 + 10.0) * 1024
    assert d["updates"] == 20 and abs(d["per_update_bytes"] - per_update) < 1e-6
    assert abs(d["one_time_bytes"] - 4000.0 * 1024) < 1e-3
    assert abs(d["kernels"]["node_rr_fwd_kernel"]["hbm_bytes_per_launch"] - (2 * 50.0 + 10.0) * 1024) < 1e-6
    assert abs(d["per_update_by_kernel"]["node_rr_fwd_kernel"] - per_update) < 1e-6


def test_host_reader_of_the_stamped_control_block():
    """``AffineNodeSolver._ctl_poll`` (host side of nlbac_rk_chain::ctl_seq, no GPU needed): a problem's block is complete
    when it carries the stamp of the launch waited for — or an earlier stamp of the same solve with the done flag (launches
    skip finished problems); a block whose stamp is negative (a writer is in the middle of it), older than the solve or
    not yet there is waited for; torn reads (stamp changes under the copy) are retried."""
    import threading
    import time
    from nlbac_amd.odeint import AffineNodeSolver

    class Fake:
        stats = {}
        _ctl_poll = AffineNodeSolver._ctl_poll

    f = Fake()
    pin = torch.zeros(2, _lib.DOPRI_CTL, dtype=torch.float64)
    f._ctl_pin = {2: pin}
    arr = pin.numpy()
    # both problems carry the stamp waited for
    arr[:, 15] = 7.0
    arr[:, 0] = (0.25, 0.5)
    c = f._ctl_poll(2, 7, 7)
    assert c[0, 0] == 0.25 and c[1, 0] == 0.5 and c.data_ptr() != pin.data_ptr()
    # problem 1 finished two launches ago (done flag, stamp 5 of the same solve 4..7): accepted as it stands
    arr[1, 15], arr[1, 4] = 5.0, 1.0
    c = f._ctl_poll(2, 4, 7)
    assert c[1, 15] == 5.0 and c[1, 4] == 1.0
    # ... but not a done block of an EARLIER solve (stamp 3 < first 4), and not a block being written (negative stamp):
    # the reader waits until the writer has finished
    arr[1, 15] = 3.0

    def writer():
        time.sleep(0.01)
        arr[1, 15] = -7.0          # the launch's first store
        time.sleep(0.01)
        arr[1, 4], arr[1, 0] = 0.0, 0.125
        time.sleep(0.005)
        arr[1, 15] = 7.0

    t = threading.Thread(target=writer)
    t0 = time.perf_counter()
    t.start()
    c = f._ctl_poll(2, 4, 7, patience=5.0)
    t.join()
    assert time.perf_counter() - t0 >= 0.02
    assert c[1, 15] == 7.0 and c[1, 0] == 0.125 and c[1, 4] == 0.0
