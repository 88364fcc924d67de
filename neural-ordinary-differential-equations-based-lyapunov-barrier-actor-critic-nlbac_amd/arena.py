"""Flat device buffers for parameters / gradients / Adam state, and the MLP
descriptors (``struct nlbac_mlp``) that point into them.

HBM layout (one ``Arena`` per optimiser group):

    theta   [n]                canonical fp32 parameters (torch.nn.Linear layout);
                               every ``nn.Parameter`` of the bound modules is a
                               *view* into it, so ``state_dict()`` keeps the
                               reference key names and checkpoints interoperate
    grad    [n_slabs][n]       per-row-range gradient slabs written by
                               ``nlbac_mlp_bwd_weights`` (deterministic), summed
                               in slab order inside ``nlbac_adam_step``
    m, v    [n]                Adam moments
    target  [n] (optional)     Polyak-averaged copy (critic/Lyapunov targets)
    state   16 B               {step, lr/(1-b1^t), sqrt(1-b2^t)} on device

Every tensor group starts on a 16-byte boundary (float4 loads in the kernels).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import MAX_LAYERS, Mlp, MlpIO


def _align4(n):
    return (n + 3) & ~3


class Arena:
    def __init__(self, device, n_slabs=8, with_target=False):
        self.device = torch.device(device)
        self.n_slabs = n_slabs
        self.with_target = with_target
        self._pending = []      # (param, offset)
        self.size = 0
        self.theta = None
        self.handles = []       # the MlpHandles whose parameters live here
        self._scatter = None

    def add_group(self, params):
        """Reserve contiguous space for a list of nn.Parameters (no padding
        between them); returns their offsets."""
        offs = []
        self.size = _align4(self.size)
        for p in params:
            offs.append(self.size)
            self._pending.append((p, self.size))
            self.size += p.numel()
        return offs

    def finalize(self):
        n = _align4(self.size)
        self.n = n
        dev = self.device
        self.theta = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_slabs, n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.target = torch.zeros(n, dtype=torch.float32, device=dev) if self.with_target else None
        self.state = torch.zeros(4, dtype=torch.int32, device=dev)   # AdamState
        self.offset_of = {}
        with torch.no_grad():
            for p, off in self._pending:
                view = self.theta[off:off + p.numel()].view(p.shape)
                view.copy_(p.data.to(dev))
                p.data = view
                self.offset_of[id(p)] = off
        return self

    def hard_update_target(self):
        self.target.copy_(self.theta)

    def scatter_tables(self):
        """(scatter, scatter_target) for ``nlbac_adam_fused``: per parameter the device addresses of its forward /
        backward MFMA-fragment slots in the nets' ``packed`` (``packed_target``) buffers, so the optimiser step
        refreshes those copies itself.  Found by packing an index ramp once (the fragment layout stays the business
        of ``nlbac_mlp_pack`` alone) and checked against a real pack."""
        if self._scatter is not None:
            return self._scatter
        import numpy as np
        hs = [h for h in self.handles if h.desc is not None]
        n = self.n
        assert hs and n < (1 << 24), "index ramp must stay exact in fp32"
        keep = self.theta.clone()
        self.theta.copy_(torch.arange(1, n + 1, dtype=torch.float32))
        pack(hs)
        torch.cuda.synchronize()
        src_all, addr_all, addr_t_all = [], [], []
        for h in hs:
            pk = h.packed.cpu().numpy()
            slots = np.nonzero(pk)[0]
            src_all.append(pk[slots].astype(np.int64) - 1)
            addr_all.append(np.uint64(h.packed.data_ptr()) + slots.astype(np.uint64) * np.uint64(4))
            if self.with_target:
                addr_t_all.append(np.uint64(h.packed_target.data_ptr()) + slots.astype(np.uint64) * np.uint64(4))
        src = np.concatenate(src_all)
        order = np.argsort(src, kind="stable")
        src = src[order]
        # k-th occurrence of each parameter among its fragment slots (forward / backward pack: <= 2; with RR packs: 4)
        start = np.ones(len(src), dtype=bool)
        start[1:] = src[1:] != src[:-1]
        first_idx = np.maximum.accumulate(np.where(start, np.arange(len(src)), 0))
        occ = np.arange(len(src)) - first_idx
        S = 2 if occ.max() < 2 else 4
        assert occ.max() < S, "a parameter has more than four fragment slots"

        def table(addrs):
            a = np.concatenate(addrs)[order]
            t = np.zeros((n, S), dtype=np.uint64)
            t[src, occ] = a
            return torch.from_numpy(t.view(np.int64).reshape(-1)).to(self.device)

        tab = table(addr_all)
        tab_t = table(addr_t_all) if self.with_target else None
        self.theta.copy_(keep)
        pack(hs)
        if self.with_target:
            pack(hs, target=True)
        # check: scattering theta through the table reproduces the packed buffers
        th = self.theta.cpu().numpy()
        t_np = tab.cpu().numpy().view(np.uint64).reshape(n, S)
        for h in hs:
            pk = h.packed.cpu().numpy()
            base = np.uint64(h.packed.data_ptr())
            chk = np.zeros_like(pk)
            for c in range(S):
                a = t_np[:, c]
                sel = (a >= base) & (a < base + np.uint64(4 * pk.size))
                chk[((a[sel] - base) // np.uint64(4)).astype(np.int64)] = th[sel]
            assert np.array_equal(chk, pk), "scatter table does not reproduce nlbac_mlp_pack for %s" % h.name
        self.scatter_slots = S
        self._scatter = (tab, tab_t)
        return self._scatter

    def grad_view(self, p, slab=None):
        off = self.offset_of[id(p)]
        g = self.grad[:, off:off + p.numel()]
        return g.sum(0).view(p.shape) if slab is None else g[slab].view(p.shape)


class MlpHandle:
    """One ReLU MLP inside an Arena: builds ``struct nlbac_mlp`` (for the live
    parameters and optionally for the target copy) and owns the packed weights."""

    def __init__(self, arena, layers, name=""):
        """layers: list of (weight_param(s), bias_param(s)); the last entry may
        hold lists (policy heads: [mean_W, logstd_W], [mean_b, logstd_b])."""
        self.arena, self.name = arena, name
        self.layers = layers
        flat = []
        for W, b in layers:
            Ws = list(W) if isinstance(W, (list, tuple)) else [W]
            bs = list(b) if isinstance(b, (list, tuple)) else [b]
            flat.append((Ws, bs))
        self._flat = flat
        self.n_layers = len(flat)
        assert 2 <= self.n_layers <= MAX_LAYERS
        W0 = flat[0][0][0]
        self.in_dim, self.hid = W0.shape[1], W0.shape[0]
        self.out_dim = sum(w.shape[0] for w in flat[-1][0])
        for Ws, bs in flat:               # contiguous groups: W's then b's
            arena.add_group(Ws)
            arena.add_group(bs)
        arena.handles.append(self)
        self.desc = None
        self.desc_target = None

    def bind(self):
        a = self.arena
        d = Mlp()
        d.n_layers, d.in_dim, d.hid, d.out_dim = self.n_layers, self.in_dim, self.hid, self.out_dim
        for l, (Ws, bs) in enumerate(self._flat):
            d.w_off[l] = a.offset_of[id(Ws[0])]
            d.b_off[l] = a.offset_of[id(bs[0])]
        a._scatter = None                 # (new packed buffers: the slot addresses change)
        lib = _lib.load()
        n_packed = lib.nlbac_mlp_pack_layout(C.byref(d))
        self.packed = torch.zeros(max(n_packed, 4), dtype=torch.float32, device=a.device)
        d.params = a.theta.data_ptr()
        d.packed = self.packed.data_ptr()
        self.desc = d
        if a.with_target:
            t = Mlp()
            C.memmove(C.byref(t), C.byref(d), C.sizeof(Mlp))
            self.packed_target = torch.zeros_like(self.packed)
            t.params = a.target.data_ptr()
            t.packed = self.packed_target.data_ptr()
            self.desc_target = t
        return self

    def params(self):
        out = []
        for Ws, bs in self._flat:
            out += Ws + bs
        return out


def mlp_array(descs):
    arr = (Mlp * len(descs))()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr, i * C.sizeof(Mlp)), C.byref(d), C.sizeof(Mlp))
    return arr


def io_array(n):
    return (MlpIO * n)()


def ptr(t):
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device, as an int for the C ABI.  (The raw accessor skips
    building a ``torch.cuda.Stream`` object per call — ~9 us each, a dozen calls per update.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def pack(handles, target=False):
    """Rewrite the MFMA-fragment copies of the nets' weights.  ``target``: False the trained nets, True their Polyak
    targets, "both" the two sets in one launch (up to NLBAC_MAX_NETS descriptors per launch)."""
    # descriptor arrays are cached on the first handle (they die with it; a global cache keyed by id()
    # would hand stale pointers to a new net that happens to reuse a freed object's id)
    cache = handles[0].__dict__.setdefault("_pack_cache", {})
    key = (tuple(id(h) for h in handles), target)
    arrs = cache.get(key)
    if arrs is None:
        if target == "both":
            descs = [h.desc for h in handles] + [h.desc_target for h in handles]
        else:
            descs = [h.desc_target if target else h.desc for h in handles]
        arrs = [(mlp_array(descs[i:i + _lib.MAX_NETS]), len(descs[i:i + _lib.MAX_NETS]))
                for i in range(0, len(descs), _lib.MAX_NETS)]
        cache[key] = (arrs, list(handles))       # keep the other handles alive as long as the entry exists
    else:
        arrs = arrs[0]
    s = stream_ptr()
    for arr, n in arrs:
        _lib.call("nlbac_mlp_pack", arr, n, s)


_BW_WS = {}
DZ_SKIP0 = __import__("os").environ.get("NLBAC_DZ_SKIP0", "1") != "0"


def skinny_partials_ws(nets, io_sets, n_nets, B, device):
    """A workspace of ``nlbac_mlp_bwd_weights`` of its own for these nets, with every net's block entered into the
    launch descriptors of ``io_sets`` (the data-backward's and the weight-backward's): ``nlbac_mlp_bwd_data`` then
    leaves the skinny-gradient partial sums there and ``nlbac_mlp_bwd_weights(..., ws=<this>)`` only reduces them.
    Returns None (separate partial pass) for batches / widths the fused form does not cover."""
    if B > 32768 or any(nets[i].hid > 256 for i in range(n_nets)):
        return None
    need = _lib.load().nlbac_mlp_bwd_weights_ws_floats(nets, n_nets, B)
    ws = torch.empty(max(need, 4), dtype=torch.float32, device=device)
    for i in range(n_nets):
        for io in io_sets:
            io[i].skinny_ws = ws.data_ptr() + 4 * i * (need // n_nets)
            io[i].dz_first = 1 if DZ_SKIP0 else 0      # (layer 0's gradients travel as those partial sums: its dz rows are not read)
    return ws


def bwd_weights(nets, io, n_nets, B, n_slabs, slab_stride, device, ws=None):
    """nlbac_mlp_bwd_weights with a cached per-device workspace for the skinny-gradient partials (or the caller's,
    see ``skinny_partials_ws``)."""
    lib = _lib.load()
    need = lib.nlbac_mlp_bwd_weights_ws_floats(nets, n_nets, B)
    key = str(device)
    if ws is None:
        ws = _BW_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=device)
        _BW_WS[key] = ws
    _lib.call("nlbac_mlp_bwd_weights", nets, io, n_nets, B, n_slabs, slab_stride, ws.data_ptr(), ws.numel(),
              stream_ptr())
