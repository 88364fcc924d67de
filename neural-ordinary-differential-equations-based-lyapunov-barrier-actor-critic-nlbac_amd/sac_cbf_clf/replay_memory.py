"""Replay buffer with the reference API (``push`` / ``sample`` / ``len`` /
``position``; ``U/sac_cbf_clf/replay_memory.py``) stored as preallocated
float64 numpy columns (struct of arrays) instead of a Python list of tuples:
``sample`` is one fancy-index gather per field rather than ``np.stack`` over
``batch_size`` tuples.  Index draws use ``random.sample`` like the reference,
so the same seed selects the same transitions.
"""
import random

import numpy as np

_FIELDS = ("state", "action", "reward", "constraint", "center_pos", "next_center_pos",
           "next_state", "mask", "t", "next_t")


class ReplayMemory:

    def __init__(self, capacity, seed, initial_rows=65536):
        random.seed(seed)
        self.capacity = int(capacity)
        self._rows = min(self.capacity, int(initial_rows))
        self._cols = None
        self._len = 0
        self.position = 0

    def _alloc(self, values):
        self._cols = []
        for v in values:
            a = np.asarray(0.0 if v is None else v, dtype=np.float64)
            self._cols.append(np.zeros((self._rows,) + a.shape, dtype=np.float64))

    def _grow(self):
        new_rows = min(self.capacity, self._rows * 2)
        self._cols = [np.concatenate([c, np.zeros((new_rows - self._rows,) + c.shape[1:])]) for c in self._cols]
        self._rows = new_rows

    def push(self, state, action, reward, constraint, center_pos, next_center_pos, next_state, mask,
             t=None, next_t=None):
        self._push((state, action, reward, constraint, center_pos, next_center_pos, next_state, mask, t, next_t))

    def _push(self, values):
        if self._cols is None:
            self._alloc(values)
        if self.position >= self._rows:
            self._grow()
        for c, v in zip(self._cols, values):
            c[self.position] = 0.0 if v is None else v
        self._len = max(self._len, self.position + 1)
        self.position = (self.position + 1) % self.capacity

    def sample(self, batch_size):
        idx = np.asarray(random.sample(range(self._len), batch_size), dtype=np.int64)
        return tuple(c[idx] for c in self._cols)

    def __len__(self):
        return self._len



class DeviceReplayMemory:
    """Replay buffer resident in HBM in the agent's minibatch row layout.

    Same ``push`` / ``sample`` / ``len`` / ``position`` API as ``ReplayMemory``; rows are staged in a pinned host
    block and uploaded in chunks, and ``sample_rows`` gathers a minibatch on the device straight into the agent's
    workspace (one ``nlbac_gather_rows`` launch) — ``SAC_CBF_CLF.update_parameters`` takes that path, so an update
    moves ``8 * batch`` index bytes over PCIe instead of the whole minibatch.  Index draws use ``random.sample`` on
    the host exactly like the reference (``replay_memory.py:22``), so the same seed selects the same transitions;
    ``device_rng=True`` draws them on the device instead (with replacement, no host involvement): index draw, gather
    and — when the caller hands in the agent's noise buffer (``eps_out``) — the update's N(0,1) policy noise are one
    ``nlbac_sample_rows`` launch.
    """

    def __init__(self, capacity, seed, agent, chunk=4096, device_rng=False):
        import torch
        random.seed(seed)
        self.capacity, self.agent, self.device_rng = int(capacity), agent, device_rng
        self.lay, self.device = agent.lay, agent.device
        self.n_fields = 11 if self.lay.sig is not None else 10
        self.rows = torch.zeros(self.capacity, self.lay.LD, dtype=torch.float32, device=self.device)
        self._stage = torch.zeros(min(chunk, self.capacity), self.lay.LD, dtype=torch.float32).pin_memory() \
            if torch.cuda.is_available() else torch.zeros(min(chunk, self.capacity), self.lay.LD)
        self._stage_np = self._stage.numpy()
        self._n_staged, self._stage_start = 0, 0
        self._len = 0
        self.position = 0
        self._seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF   # Philox key
        self._draws = 0                                                                           # Philox counter

    def push(self, *fields, t=None, next_t=None):
        """Positional fields in the order of the reference's ``push`` (10, or 11 with the barrier signal)."""
        fields = tuple(fields)
        if len(fields) == self.n_fields - 2:
            fields = fields + (t, next_t)
        assert len(fields) == self.n_fields, "push takes %d fields" % self.n_fields
        if self._n_staged == self._stage_np.shape[0] or \
                (self._n_staged and self._stage_start + self._n_staged != self.position):
            self.flush()
        if self._n_staged == 0:
            self._stage_start = self.position
        batch1 = tuple(np.asarray(0.0 if f is None else f, dtype=np.float64)[None] for f in fields)
        self._stage_np[self._n_staged] = self.agent._rows_from_host(batch1).numpy()[0]
        self._n_staged += 1
        self._len = max(self._len, self.position + 1)
        self.position = (self.position + 1) % self.capacity
        if self.position == 0:
            self.flush()                       # keep a staged run contiguous in the ring

    def push_rows(self, rows):
        """Bulk insert of minibatch-layout rows (host or device tensor, (n, LD))."""
        import torch
        self.flush()
        rows = torch.as_tensor(rows, dtype=torch.float32)
        n = rows.shape[0]
        assert rows.shape[1] == self.lay.LD and n <= self.capacity
        first = min(n, self.capacity - self.position)
        self.rows[self.position:self.position + first].copy_(rows[:first])
        if n > first:
            self.rows[:n - first].copy_(rows[first:])
        self._len = min(self.capacity, max(self._len, self.position + n))
        self.position = (self.position + n) % self.capacity

    def flush(self):
        if self._n_staged:
            self.rows[self._stage_start:self._stage_start + self._n_staged].copy_(self._stage[:self._n_staged],
                                                                                 non_blocking=False)
            self._n_staged = 0

    def sample_rows(self, batch_size, out=None, eps_out=None):
        """Minibatch-layout rows (batch_size, LD) on the device.  ``eps_out``: a contiguous fp32 device buffer to fill
        with N(0,1) draws alongside (``SAC_CBF_CLF.update_on_device(..., eps_ready=True)`` then skips its own draw)."""
        import torch
        from .. import _lib
        from ..arena import stream_ptr
        self.flush()
        if out is None:
            out = torch.empty(batch_size, self.lay.LD, dtype=torch.float32, device=self.device)
        if self.device_rng:
            self._draws += 1
            _lib.call("nlbac_sample_rows", self.rows.data_ptr(), self._len, self.lay.LD, batch_size, out.data_ptr(),
                      eps_out.data_ptr() if eps_out is not None else None,
                      eps_out.numel() if eps_out is not None else 0, self._seed, self._draws, stream_ptr())
            return out
        idx = torch.tensor(random.sample(range(self._len), batch_size), dtype=torch.int64).to(self.device)
        _lib.call("nlbac_gather_rows", self.rows.data_ptr(), self._len, self.lay.LD, idx.data_ptr(), batch_size,
                  out.data_ptr(), stream_ptr())
        if eps_out is not None:
            eps_out.normal_()
        return out

    def sample(self, batch_size):
        """Reference-shaped ``sample``: the 10 (11) stacked numpy arrays (device -> host; use ``sample_rows`` on the
        update path)."""
        rows = self.sample_rows(batch_size).cpu().numpy().astype(np.float64)
        lay = self.lay
        cols = [("obs", lay.obs_dim), ("act", lay.act_dim), ("rew", 0), ("con", 0)]
        if lay.sig is not None:
            cols.append(("sig", 0))
        cols += [("lya", lay.lya_dim), ("nlya", lay.lya_dim), ("nobs", lay.obs_dim), ("mask", 0), ("t", 0), ("nt", 0)]
        out = []
        for name, w in cols:
            c = getattr(lay, name)
            out.append(rows[:, c] if w == 0 else rows[:, c:c + w])
        return tuple(out)

    def __len__(self):
        return self._len
