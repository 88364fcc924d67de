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
        idx = np.asarray(random.sample(range(self._len), batch_size))
        return tuple(c[idx] for c in self._cols)

    def __len__(self):
        return self._len
