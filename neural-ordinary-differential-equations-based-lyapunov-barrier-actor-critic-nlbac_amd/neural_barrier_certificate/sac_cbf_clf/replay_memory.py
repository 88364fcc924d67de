"""Replay buffer of the learned-barrier copies: 11 fields, the barrier signal after the constraint
(NU/sac_cbf_clf/replay_memory.py:13-25)."""
from ...sac_cbf_clf.replay_memory import DeviceReplayMemory  # noqa: F401  (takes the 11 fields as is)
from ...sac_cbf_clf.replay_memory import ReplayMemory as _Base


class ReplayMemory(_Base):

    def push(self, state, action, reward, constraint, barrier_signal, center_pos, next_center_pos, next_state, mask,
             t=None, next_t=None):
        self._push((state, action, reward, constraint, barrier_signal, center_pos, next_center_pos, next_state, mask,
                    t, next_t))
