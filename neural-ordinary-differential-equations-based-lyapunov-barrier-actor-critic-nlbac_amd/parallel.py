"""Sample-sharded data parallelism for the update path.

One process per GPU, ``torch.distributed`` (backend ``nccl`` is RCCL over xGMI
on ROCm).  Every rank holds full replicas of all parameters / optimiser state
and its own shard of the minibatch; all couplings between samples are batch
sums, so the exchanges are fp32 SUM all-reduces of
  * the flat critic+Lyapunov gradient (+3 loss sums)    before its Adam step,
  * 20 constraint / actor partial sums                  before the augmented-Lagrangian scalars,
  * the flat actor gradient                             before its Adam step,
  * the NODE-fit gradient (+1 loss sum)                 every NODE_model_update_interval updates,
  * for dopri5 with step_control="global": 2 floats per problem per norm, so all ranks share one step size /
    accept decision (what the parity tests pin).  With step_control="shard" (bench.py's default for N > 1) every rank
    controls the steps of its own rows and NO collective runs inside a solve: the solve is the single-device one with
    ``row_groups = world`` (odeint.py), tests/test_data_parallel.py::test_two_rank_update_with_per_shard_step_control.
Messages are <= 1.4 MB (latency-bound): one flat buffer per phase, no bucketing.
The reference has no distributed path (its mpi4py helpers are dead code).
"""
import torch


class DataParallel:
    def __init__(self, dist, group=None, always_collective=False):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        # True: issue the collective even in a one-rank group (tests: the RCCL code path on a single GPU)
        self.always_collective = always_collective
        # what went over the wire: (calls, bytes) per collective kind, for tests and bench lines to count per update
        self.stats = {"all_reduce": [0, 0], "broadcast": [0, 0]}

    def all_reduce_(self, t):
        """In-place SUM over ranks.  gloo cannot reduce device tensors here: stage through the host
        (used by the CPU / single-GPU rehearsal tests only)."""
        if self.world == 1 and not self.always_collective:
            return t
        self.stats["all_reduce"][0] += 1
        self.stats["all_reduce"][1] += t.numel() * t.element_size()
        if t.is_cuda and self.backend == "gloo":
            h = t.detach().cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def broadcast_(self, t, src=0):
        if self.world == 1 and not self.always_collective:
            return t
        self.stats["broadcast"][0] += 1
        self.stats["broadcast"][1] += t.numel() * t.element_size()
        if t.is_cuda and self.backend == "gloo":
            h = t.detach().cpu()
            self.dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        else:
            self.dist.broadcast(t, src=src, group=self.group)
        return t

    def shard(self, n_rows):
        """Contiguous row range [lo, hi) of this rank for a global batch of n_rows."""
        per = (n_rows + self.world - 1) // self.world
        lo = min(n_rows, self.rank * per)
        return lo, min(n_rows, lo + per)
