"""Learned-barrier-certificate copies of the agent (reference directory ``neural_barrier_certificate/``): a
BarrierNetwork trained with the critics replaces the hand-written CBFs, there is no backup controller and the
replay rows carry a barrier signal."""
