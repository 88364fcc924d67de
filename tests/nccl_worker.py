"""One rank, backend ``nccl`` (= RCCL on ROCm), initialised BEFORE any other GPU call: device tensors through
``nlbac_amd.parallel.DataParallel.all_reduce_ / broadcast_`` with the one-rank short-cut switched off, so that the
device-tensor branch of the exchange layer — the one ``bench.py --gpus N`` runs on a multi-GPU node — has executed.
Launched by tests/test_data_parallel.py."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

if __name__ == "__main__":
    port = sys.argv[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    import nlbac_amd  # noqa: E402,F401
    from nlbac_amd.parallel import DataParallel  # noqa: E402
    dp = DataParallel(dist, always_collective=True)
    assert dp.backend == "nccl" and dp.world == 1
    t = torch.arange(1 << 16, dtype=torch.float32, device="cuda") * 0.5
    ref = t.clone()
    dp.all_reduce_(t)                    # SUM over one rank: unchanged, but the collective ran on the device buffer
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    small = torch.tensor([1.0, 2.0, 3.0, 4.0], device="cuda")      # the size of the dopri5 norm exchanges
    dp.all_reduce_(small)
    dp.broadcast_(t, src=0)
    torch.cuda.synchronize()
    assert torch.equal(t, ref) and small.tolist() == [1.0, 2.0, 3.0, 4.0]
    # one data-parallel update of the product agent on this group: every exchange of the update goes through RCCL
    from nlbac_amd import synth
    from test_agent_parity_gpu import make_agent
    from common import case_inputs, load_golden
    g = load_golden("dopri5", 128, "Unicycle")
    agent, env = make_agent(128, int(g["meta_hidden"]), int(g["meta_seed"]), "dopri5", "Unicycle", 50.0)
    agent.enable_data_parallel(dist, always_collective=True)     # parameter / scalars broadcasts over RCCL
    tr = synth.transitions("Unicycle", 4096, seed=int(g["meta_seed"]) + 1, env=env)
    batch, eps, node, updates = case_inputs(g, 0, tr)
    agent.set_noise(eps)
    ret = agent.update_from_host(tuple(batch[f].numpy() for f in synth.fields("Unicycle")), updates,
                                 tuple(x.numpy() for x in node) if updates % 10 == 0 else None)
    torch.cuda.synchronize()
    import numpy as np
    assert np.allclose(np.array(ret), g["c0_ret"], rtol=1e-4, atol=1e-6), (ret, g["c0_ret"])
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_OK")
