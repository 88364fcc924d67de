"""GPU: ``bench.py`` itself, as the driver launches it — one process, and two ranks through ``torch.distributed.run``
(gloo, both ranks on the test box's one card: a rehearsal of the N > 1 line, not a scaling number) — must finish and
print ONE JSON line with the contract's keys.  Every rank has to go through the same barriers whatever extra passes rank
0 runs (the event-timed roofline pass): a rank-0-only barrier hangs the N > 1 run, which only shows here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(cmd, env=None, timeout=420):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_one_process_line_has_the_contract_keys():
    d = run([sys.executable, "bench.py", "--steps", "20", "--warmup", "10", "--profile-steps", "10", "--no-cpu-baseline"])
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 10 and d["dtype"] == "f32" and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    u = r["update"]
    # (round 4: the update's roofline is taken over the timed region itself — its 20 solves —, the event-timed pass
    #  that follows the region rides along as `event_pass`)
    assert 0 < u["frac"] < 1 and u["solver_stats"]["solves"] == 20 and "window" in u
    assert abs(u["frac"] - u["executed_mfma_kernel_flop_per_update"] / (d["ms_per_step"] * 1e-3) / 1e12 / r["peak"]) < 1e-9
    e = u["event_pass"]
    assert 0 < e["frac"] <= u["frac"] * 1.05 and e["solver_stats"]["solves"] == 10 and "window" in e
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]


@pytest.mark.parametrize("mode", ["shard", "global"])
def test_two_rank_rehearsal_finishes(mode):
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "10",
             "--profile-steps", "10", "--no-cpu-baseline", "--dp-step-control", mode],
            env={"NLBAC_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["dp_step_control"] == mode
    assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"] and d["value"] > 0
    assert d["strong"]["scaling"] == "strong" and d["strong"]["global_batch"] == d["config"]["batch_per_gpu"]
    o = d["other_step_control"]          # the same line under the other dopri5 step control
    assert o["dp_step_control"] == ("global" if mode == "shard" else "shard") and o["value"] > 0
    assert o["matches_single_device_decisions"] == (mode == "shard")
