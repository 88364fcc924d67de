#!/bin/bash
# Round-2 measurement set on the GPU box (everything lands under gpurun_out/r02/; the summaries that are kept go to
# profiles/r02_*):  bash tools/gpu_r02_measure.sh
set -e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
echo "== headline"; python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
echo "== variants"
python bench.py --solver euler --steps 100 > $O/bench_euler.json 2> $O/bench_euler.err
python bench.py --solver rk4 --steps 100 > $O/bench_rk4.json 2> $O/bench_rk4.err
python bench.py --env SimulatedCars --batch 8192 --solver rk4 --steps 100 > $O/bench_cars.json 2> $O/bench_cars.err
python bench.py --env Pvtol --batch 16384 --steps 60 > $O/bench_pvtol.json 2> $O/bench_pvtol.err
python bench.py --env Pvtol --batch 16384 --adjoint --steps 60 > $O/bench_pvtol_adjoint.json 2> $O/bench_pvtol_adjoint.err
python bench.py --env UnicycleBarrier --batch 32768 --steps 60 > $O/bench_nbc_unicycle.json 2> $O/bench_nbc_unicycle.err
python bench.py --env QuadrotorLike --batch 32768 --steps 60 > $O/bench_quadrotorlike.json 2> $O/bench_quadrotorlike.err
echo "== 2 ranks sharing the card (gloo rehearsal of the N>1 line: weak + strong)"
NLBAC_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || tail -5 $O/bench_2rank_gloo.err
echo "== kernel stats"
bash tools/gpu_prof.sh r02_headline --steps 200 --warmup 20 > $O/prof_headline.log 2>&1
python tools/update_table.py gpurun_out/prof_r02_headline/run_kernel_trace.csv 68 77 > $O/update_table_single_step.txt 2>&1 || true
python tools/gap_report.py gpurun_out/prof_r02_headline/run_kernel_trace.csv 68 77 > $O/gap_report.txt 2>&1 || true
bash tools/gpu_prof.sh r02_pvtol --env Pvtol --batch 16384 --steps 40 --warmup 10 > $O/prof_pvtol.log 2>&1
bash tools/gpu_prof.sh r02_pvtol_adjoint --env Pvtol --batch 16384 --adjoint --steps 40 --warmup 10 > $O/prof_pvtol_adjoint.log 2>&1
echo done
