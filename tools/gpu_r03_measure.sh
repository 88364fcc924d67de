#!/bin/bash
# Round-3 measurement set on the GPU box (everything lands under gpurun_out/r03/; the summaries that are kept go to
# profiles/r03_*):  bash tools/gpu_r03_measure.sh <part>      part = a | b | c  (each fits one gpurun call)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-a}
if [ $PART = a ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== headline"; python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
echo "== kernel stats of the same command"
bash tools/gpu_prof.sh r03_headline --steps 200 --warmup 20 > $O/prof_headline.log 2>&1
python tools/update_table.py gpurun_out/prof_r03_headline/run_kernel_trace.csv 68 77 > $O/update_table_single_step.txt 2>&1 || true
python tools/gap_report.py gpurun_out/prof_r03_headline/run_kernel_trace.csv 68 77 > $O/gap_report.txt 2>&1 || true
echo "== variants"
python bench.py --solver euler --steps 100 --no-cpu-baseline > $O/bench_euler.json 2> $O/bench_euler.err
python bench.py --solver rk4 --steps 100 --no-cpu-baseline > $O/bench_rk4.json 2> $O/bench_rk4.err
python bench.py --env SimulatedCars --batch 8192 --solver rk4 --steps 100 --no-cpu-baseline > $O/bench_cars.json 2> $O/bench_cars.err
python bench.py --env UnicycleBarrier --batch 32768 --steps 60 --no-cpu-baseline > $O/bench_nbc_unicycle.json 2> $O/bench_nbc_unicycle.err
python bench.py --env QuadrotorLike --batch 32768 --steps 60 --no-cpu-baseline > $O/bench_quadrotorlike.json 2> $O/bench_quadrotorlike.err
fi
if [ $PART = b ]; then
echo "== Pvtol B=16384 direct / adjoint"
python bench.py --env Pvtol --batch 16384 --steps 60 --no-cpu-baseline > $O/bench_pvtol.json 2> $O/bench_pvtol.err
python bench.py --env Pvtol --batch 16384 --adjoint --steps 60 --no-cpu-baseline > $O/bench_pvtol_adjoint.json 2> $O/bench_pvtol_adjoint.err
fi
if [ $PART = b2 ]; then
echo "== 2 ranks sharing the card (gloo rehearsal of the N>1 line), per-shard and all-reduced step control"
for M in shard global; do
NLBAC_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline --dp-step-control $M > $O/bench_2rank_gloo_$M.json 2> $O/bench_2rank_gloo_$M.err || tail -5 $O/bench_2rank_gloo_$M.err
done
echo "== micro-benchmarks"
python tools/fit_span.py 32768 51 > $O/fit_span.txt 2>&1
python tools/microbench_dw.py > $O/microbench_dw.txt 2>&1
NLBAC_MLP_DW16=0 python tools/microbench_dw.py >> $O/microbench_dw.txt 2>&1
python tools/microbench_node.py > $O/microbench_node_fwd.txt 2>&1
python tools/microbench_node_bwd.py > $O/microbench_node_bwd.txt 2>&1
python tools/microbench_mlp.py > $O/microbench_mlp.txt 2>&1
fi
if [ $PART = c ]; then
bash tools/gpu_launch_table.sh r03
cp gpurun_out/r03_launches.txt $O/launches_per_update.txt
echo "== calibration: a register-resident layer chain alone on the chip"
mkdir -p tools/micro/bin
hipcc --offload-arch=gfx950 -O3 tools/micro/rr_chain.hip -o tools/micro/bin/rr_chain && tools/micro/bin/rr_chain > $O/calibration_rr_chain.txt 2>&1
echo "== shader-clock stamps of the fused RK forward (the -DRR_TIMING build made in the build container)"
V=$(echo neural-*-nlbac_amd/lib/variants/libnlbac_hip_rrtiming.so)
NLBAC_HIP_LIB=$V python tools/phase_times_rr.py 8192 1 > $O/phase_times_node_rr_fwd.txt 2>&1
NLBAC_HIP_LIB=$V python tools/phase_times_rr.py 32768 0 >> $O/phase_times_node_rr_fwd.txt 2>&1
fi
echo done
