#!/bin/bash
# Round-4 measurement set on the GPU box (everything lands under gpurun_out/r04/; the summaries that are kept go to
# profiles/r04_*):  bash tools/gpu_r04_measure.sh <part>      part = a | b | c | d  (each fits one gpurun call)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-a}
if [ $PART = a ]; then
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; grep smoke $O/smoke.log | tail -8
echo "== headline (defaults) and the driver's form (20 steps)"
python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
python bench.py --steps 20 --no-cpu-baseline > $O/bench_20steps.json 2> $O/bench_20steps.err
echo "== kernel stats of the same command"
bash tools/gpu_prof.sh r04_headline --steps 200 --warmup 20 > $O/prof_headline.log 2>&1
python tools/update_table.py gpurun_out/prof_r04_headline/run_kernel_trace.csv 70 78 > $O/update_table_single_step.txt 2>&1 || true
python tools/gap_report.py gpurun_out/prof_r04_headline/run_kernel_trace.csv 70 78 > $O/gap_report.txt 2>&1 || true
python tools/fit_table.py gpurun_out/prof_r04_headline/run_kernel_trace.csv > $O/fit_table.txt 2>&1 || true
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
echo "== 2 ranks sharing the card (gloo rehearsal of the N>1 line), per-shard and all-reduced step control"
for M in shard global; do
NLBAC_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline --dp-step-control $M > $O/bench_2rank_gloo_$M.json 2> $O/bench_2rank_gloo_$M.err || tail -5 $O/bench_2rank_gloo_$M.err
done
echo "== micro-benchmarks"
python tools/microbench_node.py > $O/microbench_node_fwd.txt 2>&1
NLBAC_NODE_SPLIT=0 python tools/microbench_node.py > $O/microbench_node_fwd_nosplit.txt 2>&1
python tools/microbench_node_bwd.py > $O/microbench_node_bwd.txt 2>&1
NLBAC_NODE_SPLIT=0 python tools/microbench_node_bwd.py > $O/microbench_node_bwd_nosplit.txt 2>&1
python tools/microbench_mlp.py > $O/microbench_mlp.txt 2>&1
NLBAC_MLP_RRQ=0 python tools/microbench_mlp.py > $O/microbench_mlp_halfpanel.txt 2>&1
NLBAC_MLP_RRQ=0 NLBAC_MLP_RR_BWD=0 NLBAC_MLP_DW64=0 python tools/microbench_mlp.py > $O/microbench_mlp_r03_kernels.txt 2>&1
python tools/microbench_concat.py > $O/microbench_concat.txt 2>&1 || true
python tools/fit_span.py 32768 51 > $O/fit_span.txt 2>&1 || true
mkdir -p tools/micro/bin
hipcc --offload-arch=gfx950 -O3 -w tools/micro/store_pattern.hip -o tools/micro/bin/store_pattern && tools/micro/bin/store_pattern > $O/calibration_store_pattern.txt 2>&1
python tools/prefetch_probe.py > $O/prefetch_probe.txt 2>&1
fi
if [ $PART = c ]; then
bash tools/gpu_launch_table.sh r04
cp gpurun_out/r04_launches.txt $O/launches_per_update.txt
echo "== shader-clock stamps (the -DRR_TIMING build made in the build container from the final sources)"
V=$(echo neural-*-nlbac_amd/lib/variants/libnlbac_hip_rrtiming.so)
NLBAC_HIP_LIB=$V python tools/phase_times_rr.py 8192 1 > $O/phase_times_node_rr_fwd.txt 2>&1
NLBAC_NODE_SPLIT=0 NLBAC_HIP_LIB=$V python tools/phase_times_rr.py 8192 1 > $O/phase_times_node_rr_fwd_nosplit.txt 2>&1
NLBAC_HIP_LIB=$V python tools/phase_times_mlp_rr.py > $O/phase_times_mlp_rrq_fwd.txt 2>&1
NLBAC_HIP_LIB=$V python tools/phase_times_mlp_rr_bwd.py > $O/phase_times_mlp_rrq_bwd.txt 2>&1
fi
if [ $PART = d ]; then
echo "== counter passes: HBM traffic (FETCH_SIZE / WRITE_SIZE separately), MFMA busy"
bash tools/gpu_pmc.sh r4_unicycle_dopri5_B4096 > $O/pmc_unicycle.log 2>&1
cp gpurun_out/pmc_r4_unicycle_dopri5_B4096_traffic.json $O/pmc_hbm_traffic_unicycle_dopri5_B4096.json
bash tools/gpu_pmc_mfma.sh r4_unicycle_dopri5_B4096 --steps 30 --warmup 10 > $O/pmc_mfma.log 2>&1
cp gpurun_out/pmc_r4_unicycle_dopri5_B4096_mfma.json $O/pmc_mfma_busy_unicycle_dopri5_B4096.json
fi
echo done
