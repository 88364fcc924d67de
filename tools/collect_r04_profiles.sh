#!/bin/bash
# Build container: copy the summaries of tools/gpu_r04_measure.sh a..d, tools/gpu_cadj_measure.sh and the Pvtol traces
# (gpurun_out/, scratch) into profiles/r04_* (tracked).
R=gpurun_out/r04; P=profiles
strip() { grep -v "amdgpu.ids" "$1"; }
cp $R/bench_headline.json $P/r04_bench_unicycle_dopri5_B4096_bench_line.json
cp $R/bench_20steps.json $P/r04_bench_unicycle_dopri5_B4096_20steps_bench_line.json
cp gpurun_out/prof_r04_headline/run_kernel_stats.csv $P/r04_bench_unicycle_dopri5_B4096_kernel_stats.csv
cp gpurun_out/prof_r04_headline/bench.json $P/r04_bench_unicycle_dopri5_B4096_profiled_bench_line.json
for v in euler rk4 cars nbc_unicycle; do cp $R/bench_$v.json $P/r04_bench_variant_$v.json; done
cp $R/bench_quadrotorlike.json $P/r04_bench_quadrotorlike_B32768.json
cp $R/bench_pvtol.json $P/r04_bench_pvtol_B16384_dopri5_direct.json
cp $R/bench_pvtol_adjoint.json $P/r04_bench_pvtol_B16384_dopri5_adjoint.json
for m in shard global; do cp $R/bench_2rank_gloo_$m.json $P/r04_bench_2rank_gloo_one_card_$m.json; done
cp $R/update_table_single_step.txt $P/r04_update_kernel_table_single_step.txt
cp $R/gap_report.txt $P/r04_gap_report_single_step.txt
cp $R/fit_table.txt $P/r04_fit_table.txt; strip $R/fit_span.txt > $P/r04_fit_span.txt
cp $R/launches_per_update.txt $P/r04_launches_per_update.txt
strip $R/phase_times_node_rr_bwd.txt > $P/r04_phase_times_node_rr_bwd.txt
strip $R/phase_times_node_rr_fwd.txt > $P/r04_phase_times_node_rr_fwd.txt
{ strip $R/phase_times_mlp_rrq_fwd.txt; strip $R/phase_times_mlp_rrq_bwd.txt; } > $P/r04_phase_times_mlp_rrq.txt
cp $R/pmc_hbm_traffic_unicycle_dopri5_B4096.json $P/r04_pmc_hbm_traffic_unicycle_dopri5_B4096.json
cp $R/pmc_mfma_busy_unicycle_dopri5_B4096.json $P/r04_pmc_mfma_busy_unicycle_dopri5_B4096.json
strip $R/prefetch_probe.txt > $P/r04_prefetch_probe.txt
strip $R/calibration_store_pattern.txt > $P/r04_calibration_store_pattern.txt
C=gpurun_out/r4/cadj
for s in rk4 dopri5; do for k in direct adjoint adjoint_staged; do cp $C/cars_${s}_$k.json $P/r04_bench_cars_${s}_$k.json; done; done
cp $C/quad_adjoint.json $P/r04_bench_quadrotorlike_adjoint.json; cp $C/quad_adjoint_staged.json $P/r04_bench_quadrotorlike_adjoint_staged.json
cp gpurun_out/trace_pvtol_direct/table.txt $P/r04_update_kernel_table_pvtol_direct.txt
cp gpurun_out/trace_pvtol_adjoint/table.txt $P/r04_update_kernel_table_pvtol_adjoint.txt
cp gpurun_out/trace_pvtol_direct/gap_report.txt $P/r04_gap_report_pvtol_direct.txt
python3 - <<'PY'
R = 'gpurun_out/r04/'
def body(f): return "".join(l for l in open(R + f) if "amdgpu.ids" not in l)
old = open('profiles/r04_microbench_kernels.txt').read()
adam = old[old.index("== tools/microbench_adam.py"):]
out = ("== tools/microbench_node.py (fused RK forward, 8192 rows)\n" + body("microbench_node_fwd.txt") +
       "== the same with NLBAC_NODE_SPLIT=0\n" + body("microbench_node_fwd_nosplit.txt") +
       "== tools/microbench_node_bwd.py\n" + body("microbench_node_bwd.txt") +
       "== the same with NLBAC_NODE_SPLIT=0\n" + body("microbench_node_bwd_nosplit.txt") +
       "== tools/microbench_mlp.py (B = 4096)\n" + body("microbench_mlp.txt") +
       "== the same with NLBAC_MLP_RRQ=0 (half-panel kernels)\n" + body("microbench_mlp_halfpanel.txt") +
       "== the same with NLBAC_MLP_RRQ=0 NLBAC_MLP_RR_BWD=0 NLBAC_MLP_DW64=0 (round 3's kernels)\n" + body("microbench_mlp_r03_kernels.txt") +
       "== tools/microbench_concat.py (SimulatedCars' NODE, 2 x 8192 rows; four waves per workgroup, odd LDS stride)\n" + body("microbench_concat.txt") + adam)
open('profiles/r04_microbench_kernels.txt', 'w').write(out)
PY
echo copied
