#!/bin/bash
# Round-2 counter passes, calibrations and the long training probe on the GPU box (everything lands under gpurun_out/r02/; the summaries that are kept go to
# profiles/r02_*):  bash tools/gpu_r02_counters.sh
set -e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
echo "== counters"
bash tools/gpu_pmc.sh unicycle_dopri5_B4096 > $O/pmc_u.log 2>&1
bash tools/gpu_pmc.sh pvtol_dopri5_B16384_adjoint --env Pvtol --batch 16384 --adjoint > $O/pmc_pa.log 2>&1
bash tools/gpu_pmc.sh pvtol_dopri5_B16384 --env Pvtol --batch 16384 > $O/pmc_pd.log 2>&1
bash tools/gpu_pmc_mfma.sh unicycle_dopri5_B4096 --steps 30 --warmup 10 > $O/pmc_mfma.log 2>&1
echo "== calibrations / phase stamps (tools/micro, -DEXP_TIMING build)"
tools/micro/bin/mfma_rate > $O/mfma_rate.txt 2>&1 || true
tools/micro/bin/gemm_loop_d4 > $O/gemm_loop.txt 2>&1 || true
V=$PWD/$(echo neural-*-nlbac_amd/lib)/variants
PHASE_FWD_ONLY=1 NLBAC_HIP_LIB=$V/libnlbac_hip_timing.so python tools/phase_times.py 8192 > $O/phase_times_fwd.txt 2>&1 || true
echo "== long training probe (stiff regime)"
timeout -k 10 300 python tools/train_probe.py 6000 > $O/train_probe.log 2>&1 || tail -3 $O/train_probe.log
echo done
