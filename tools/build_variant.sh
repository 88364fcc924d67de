#!/bin/bash
# Build an experiment variant of the HIP library next to the product one (never loaded unless NLBAC_HIP_LIB names it):
#   tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DEXP_FOO>"   ->  <pkg>/lib/variants/libnlbac_hip_<name>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$(echo "$ROOT"/neural-*-nlbac_amd/csrc)
OUT=$(echo "$ROOT"/neural-*-nlbac_amd/lib)/variants
NAME=$1; shift
TMP=$(mktemp -d)
mkdir -p "$OUT"
for f in mlp_kernels mlp_rr_kernels mlp_rrq_kernels mlp_dw16_kernels node_kernels node_rr_kernels node_adjoint_kernels node_adj_rr_kernels concat_node_kernels concat_rr_kernels concat_adj_rr_kernels optim_kernels agent_kernels ode_kernels env_kernels; do
  X=""; [ $f = node_rr_kernels -o $f = node_adj_rr_kernels -o $f = mlp_rr_kernels -o $f = mlp_rrq_kernels -o $f = concat_rr_kernels -o $f = concat_adj_rr_kernels ] && X="-mllvm -amdgpu-mfma-vgpr-form -DRR_ACC_VGPR"     # (as in csrc/Makefile)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -I"$ROOT"/include -I"$SRC" $X "$@" -c "$SRC/$f.hip" -o "$TMP/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$TMP"/*.o -o "$OUT/libnlbac_hip_$NAME.so"
rm -rf "$TMP"
echo "$OUT/libnlbac_hip_$NAME.so"
