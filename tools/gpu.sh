#!/bin/bash
# Local helper: rebuild the HIP library, then run a command on the MI355X box.
#   tools/gpu.sh [--timeout N] -- '<command>'
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -s -C "$ROOT"/neural-*-nlbac_amd/csrc -j4
exec /usr/local/graft/bin/gpurun "$@"
