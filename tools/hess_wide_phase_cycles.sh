#!/bin/bash
# Diagnostic: builds a timing variant of the library (not the shipped one) and prints the cycles
# k_hess_wide spends per phase.  Run from the repo root on a GPU box.  Usage: tools/hess_wide_phase_cycles.sh [B]
set -e
SRC=bayesian_optimization_amd/csrc
D=$(mktemp -d)
for f in oei_model oei_gemm oei_sdp oei_acq oei_hess oei_fit oei_lbfgs oei_newton oei_multistart; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -mllvm -amdgpu-mfma-vgpr-form --offload-arch=gfx950 -DOEI_HESS_TIMING -c $SRC/$f.hip -o $D/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboei_timing.so $D/*.o
OEI_LIB_OVERRIDE=$D/liboei_timing.so python tools/hess_wide_phase_cycles.py "${1:-256}"
