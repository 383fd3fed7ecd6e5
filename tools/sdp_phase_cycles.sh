#!/bin/bash
# Diagnostic: builds a timing variant of the library (not the shipped one) and prints the cycles
# the SDP solver spends per phase for one wave.  Run from the repo root on a GPU box.
# Usage: [SDP_EXTRA_FLAGS=-D...] tools/sdp_phase_cycles.sh <q> <B> [source dir (default: the package's csrc)]
set -e
SRC=${3:-bayesian_optimization_amd/csrc}
D=$(mktemp -d)
for f in oei_model oei_gemm oei_sdp oei_acq oei_hess oei_fit oei_lbfgs oei_newton oei_multistart; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -mllvm -amdgpu-mfma-vgpr-form --offload-arch=gfx950 -DOEI_SDP_TIMING $SDP_EXTRA_FLAGS -c $SRC/$f.hip -o $D/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboei_timing.so $D/*.o
OEI_LIB_OVERRIDE=$D/liboei_timing.so python tools/sdp_phase_cycles.py "$1" "$2"
if [ "$1" = 20 ]; then OEI_LIB_OVERRIDE=$D/liboei_timing.so python tools/stage2_phase_cycles.py "$2"; fi
if [ "$1" = 50 ]; then OEI_LIB_OVERRIDE=$D/liboei_timing.so python tools/stage2_phase_cycles.py "$2" cfg3; fi
