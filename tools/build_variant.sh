#!/bin/bash
# Diagnostic: builds a variant of the library with extra compiler flags into tools/_variants/<name>.so
# (not the shipped library).  Usage: tools/build_variant.sh <name> "<flags>"; run a tool against it
# with OEI_LIB_OVERRIDE=tools/_variants/<name>.so.
set -e
NAME=$1; FLAGS=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/tools/_variants/obj_$NAME
mkdir -p $D
for f in oei_model oei_gemm oei_sdp oei_acq oei_hess oei_fit oei_lbfgs oei_newton oei_multistart; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=1000000 -mllvm -amdgpu-mfma-vgpr-form --offload-arch=gfx950 $FLAGS -c $ROOT/bayesian_optimization_amd/csrc/$f.hip -o $D/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/_variants/$NAME.so $D/*.o
rm -rf $D
echo built tools/_variants/$NAME.so
