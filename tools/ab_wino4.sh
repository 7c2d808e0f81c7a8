#!/bin/bash
# A/B of F(4x4) convolution builds on the GPU box: isolated timing (alternating, ROUNDS rounds), optional stamps, optional tests.
# usage: bash tools/ab_wino4.sh <tag> <lib suffix> [<lib suffix> ...]     ("" or "default" = libbmc_hip.so; "w4il0" = libbmc_hip_w4il0.so)
#   env: ROUNDS (2), STAMPS ("w4ilstamp1 w4ilstamp0": stamp builds to print), TESTS (1: tests/test_gpu_r4.py with the default library)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=$1; shift
L=$R/bmcnet-esr_amd/csrc
lib() { [ "$1" = default ] && echo $L/libbmc_hip.so || echo $L/libbmc_hip_$1.so; }
{
for r in $(seq ${ROUNDS:-2}); do
  for s in "$@"; do
    echo "== round $r lib $s: $(W4_ONLY=1 KB_ITERS=200 BMC_HIP_LIB=$(lib $s) timeout 200 python tools/time_wino4.py ${SHAPE:-8 180 240} 2>&1 | grep 'F(4x4)' | sed 's/algorithmic.*executed//' | tr '\n' ' ')"
  done
done
for s in ${STAMPS:-}; do echo "== stamps $s"; BMC_HIP_LIB=$(lib $s) timeout 200 python tools/w4_stamps.py 8 180 240 2>&1 | grep -v amdgpu.ids; done
if [ "${TESTS:-0}" = 1 ]; then timeout 1200 python -m pytest tests/test_gpu_r4.py -x -q -m gpu 2>&1 | tail -5; fi
} > $O/${TAG}_ab.log 2>&1
cat $O/${TAG}_ab.log
