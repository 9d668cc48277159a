#!/bin/bash
# One iteration on the GPU box: the GPU suite, then passes of final_scene with the baseline build (if there is one)
# and the current one.   tools/gpu_iter.sh [spp]
set -o pipefail
mkdir -p gpurun_out/iter
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/iter/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/iter/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
libs=cur; [ -f raytracer_amd/librt_hip_base.so ] && libs="base cur"
tools/abn.sh "$libs" ${1:-512} 2
