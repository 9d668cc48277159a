#!/bin/bash
# One iteration on the GPU box: the GPU suite, then final_scene passes at a few scheduling thresholds.
#   tools/gpu_iter.sh "<refill values>" "<leaf values>" [spp]
set -o pipefail
mkdir -p gpurun_out/iter
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/iter/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/iter/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
spp=${3:-512}
timeout -k 10 100 python tools/render_once.py --spp $spp --passes 2 || exit 1
for r in $1; do for l in $2; do
  echo -n "refill=$r leaf=$l: "; RT_TUNE_REFILL=$r RT_TUNE_LEAF=$l timeout -k 10 100 python tools/render_once.py --spp $spp --passes 2 | sed 's/.*Mrays\/s; ms: \(.*\)/\1/' || exit 1
done; done
