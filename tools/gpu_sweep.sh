#!/bin/bash
# final_scene passes over a grid of k_extend_tree scheduling thresholds:
#   tools/gpu_sweep.sh "<refill values>" "<leaf values>" "<medium values>" [spp] [extra render_once.py arguments]
set -o pipefail
spp=${4:-512}
for r in $1; do for l in $2; do for m in $3; do
  echo -n "refill=$r leaf=$l medium=$m: "; RT_TUNE_REFILL=$r RT_TUNE_LEAF=$l RT_TUNE_MEDIUM=$m timeout -k 10 100 python tools/render_once.py --spp $spp --passes 2 $5 | sed 's/.*s, \([0-9.]* Mrays\/s\); ms: \(.*\)/\1 \2/' || exit 1
done; done; done
