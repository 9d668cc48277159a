#!/bin/bash
# sweeps the k_extend_tree scheduling knobs on the GPU box: tune.sh "<refill values>" "<leaf values>" [scene args]
for r in ${1:-4 8 16 32}; do for l in ${2:-8 16 32 48}; do
  echo -n "refill=$r leaf=$l: "; RT_TUNE_REFILL=$r RT_TUNE_LEAF=$l timeout -k 10 100 python tools/render_once.py --spp 64 --passes 1 $3 | sed 's/.*tree \([0-9.]*\).*/tree \1/' || exit 1
done; done
