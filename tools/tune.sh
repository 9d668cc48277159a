#!/bin/bash
# sweeps the k_extend_tree scheduling knobs on the GPU box: tune.sh "<refill lanes>" "<leaf per cents>" [scene args]
for r in ${1:-30 36 42 48}; do for l in ${2:-60 70 80 90}; do
  echo -n "refill=$r leaf=$l%: "; RT_TUNE_REFILL=$r RT_TUNE_LEAF=$l timeout -k 10 100 python tools/render_once.py --spp 256 --passes 2 $3 | sed 's/.*tree \([0-9.]*\).*/tree \1/' || exit 1
done; done
