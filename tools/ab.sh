#!/bin/bash
# A/B on one box: the same passes with the baseline build (raytracer_amd/librt_hip_base.so) and the current one.
#   tools/ab.sh [spp] [passes] [extra render_once.py arguments]
spp=${1:-512}; passes=${2:-2}
for rep in 1 2; do
for lib in base cur; do
  if [ $lib = base ]; then export RT_HIP_LIB=raytracer_amd/librt_hip_base.so; else unset RT_HIP_LIB; fi
  echo -n "$lib: "; timeout -k 10 100 python tools/render_once.py --spp $spp --passes $passes $3 | sed 's/.*s, \([0-9.]* Mrays\/s\); ms: \(.*\)/\1 \2/' || exit 1
done; done
