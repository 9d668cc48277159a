#!/bin/bash
# Several builds of the library on the same passes: tools/abn.sh "<names>" [spp] [passes] [extra render_once.py arguments]
# (name x = raytracer_amd/librt_hip_x.so; "cur" = the current build)
spp=${2:-512}; passes=${3:-2}
for rep in 1 2; do
for lib in $1; do
  if [ $lib = cur ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=raytracer_amd/librt_hip_$lib.so; fi
  echo -n "$lib: "; timeout -k 10 100 python tools/render_once.py --spp $spp --passes $passes $4 | sed 's/.*s, \([0-9.]* Mrays\/s\); ms: \(.*\)/\1 \2/' || exit 1
done; done
