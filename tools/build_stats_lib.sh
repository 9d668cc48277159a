#!/bin/bash
# The library with the debug counters of k_extend_tree compiled in (how its waves spend their iterations and cycles):
# raytracer_amd/librt_hip_stats.so; use with RT_HIP_LIB=raytracer_amd/librt_hip_stats.so python tools/render_once.py ...
set -e
cd "$(dirname "$0")/.."
c=raytracer_amd/csrc
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -DRT_PHASE_STATS "$@" $c/rt_kernels.hip $c/rt_tree_build.hip $c/rt_builder.cpp $c/rt_scenes.cpp -o raytracer_amd/librt_hip_stats.so
