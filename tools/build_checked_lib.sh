#!/bin/bash
# The library with every reference checked before it is followed (-DRT_BOUNDS_CHECK: type and index against the section
# lengths; violations are printed from the device and the access is skipped): raytracer_amd/librt_hip_chk.so; use with
#   RT_HIP_LIB=raytracer_amd/librt_hip_chk.so python -m pytest tests -m gpu ...
set -e
cd "$(dirname "$0")/.."
c=raytracer_amd/csrc
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -DRT_BOUNDS_CHECK "$@" $c/rt_kernels.hip $c/rt_tree_build.hip $c/rt_builder.cpp $c/rt_scenes.cpp -o raytracer_amd/librt_hip_chk.so
