#!/bin/bash
# Registers, spills and LDS of every kernel in rt_kernels.hip (device-only assembly; no GPU needed):
#   tools/kernel_regs.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
out=${TMPDIR:-/tmp}/rt_kernels_gfx950.s
/opt/rocm/bin/hipcc -w -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --offload-device-only -S "$@" raytracer_amd/csrc/rt_kernels.hip -o "$out"
awk '/^    \.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_spill_count:/ {sp=$2} /\.private_segment_fixed_size:/ {pr=$2} /\.group_segment_fixed_size:/ {lds=$2} /\.wavefront_size:/ {printf "%-110s vgpr %3s sgpr %3s spill %3s scratch %5s lds %6s\n", name, v, sg, sp, pr, lds}' "$out" | c++filt | sort
