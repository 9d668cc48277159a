#!/bin/bash
# SQ counters of k_shade_main, one 256-spp pass of the bench scene.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcs
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-include-regex "k_shade_main" -d gpurun_out/pmcs -o a --output-format csv -- python3 tools/render_once.py --spp 256 --passes 1 > gpurun_out/pmcs/a.log 2>&1 || { echo pass a failed; exit 1; }
echo ok
