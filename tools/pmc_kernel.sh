#!/bin/bash
# SQ counters of one kernel on one scene: tools/pmc_kernel.sh <kernel regex> <render_once.py arguments...>
set -o pipefail
export TMPDIR=/tmp
k="$1"; shift
mkdir -p gpurun_out/pmck
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-include-regex "$k" -d gpurun_out/pmck -o a --output-format csv -- python3 tools/render_once.py "$@" > gpurun_out/pmck/a.log 2>&1 || { echo pass a failed; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH --kernel-include-regex "$k" -d gpurun_out/pmck -o b --output-format csv -- python3 tools/render_once.py "$@" > gpurun_out/pmck/b.log 2>&1 || { echo pass b failed; exit 1; }
echo ok
