#!/bin/bash
# SQ counters of k_extend_tree, two passes (8 SQ counters each), one 256-spp pass of the bench scene each.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcx
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-include-regex "k_extend_tree" -d gpurun_out/pmcx -o a --output-format csv -- python3 tools/render_once.py --spp 256 --passes 1 > gpurun_out/pmcx/a.log 2>&1 || { echo pass a failed; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_ANY --kernel-include-regex "k_extend_tree" -d gpurun_out/pmcx -o b --output-format csv -- python3 tools/render_once.py --spp 256 --passes 1 > gpurun_out/pmcx/b.log 2>&1 || { echo pass b failed; exit 1; }
echo ok
