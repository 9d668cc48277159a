#!/bin/bash
# LDS side of one kernel on one scene: tools/pmc_lds.sh <library name or cur> <kernel regex> <render_once.py arguments...>
set -o pipefail
export TMPDIR=/tmp
lib="$1"; k="$2"; shift 2
if [ "$lib" != cur ]; then export RT_HIP_LIB=raytracer_amd/librt_hip_$lib.so; fi
d=gpurun_out/pmcl_$lib
mkdir -p $d
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-include-regex "$k" -d $d -o a --output-format csv -- python3 tools/render_once.py "$@" > $d/a.log 2>&1 || { echo pass a failed; tail -5 $d/a.log; exit 1; }
python3 tools/pmc_by_kernel.py $(find $d -name "a_counter_collection.csv")
