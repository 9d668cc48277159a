#!/bin/bash
# HBM bytes (FETCH_SIZE, WRITE_SIZE: KB) and time of every kernel but the traversal, one 1536-spp pass of the bench scene:
#   tools/pmc_hbm.sh   -> gpurun_out/hbm/{fetch,write}_counter_collection.csv, trace_kernel_stats.csv
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/hbm
mkdir -p $out
K="k_shade_main|k_raygen|k_accumulate|k_shade<"
B="python3 tools/render_once.py --spp 1536 --passes 1"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$K" -d $out -o fetch --output-format csv -- $B > $out/fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$K" -d $out -o write --output-format csv -- $B > $out/write.log 2>&1 || { echo write failed; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out -o trace --output-format csv -- $B > $out/trace.log 2>&1 || { echo trace failed; exit 1; }
echo ok
