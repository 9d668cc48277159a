#!/bin/bash
# Every profile the benchmark line quotes, in one GPU-box session (each counter set in a run of its own):
#   tools/gpu_profile.sh            -> gpurun_out/prof/...; then `python tools/profile_summary.py r02` on the CPU side
# Bench scene at the bench's pass size (1536 spp), and the 1 M-sphere scene (nodes out of cache) at 64 spp.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/prof
mkdir -p $out
B="python3 tools/render_once.py --spp 1536 --passes 1"
C="python3 tools/render_once.py --scene spheres_grid --n 1000000 --width 3840 --height 2160 --spp 64 --passes 1"
run() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  timeout -s KILL $secs "$@" > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; exit 1; }
  echo "$name ok"
}
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
echo "bench ok"
run trace 300 rocprofv3 --kernel-trace --stats -d $out -o trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0
run fetch 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_extend_tree" -d $out -o fetch --output-format csv -- $B
run write 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_extend_tree" -d $out -o write --output-format csv -- $B
run sqx 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-include-regex "k_extend_tree" -d $out -o sqx --output-format csv -- $B
run sqs 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-include-regex "k_shade_main" -d $out -o sqs --output-format csv -- $B
run tcc 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-include-regex "k_extend_tree|k_shade_main" -d $out -o tcc --output-format csv -- $B
run tcp 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --kernel-include-regex "k_extend_tree|k_shade_main" -d $out -o tcp --output-format csv -- $B
# the out-of-cache regime
run c4_trace 300 rocprofv3 --kernel-trace --stats -d $out -o c4_trace --output-format csv -- $C
run c4_fetch 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_extend_tree" -d $out -o c4_fetch --output-format csv -- $C
run c4_write 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_extend_tree" -d $out -o c4_write --output-format csv -- $C
run c4_tcc 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-include-regex "k_extend_tree" -d $out -o c4_tcc --output-format csv -- $C
run c4_sqx 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --kernel-include-regex "k_extend_tree" -d $out -o c4_sqx --output-format csv -- $C
run c4_visits 300 python3 tools/render_once.py --scene spheres_grid --n 1000000 --width 3840 --height 2160 --spp 64 --passes 2
ls $out | head -80
echo all ok
