#!/bin/bash
# One GPU-box session: tests, bench, kernel trace of the bench, HBM counters of the traversal kernel.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/round
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/round/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/round/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err || { echo bench failed; tail -5 gpurun_out/round/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/round -o trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/round/bench_traced.json 2> gpurun_out/round/trace.err || { echo trace failed; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_extend_tree" -d gpurun_out/round -o fetch --output-format csv -- python3 tools/render_once.py --spp 1536 > /dev/null 2>&1 || { echo fetch pmc failed; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_extend_tree" -d gpurun_out/round -o write --output-format csv -- python3 tools/render_once.py --spp 1536 > /dev/null 2>&1 || { echo write pmc failed; exit 1; }
echo ok
