#!/bin/bash
# Bench line + rocprofv3 evidence for ONE configuration, in one GPU-box session (each counter set in a run of its own):
#   tools/gpu_profile_config.sh <name> <spp per pass> [bench.py / render_once.py scene arguments...]
#   e.g. tools/gpu_profile_config.sh cornell_box 1536 --scene cornell_box
#        tools/gpu_profile_config.sh spheres_grid 64 --scene spheres_grid --n 1000000 --width 3840 --height 2160
# Leaves under gpurun_out/cfg_<name>/out/ what belongs in profiles/: <name>_extend_hbm_traffic.json, <name>_sq_counters.json
# (bench.py quotes them for this scene and image size), kernel_stats.csv of the bench command, and bench.json - the bench
# line taken AFTER the counters, so that it carries them.
set -o pipefail
export TMPDIR=/tmp
name=$1; spp=$2; shift 2
out=gpurun_out/cfg_$name
mkdir -p $out/out
R="python3 tools/render_once.py --spp $spp --passes 1 $*"
run() { local n=$1 secs=$2; shift 2; timeout -s KILL $secs "$@" > $out/$n.log 2>&1 || { echo "$n failed"; tail -5 $out/$n.log; exit 1; }; echo "$n ok"; }
run fetch 240 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_extend_" -d $out -o fetch --output-format csv -- $R
run write 240 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_extend_" -d $out -o write --output-format csv -- $R
run sqx 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --kernel-include-regex "k_extend_" -d $out -o sqx --output-format csv -- $R
python3 tools/profile_config_summary.py $name $spp $out "$@" || exit 1
# the counters just written are what bench.py will quote: copy them where it looks
cp $out/out/${name}_extend_hbm_traffic.json $out/out/${name}_sq_counters.json profiles/
run trace 400 rocprofv3 --kernel-trace --stats -d $out -o trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --spp $spp "$@"
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --cpu-seconds 8 --spp $spp "$@" > $out/out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
cp "$(find $out -name trace_kernel_stats.csv | head -1)" $out/out/kernel_stats.csv
python3 - <<PY
import json
j = json.loads(open("$out/out/bench.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("$name: %.3f G rays/s, %.1f ms/step; %s %.3f ms/launch, hbm frac %.3f, valu %s, traffic %s" % (j["value"] / 1e9, j["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], r["valu"], r["traffic"]))
PY
