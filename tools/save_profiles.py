"""Copies the summaries of tools/gpu_round.sh from gpurun_out/round/ into profiles/ under a tag: python tools/save_profiles.py r01"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "round")
dst = os.path.join(ROOT, "profiles")
tag = sys.argv[1]
shutil.copy(os.path.join(src, "trace_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
line = open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]
open(os.path.join(dst, tag + "_bench.json"), "w").write(line + "\n")
tot = collections.defaultdict(float)
n = collections.defaultdict(int)
for name in ("fetch", "write"):
    for r in csv.DictReader(open(os.path.join(src, name + "_counter_collection.csv"))):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
launches = n["FETCH_SIZE"]
segments = json.loads(line)["roofline"]["segments_per_launch"] * launches
fetch, write = tot["FETCH_SIZE"] * 1024.0, tot["WRITE_SIZE"] * 1024.0
out = {
    "kernel": "k_extend_tree",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex k_extend_tree -- python3 tools/render_once.py --spp 1536 (separate passes)",
    "launches": launches,
    "FETCH_SIZE_KB_sum": tot["FETCH_SIZE"],
    "WRITE_SIZE_KB_sum": tot["WRITE_SIZE"],
    "hbm_bytes_per_launch": (fetch + write) / launches,
    # gfx950 tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes: fetched bytes are twice the figure
    "hbm_bytes_per_launch_corrected": (2.0 * fetch + write) / launches,
    "spp_per_pass": json.loads(line)["config"]["spp_per_step"],
    "segments_per_launch": segments / launches,
    "note": "FETCH_SIZE / WRITE_SIZE as reported are KB. The kernel reads 64-byte ray records, 16 B per lane per load, lanes "
            "on consecutive records: the gfx950 x2 correction for wide streaming reads applies to FETCH_SIZE. One pass = %d "
            "launches, %.1f M segments: %.0f B fetched (corrected) and %.0f B written per segment against 56 B and 16 B "
            "algorithmic." % (launches, segments / 1e6, 2.0 * fetch / segments, write / segments),
}
for name in (tag + "_extend_hbm_traffic.json", "extend_hbm_traffic.json"):
    json.dump(out, open(os.path.join(dst, name), "w"), indent=1)
print(out["note"])
