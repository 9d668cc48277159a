"""Summarises the SQ counter passes of tools/pmc_extend.sh and tools/pmc_shade.sh (gpurun_out/pmcx, gpurun_out/pmcs) into
profiles/<tag>_sq_counters.json: python tools/sq_summary.py r01

VALU busy = 4 cycles per VALU instruction / (1024 SIMDs x kernel cycles); kernel cycles = SQ_BUSY_CYCLES / 32 (the counter
is summed over the 32 shader engines of the 8 XCDs); lane use = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def dispatches(path, prefix):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].replace("void ", "").startswith(prefix):
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def summarise(rows, pick):
    out = []
    for i in pick:
        if i >= len(rows):
            continue
        d = rows[i]
        cycles = d["SQ_BUSY_CYCLES"] / 32.0
        out.append({
            "bounce": i,
            "kernel_cycles": cycles,
            "valu_instructions": d["SQ_INSTS_VALU"],
            "valu_busy": 4.0 * d["SQ_INSTS_VALU"] / (1024.0 * cycles),
            "lane_use": d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"]),
            "salu_instructions": d.get("SQ_INSTS_SALU"),
        })
    tot_i = sum(d["SQ_INSTS_VALU"] for d in rows)
    tot_c = sum(d["SQ_BUSY_CYCLES"] / 32.0 for d in rows)
    tot_t = sum(d["SQ_THREAD_CYCLES_VALU"] for d in rows)
    tot_a = sum(d["SQ_ACTIVE_INST_VALU"] for d in rows)
    return {"launches": len(rows), "valu_busy": 4.0 * tot_i / (1024.0 * tot_c), "lane_use": tot_t / (64.0 * tot_a),
            "valu_instructions": tot_i, "by_bounce": out}


res = {"command": "rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU ... "
                  "--kernel-include-regex <kernel> -- python3 tools/render_once.py --spp 256 --passes 1 (final_scene 800x800)"}
ext = dispatches(os.path.join(ROOT, "gpurun_out", "pmcx", "a_counter_collection.csv"), "k_extend_tree")
res["k_extend_tree"] = summarise(ext, [0, 1, 2, 5, 10, 20, 40])
sh = os.path.join(ROOT, "gpurun_out", "pmcs", "a_counter_collection.csv")
if os.path.exists(sh):
    res["k_shade_main"] = summarise(dispatches(sh, "k_shade_main"), [0, 1, 5, 20])
json.dump(res, open(os.path.join(ROOT, "profiles", tag + "_sq_counters.json"), "w"), indent=1)
json.dump(res, open(os.path.join(ROOT, "profiles", "sq_counters.json"), "w"), indent=1)
for k in ("k_extend_tree", "k_shade_main"):
    if k in res:
        print(k, "VALU busy %.2f, lane use %.2f" % (res[k]["valu_busy"], res[k]["lane_use"]))
        for b in res[k]["by_bounce"]:
            print("  bounce %2d: busy %.2f lanes %.2f" % (b["bounce"], b["valu_busy"], b["lane_use"]))
