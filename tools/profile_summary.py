"""Turns what tools/gpu_profile.sh left in gpurun_out/prof/ into the summaries kept under profiles/:

    python tools/profile_summary.py r02

  <tag>_bench.json, <tag>_bench_kernel_stats.csv        the bench line and the rocprofv3 --kernel-trace --stats of the same command
  <tag>_extend_hbm_traffic.json (+ extend_hbm_traffic.json, which bench.py reads)   FETCH_SIZE / WRITE_SIZE of k_extend_tree
  <tag>_sq_counters.json (+ sq_counters.json)           SQ counters of k_extend_tree and k_shade_main at the bench's pass size
  <tag>_l2_counters.json                                TCC hit / miss / request counters of both kernels
  <tag>_c4_extend.json                                  the same figures for k_extend_tree on the 1 M-sphere scene

Units and corrections: FETCH_SIZE and WRITE_SIZE are KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
16-byte-per-lane streaming reads at 64 bytes, so fetched bytes are twice the figure; SQ_BUSY_CYCLES is summed over the 32
shader engines; one VALU instruction occupies its SIMD for 4 cycles; the chip has 1024 SIMDs."""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1]


def find(name):
    """rocprofv3 -d DIR -o NAME writes NAME_<what>.csv, sometimes below a host directory."""
    for base, _, files in os.walk(SRC):
        if name in files:
            return os.path.join(base, name)
    raise FileNotFoundError(name)


def per_dispatch(path, kernel):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].replace("void ", "").startswith(kernel):
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def totals(path, kernel):
    tot, n = collections.defaultdict(float), 0
    for d in per_dispatch(path, kernel):
        n += 1
        for k, v in d.items():
            tot[k] += v
    return dict(tot), n


def render_line(log):
    """'final_scene: S samples, G segments, T s, R Mrays/s; ms: raygen a tree b shade_main c ...' of tools/render_once.py"""
    for line in open(log):
        m = re.search(r"(\d+) samples, (\d+) segments, ([\d.]+) s, ([\d.]+) Mrays/s; ms: raygen ([\d.]+) tree ([\d.]+) shade_main ([\d.]+)", line)
        if m:
            return {"samples": int(m.group(1)), "segments": int(m.group(2)), "seconds": float(m.group(3)), "mrays_per_s": float(m.group(4)),
                    "tree_ms": float(m.group(6)), "shade_main_ms": float(m.group(7))}
    return None


def sq_summary(rows, pick):
    def one(d):
        cycles = d["SQ_BUSY_CYCLES"] / 32.0
        return {"kernel_cycles": cycles, "valu_instructions": d["SQ_INSTS_VALU"],
                "valu_busy": 4.0 * d["SQ_INSTS_VALU"] / (1024.0 * cycles),
                "lane_use": d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])}
    tot = collections.defaultdict(float)
    for d in rows:
        for k, v in d.items():
            tot[k] += v
    out = one(tot)
    out["launches"] = len(rows)
    out["by_bounce"] = [dict(one(rows[i]), bounce=i) for i in pick if i < len(rows)]
    if "SQ_WAVE_CYCLES" in tot and "SQ_WAIT_INST_ANY" in tot:
        out["wave_cycles_waiting_to_issue"] = tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_ANY" in tot:
        out["wave_cycles_waiting_for_memory_or_barrier"] = tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]
    return out


def kernel_avg_ms(stats_csv, prefix):
    """(calls, average ms) of the kernels whose name starts with prefix in a rocprofv3 kernel_stats.csv"""
    calls, total = 0, 0.0
    for r in csv.DictReader(open(stats_csv)):
        if r["Name"].replace("void ", "").startswith(prefix):
            calls += int(r["Calls"])
            total += float(r["TotalDurationNs"])
    return calls, (total / calls / 1e6 if calls else 0.0)


# ---- the bench scene ----------------------------------------------------------------------------------------------
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
open(os.path.join(DST, tag + "_bench.json"), "w").write(json.dumps(bench) + "\n")
stats_csv = find("trace_kernel_stats.csv")
shutil.copy(stats_csv, os.path.join(DST, tag + "_bench_kernel_stats.csv"))
calls, avg_ms = kernel_avg_ms(stats_csv, "k_extend_tree")
print("bench: %.3f G rays/s; k_extend_tree HIP events %.3f ms per launch, rocprofv3 trace %.3f ms over %d launches" %
      (bench["value"] / 1e9, bench["roofline"]["avg_launch_ms"], avg_ms, calls))

line = render_line(os.path.join(SRC, "fetch.log"))
f_tot, launches = totals(find("fetch_counter_collection.csv"), "k_extend_tree")
w_tot, _ = totals(find("write_counter_collection.csv"), "k_extend_tree")
fetch, write = f_tot["FETCH_SIZE"] * 1024.0, w_tot["WRITE_SIZE"] * 1024.0
seg = line["segments"]
traffic = {
    "kernel": "k_extend_tree",
    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-include-regex k_extend_tree -- python3 tools/render_once.py --spp 1536 --passes 1 (separate runs)",
    "spp_per_pass": 1536, "launches": launches, "segments_per_launch": seg / launches,
    "FETCH_SIZE_KB_sum": f_tot["FETCH_SIZE"], "WRITE_SIZE_KB_sum": w_tot["WRITE_SIZE"],
    "hbm_bytes_per_launch": (fetch + write) / launches,
    "hbm_bytes_per_launch_corrected": (2.0 * fetch + write) / launches,
    "fetched_bytes_per_segment_corrected": 2.0 * fetch / seg, "written_bytes_per_segment": write / seg,
    "algorithmic_bytes_per_segment": {"read": 56, "written": 16},
    "note": "FETCH_SIZE / WRITE_SIZE are KB as reported; the kernel reads 64-byte ray records with 16-byte loads, lanes on "
            "consecutive records, so the gfx950 x2 correction for wide streaming reads applies to FETCH_SIZE.",
}
for name in (tag + "_extend_hbm_traffic.json", "extend_hbm_traffic.json"):
    json.dump(traffic, open(os.path.join(DST, name), "w"), indent=1)
print("k_extend_tree HBM: %.1f B fetched (corrected) + %.1f B written per segment (algorithmic 56 + 16)" %
      (traffic["fetched_bytes_per_segment_corrected"], traffic["written_bytes_per_segment"]))

sq = {"command": "rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY ... "
                 "--kernel-include-regex <kernel> -- python3 tools/render_once.py --spp 1536 --passes 1 (final_scene 800x800)",
      "spp_per_pass": 1536,
      "k_extend_tree": sq_summary(per_dispatch(find("sqx_counter_collection.csv"), "k_extend_tree"), [0, 1, 2, 5, 10, 20, 40]),
      "k_shade_main": sq_summary(per_dispatch(find("sqs_counter_collection.csv"), "k_shade_main"), [0, 1, 5, 20])}
for name in (tag + "_sq_counters.json", "sq_counters.json"):
    json.dump(sq, open(os.path.join(DST, name), "w"), indent=1)
for k in ("k_extend_tree", "k_shade_main"):
    print("%s: VALU busy %.3f, lane use %.3f" % (k, sq[k]["valu_busy"], sq[k]["lane_use"]))

l2 = {"command": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum (and TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum "
                 "TCP_TCC_WRITE_REQ_sum) --kernel-include-regex 'k_extend_tree|k_shade_main' -- python3 tools/render_once.py --spp 1536 --passes 1",
      "spp_per_pass": 1536,
      "note": "L2 (TCC) requests are 64 or 128 bytes each; l2_gbs_at_64B_per_request is the lower bound, twice that the upper. "
              "Kernel time from the HIP events of the same run."}
tline = render_line(os.path.join(SRC, "tcc.log"))
for k, ms_key in (("k_extend_tree", "tree_ms"), ("k_shade_main", "shade_main_ms")):
    t, n = totals(find("tcc_counter_collection.csv"), k)
    try:
        p, _ = totals(find("tcp_counter_collection.csv"), k)
    except FileNotFoundError:
        p = {}
    secs = tline[ms_key] * 1e-3
    l2[k] = {"launches": n, "kernel_seconds": secs, **t, **p,
             "l2_hit_rate": t["TCC_HIT_sum"] / max(1.0, t["TCC_HIT_sum"] + t["TCC_MISS_sum"]),
             "l2_requests_per_segment": t["TCC_REQ_sum"] / tline["segments"],
             "l2_gbs_at_64B_per_request": 64.0 * t["TCC_REQ_sum"] / secs / 1e9}
    print("%s: L2 hit rate %.3f, %.2f requests per segment, >= %.0f GB/s" % (k, l2[k]["l2_hit_rate"], l2[k]["l2_requests_per_segment"],
                                                                             l2[k]["l2_gbs_at_64B_per_request"]))
json.dump(l2, open(os.path.join(DST, tag + "_l2_counters.json"), "w"), indent=1)

# ---- the 1 M-sphere scene: nodes out of cache -------------------------------------------------------------------------
c4_line = render_line(os.path.join(SRC, "c4_fetch.log"))
shutil.copy(find("c4_trace_kernel_stats.csv"), os.path.join(DST, tag + "_c4_kernel_stats.csv"))
calls, avg_ms = kernel_avg_ms(find("c4_trace_kernel_stats.csv"), "k_extend_tree")
f_tot, launches = totals(find("c4_fetch_counter_collection.csv"), "k_extend_tree")
w_tot, _ = totals(find("c4_write_counter_collection.csv"), "k_extend_tree")
t_tot, _ = totals(find("c4_tcc_counter_collection.csv"), "k_extend_tree")
seg = c4_line["segments"]
tree_s = render_line(os.path.join(SRC, "c4_tcc.log"))["tree_ms"] * 1e-3
plain = render_line(os.path.join(SRC, "c4_visits.log"))
c4 = {
    "scene": "spheres_grid n = 1 000 000, 3840x2160, 64 spp in one pass (64 MB of 64-byte nodes, 32 MB of sphere records)",
    "command": "rocprofv3 [--kernel-trace --stats | --pmc FETCH_SIZE | WRITE_SIZE | TCC_* | SQ_*] --kernel-include-regex k_extend_tree -- "
               "python3 tools/render_once.py --scene spheres_grid --n 1000000 --width 3840 --height 2160 --spp 64 --passes 1",
    "segments": seg, "launches": launches, "trace_launches": calls, "trace_avg_launch_ms": avg_ms,
    "unprofiled": plain,
    "FETCH_SIZE_KB_sum": f_tot["FETCH_SIZE"], "WRITE_SIZE_KB_sum": w_tot["WRITE_SIZE"],
    "fetched_bytes_per_segment_as_reported": f_tot["FETCH_SIZE"] * 1024.0 / seg,
    "written_bytes_per_segment": w_tot["WRITE_SIZE"] * 1024.0 / seg,
    **t_tot,
    "l2_hit_rate": t_tot["TCC_HIT_sum"] / max(1.0, t_tot["TCC_HIT_sum"] + t_tot["TCC_MISS_sum"]),
    "l2_requests_per_segment": t_tot["TCC_REQ_sum"] / seg,
    "l2_gbs_at_64B_per_request": 64.0 * t_tot["TCC_REQ_sum"] / tree_s / 1e9,
    "sq": sq_summary(per_dispatch(find("c4_sqx_counter_collection.csv"), "k_extend_tree"), [0, 1, 2, 5, 10]),
    "note": "FETCH_SIZE as reported (KB -> bytes): the node and sphere reads are 64- and 32-byte records at random addresses, not the "
            "16-byte-per-lane streaming pattern the x2 correction is calibrated for; only the ray records (56 B per segment) stream.",
}
json.dump(c4, open(os.path.join(DST, tag + "_c4_extend.json"), "w"), indent=1)
print("C4 k_extend_tree: %.1f B fetched (as reported) + %.1f B written per segment, L2 hit rate %.3f, %.1f L2 requests per segment, "
      "VALU busy %.3f, lane use %.3f" % (c4["fetched_bytes_per_segment_as_reported"], c4["written_bytes_per_segment"], c4["l2_hit_rate"],
                                        c4["l2_requests_per_segment"], c4["sq"]["valu_busy"], c4["sq"]["lane_use"]))
