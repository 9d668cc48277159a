"""Counters of the closest-hit kernel (k_extend_tree; k_extend_list for a scene without a tree) for one configuration -> the two JSON files bench.py quotes for that scene and image size.
    python tools/profile_config_summary.py <name> <spp per pass> <directory of the rocprofv3 runs> [scene arguments as given to render_once.py]
FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE x 2 on gfx950 for the 16-byte-per-lane streaming reads of the ray records;
SQ_BUSY_CYCLES is summed over 32 shader engines; a VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs."""
import collections
import csv
import json
import os
import re
import sys

name, spp, src = sys.argv[1], int(sys.argv[2]), sys.argv[3]
args = sys.argv[4:]


def opt(flag, default):
    return type(default)(args[args.index(flag) + 1]) if flag in args else default


def find(fname):
    for base, _, files in os.walk(src):
        if fname in files:
            return os.path.join(base, fname)
    raise FileNotFoundError(fname)


KERNELS = set()  # k_extend_tree, or k_extend_list for a scene without a tree


def per_dispatch(path):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"].replace("void ", "")
        if kn.startswith("k_extend_"):
            KERNELS.add(kn.split("<")[0].split("(")[0])
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def total(rows):
    t = collections.defaultdict(float)
    for d in rows:
        for k, v in d.items():
            t[k] += v
    return t


def render_line(log):
    for line in open(log):
        m = re.search(r"(\w+): (\d+) samples, (\d+) segments, ([\d.]+) s, ([\d.]+) Mrays/s; ms: raygen ([\d.]+) tree ([\d.]+) shade_main ([\d.]+)", line)
        if m:
            return {"scene": m.group(1), "samples": int(m.group(2)), "segments": int(m.group(3)), "tree_ms": float(m.group(7))}
    raise SystemExit("no render line in " + log)


line = render_line(os.path.join(src, "fetch.log"))
scene = line["scene"]
# image size and depth as render_once.py / bench.py resolve them: the scene's own unless given
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api  # noqa: E402
b = api.Builder()
b.register_reference_textures(tuple(api.TEXTURE_FILES) if scene == "my_scene" else ("earthmap.jpg",))
info = b.named_scene(scene, n=min(opt("--n", 0), 1000) if scene == "spheres_grid" else opt("--n", 0))
width, height = opt("--width", info.width), opt("--height", info.height)
common = {"scene": scene, "width": width, "height": height, "max_depth": info.max_depth, "spp_per_pass": spp, "arguments": " ".join(args)}

f_rows, w_rows = per_dispatch(find("fetch_counter_collection.csv")), per_dispatch(find("write_counter_collection.csv"))
fetch, write = total(f_rows)["FETCH_SIZE"] * 1024.0, total(w_rows)["WRITE_SIZE"] * 1024.0
launches, seg = len(f_rows), line["segments"]
kernel = sorted(KERNELS)[0] if len(KERNELS) == 1 else "k_extend_tree"
traffic = dict(common, kernel=kernel, launches=launches, segments_per_launch=seg / launches,
               command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-include-regex k_extend_ -- python3 tools/render_once.py --spp %d --passes 1 %s (separate runs)" % (spp, " ".join(args)),
               hbm_bytes_per_launch=(fetch + write) / launches, hbm_bytes_per_launch_corrected=(2.0 * fetch + write) / launches,
               fetched_bytes_per_segment_corrected=2.0 * fetch / seg, fetched_bytes_per_segment_as_reported=fetch / seg,
               written_bytes_per_segment=write / seg, algorithmic_bytes_per_segment={"read": 56, "written": 16},
               note="FETCH_SIZE x 2: calibrated for the 16-byte-per-lane streaming reads of the 64-byte ray records; where the scene's "
                    "nodes and primitives do not fit the caches (spheres_grid 1 M) their random 64- and 32-byte reads are in FETCH_SIZE "
                    "too and the truth lies between the two figures")
sq_rows = per_dispatch(find("sqx_counter_collection.csv"))


def one(d):
    cycles = d["SQ_BUSY_CYCLES"] / 32.0
    o = {"kernel_cycles": cycles, "valu_instructions": d["SQ_INSTS_VALU"], "valu_busy": 4.0 * d["SQ_INSTS_VALU"] / (1024.0 * max(cycles, 1.0)),
         "lane_use": d["SQ_THREAD_CYCLES_VALU"] / (64.0 * max(d["SQ_ACTIVE_INST_VALU"], 1.0))}  # (a launch that found no rays counts nothing)
    if d.get("SQ_WAVE_CYCLES"):
        o["wave_cycles_waiting_to_issue"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
        o["wave_cycles_waiting_for_memory_or_barrier"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    return o


sq = dict(common, command="rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY "
                          "SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --kernel-include-regex k_extend_ -- python3 tools/render_once.py --spp %d --passes 1 %s" % (spp, " ".join(args)))
sq["kernel"] = kernel
sq[kernel] = dict(one(total(sq_rows)), launches=len(sq_rows), by_bounce=[dict(one(sq_rows[i]), bounce=i) for i in (0, 1, 2, 5, 10, 20, 40) if i < len(sq_rows)])
os.makedirs(os.path.join(src, "out"), exist_ok=True)
json.dump(traffic, open(os.path.join(src, "out", name + "_extend_hbm_traffic.json"), "w"), indent=1)
json.dump(sq, open(os.path.join(src, "out", name + "_sq_counters.json"), "w"), indent=1)
k = sq[kernel]
print("%s %dx%d @ %d spp: %s VALU busy %.3f, lane use %.3f, waiting for memory %.3f; HBM %.1f B fetched (x2) + %.1f B written per segment" %
      (scene, width, height, spp, kernel, k["valu_busy"], k["lane_use"], k.get("wave_cycles_waiting_for_memory_or_barrier", float("nan")),
       traffic["fetched_bytes_per_segment_corrected"], traffic["written_bytes_per_segment"]))
