"""Per-dispatch durations of one kernel from a rocprofv3 kernel trace: python tools/trace_by_bounce.py <kernel_trace.csv> <name prefix> [live counts...]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].replace("void ", "").startswith(sys.argv[2])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
live = [int(x) for x in sys.argv[3:]]
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    extra = "  %.3f ns/ray" % (d * 1e3 / live[i]) if i < len(live) else ""
    print("%3d %8.1f us%s" % (i, d, extra))
