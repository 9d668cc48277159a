"""Sums rocprofv3 --pmc CSV output per counter (all dispatches of the selected kernels)."""
import collections
import csv
import sys

tot = collections.defaultdict(float)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()):
    print("%-28s %.5g" % (k, v))
if "SQ_THREAD_CYCLES_VALU" in tot and "SQ_ACTIVE_INST_VALU" in tot:
    print("VALU lane utilisation      %.3f" % (tot["SQ_THREAD_CYCLES_VALU"] / (64 * tot["SQ_ACTIVE_INST_VALU"])))
