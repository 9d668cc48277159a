"""Counters of single dispatches of one kernel: python tools/pmc_by_dispatch.py <counter_collection.csv> <name prefix> <dispatch index>..."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].replace("void ", "").startswith(sys.argv[2])]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
for k in sys.argv[3:]:
    d = by[ids[int(k)]]
    print("dispatch %s:" % k, "  ".join("%s %.4g" % (n, v) for n, v in sorted(d.items())))
