"""Sums rocprofv3 --pmc CSV output per kernel and counter: python tools/pmc_by_kernel.py <counter_collection.csv>..."""
import collections
import csv
import re
import sys

tot = collections.defaultdict(float)
cnt = collections.defaultdict(int)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        tot[(name, r["Counter_Name"])] += float(r["Counter_Value"])
        cnt[(name, r["Counter_Name"])] += 1
for (k, c), v in sorted(tot.items()):
    print("%-44s %-14s total %.5g  dispatches %d" % (k, c, v, cnt[(k, c)]))
