"""One line per tools/tree_build_timing.py result file: rays/s on the host-built and the device-built tree, height, climb time."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(d["scene"], "host %.3f" % (d["host_sah"]["rays_per_s"] / 1e9), "device %.3f" % (d["device_lbvh"]["rays_per_s"] / 1e9), "height", d["device_lbvh"]["height"], "refit ms %.2f" % d["device_lbvh"]["ms_refit"])
