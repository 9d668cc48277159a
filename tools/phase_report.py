"""Reads the `phase[bounce ...]` lines a -DRT_PHASE_STATS build prints and says where k_extend_tree's waves spend
their cycles: python tools/phase_report.py < log"""
import sys

names = {40: "finish (media of the lock-step list, publish)", 41: "refill (load, plain lock-step entries, slab constants)",
         42: "box-test loop", 43: "leaf round"}
rows = {}
for line in sys.stdin:
    if line.startswith("phase[bounce"):
        key = line.split("]")[0].split()[-1]
        rows[key] = [int(x) for x in line.split(":")[1].split()]   # (cumulative: the last print wins)
for key, v in rows.items():
    tot = sum(v[k] for k in names)
    if not tot:
        continue
    rays = v[2]
    print("bounce %s: %d rays, %d iterations (%.1f lanes walking), %d refills of %.1f rays, %d finishing rounds of %.1f lanes" %
          (key, rays, v[0], v[7] / max(1, v[0]), v[1], v[2] / max(1, v[1]), v[45], v[46] / max(1, v[45])))
    print("   full tests of lock-step media: %d rounds of %.1f lanes" % (v[47], v[44] / max(1, v[47])))
    print("   box iterations %d with %.1f lanes; leaf rounds %d with %.1f lanes" % (v[3], v[4] / max(1, v[3]), v[5], v[6] / max(1, v[5])))
    kinds = {2: "sphere", 3: "moving sphere", 4: "rect", 5: "box", 6: "medium", 7: "instance", 8: "list", 9: "pop", 10: "leaf named with its chain"}
    print("   leaf kinds: " + ", ".join("%s %d x %.1f" % (kinds[t], v[8 + t], v[24 + t] / v[8 + t]) for t in kinds if v[8 + t]))
    for k, nm in names.items():
        print("   %5.1f %%  %8.1f cycles/ray  %s" % (100.0 * v[k] / tot, v[k] / max(1, rays), nm))
