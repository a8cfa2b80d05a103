"""Summarise a rocprofv3 --kernel-trace --stats run: python tools/prof_summary.py <dir> [steps]

Prints the per-kernel table and, first, one line per kernel TEMPLATE (all instantiations of
igemm_kernel / wgrad_*_kernel summed): bench.py's `roofline` is quoted on the igemm_kernel
template as a whole, so its `avg_launch_us` is compared with that line.
"""
import csv
import glob
import re
import sys
from collections import OrderedDict

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fs = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)


def clean(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "")


print(f"# {fs[0]}: total kernel time {tot/1e6:.3f} ms over {steps} steps = {tot/1e6/steps:.3f} ms/step")
fam = OrderedDict()
for r in rows:
    base = re.split(r"[<(]", clean(r["Name"]))[0]
    t, c = fam.get(base, (0.0, 0))
    fam[base] = (t + float(r["TotalDurationNs"]), c + int(r["Calls"]))
print(f"{'kernel template (all instantiations)':80s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
for base, (t, c) in list(sorted(fam.items(), key=lambda kv: -kv[1][0]))[:12]:
    print(f"{base[:80]:80s} {c/steps:10.1f} {t/1e6/steps:9.3f} {t/c/1e3:9.1f} {100*t/tot:6.1f}")
print()
print(f"{'kernel':80s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
for r in rows[:50]:
    name = clean(r["Name"])
    print(f"{name[:80]:80s} {int(r['Calls'])/steps:10.1f} {float(r['TotalDurationNs'])/1e6/steps:9.3f} "
          f"{float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
