"""Summarise a rocprofv3 --kernel-trace --stats run: python tools/prof_summary.py <dir> [steps]"""
import csv
import glob
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fs = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {fs[0]}: total kernel time {tot/1e6:.3f} ms over {steps} steps = {tot/1e6/steps:.3f} ms/step")
print(f"{'kernel':80s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
for r in rows[:45]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:80]:80s} {int(r['Calls'])/steps:10.1f} {float(r['TotalDurationNs'])/1e6/steps:9.3f} "
          f"{float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
