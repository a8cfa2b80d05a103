"""Per-kernel summary of a rocprofv3 --pmc pass with the eight SQ counters of tools/refresh_profiles.sh:
python tools/sq_summary.py <pmc_dir> [substring ...]

Columns (averages per dispatch, all instantiations whose name contains one of the substrings):
  issue %   = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES   a wave issued an instruction
  stall %   = SQ_WAIT_INST_ANY  / SQ_WAVE_CYCLES    a wave wanted to issue and could not (MFMA dependency / pipe busy, LDS queue ...)
  lds-st %  = SQ_WAIT_INST_LDS  / SQ_WAVE_CYCLES    ... of which: stalled on the LDS instruction queue
  wait %    = SQ_WAIT_ANY       / SQ_WAVE_CYCLES    a wave sat at s_waitcnt / s_barrier
  (the three buckets issue / stall / wait are disjoint and add up to ~100 %: MI355X_MICROARCH.md, rocprofv3 PMC slots)
  mfma %    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch duration x 2.4 GHz): matrix-pipe busy cycles over the
              cycles the chip's 1024 SIMDs had at the MAXIMUM clock -- a lower bound of the pipe's utilisation (the
              chip holds 1.5-2.0 GHz in MFMA-dense kernels, MI355X_MICROARCH.md "DVFS give-back"); durations are
              those of the counter pass itself (profiled dispatches run a few per cent slower)
  lds %     = SQ_LDS_IDX_ACTIVE / (256 CUs x duration x 2.4 GHz)   LDS array cycles, same time base
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:] or [""]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not any(s in name for s in subs) or len(name) > 90:
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            acc[name]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print(f"# {d}: SQ counters per kernel instantiation (average per dispatch)")
print(f"{'kernel':62s} {'n':>4s} {'avg us':>7s} {'issue%':>7s} {'stall%':>7s} {'lds-st%':>7s} {'wait%':>7s} {'mfma%':>7s} {'lds%':>6s}")
rows = []
for k, c in acc.items():
    n = max(cnt[k].values())
    wc, ns = c.get("SQ_WAVE_CYCLES", 0.0), c.get("_ns", 0.0)
    if wc <= 0 or ns <= 0:
        continue
    simd_cycles, cu_cycles = ns * 2.4 * 1024, ns * 2.4 * 256
    rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), k, n, ns / n / 1e3, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc,
                 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles, 100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles))
for _, k, n, us, a, s, l, w, m, ld in sorted(rows, reverse=True)[:40]:
    print(f"{k[:62]:62s} {n:4d} {us:7.1f} {a:7.1f} {s:7.1f} {l:7.1f} {w:7.1f} {m:7.1f} {ld:6.1f}")
