"""Per-kernel HBM table: rocprofv3 kernel trace (durations) x PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
MI355X_MICROARCH.md prescribes) -> GB/s per kernel instantiation against the 8 TB/s roof.
usage: python tools/per_kernel_roofline.py <run_results.db> <pmc_hbm_traffic.json>"""
import collections, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_db import clean, load

MFMA = ("igemm_kernel", "pw_kernel", "wgrad", "dgrad2", "attn_", "pose_", "conv1_")
st, n = load(sys.argv[1])
pm = json.load(open(sys.argv[2]))
agg = collections.defaultdict(lambda: [0.0, 0.0])
for r in st:
    a = agg[clean(r[0])]
    a[0] += (r[2] - r[1]) / n
    a[1] += 1.0 / n
tot = sum(v[0] for v in agg.values())
print(f"# {n} steps, kernel time {tot/1e6:.3f} ms/step; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch, separate PMC passes")
print(f"{'kernel':66s} {'calls':>5s} {'avg us':>7s} {'ms/step':>8s} {'MB/launch':>9s} {'GB/s':>6s} {'of 8TB/s':>8s}  uses MFMA")
tb = tm = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    us = v[0] / v[1] / 1e3
    b = pm.get(k, {}).get("hbm_bytes_per_dispatch")
    if b is None:
        continue
    gbs = b / (us * 1e-6) / 1e9
    tb += b * v[1]; tm += v[0]
    if v[0] / tot < 0.002:
        continue
    print(f"{k[:66]:66s} {v[1]:5.0f} {us:7.1f} {v[0]/1e6:8.3f} {b/1e6:9.1f} {gbs:6.0f} {gbs/8000:8.1%}  {'yes' if k.startswith(MFMA) else 'no'}")
print(f"# whole step: {tb/1e9:.2f} GB of HBM traffic in {tm/1e6:.3f} ms of kernel time = {tb/tm:.0f} GB/s = {tb/tm/8000:.1%} of 8 TB/s")
