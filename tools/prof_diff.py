"""Per-kernel difference of two rocpd kernel traces of the training step: python tools/prof_diff.py A.db B.db [min_us]"""
import collections, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from prof_db import clean, load

thr = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
res = []
for p in sys.argv[1:3]:
    st, n = load(p, 10, 40)
    agg = collections.defaultdict(lambda: [0.0, 0.0])
    for r in st:
        a = agg[clean(r[0])]
        a[0] += (r[2] - r[1]) / n / 1e3
        a[1] += 1.0 / n
    res.append(agg)
A, B = res
print(f"kernel time per step: A {sum(v[0] for v in A.values())/1e3:.3f} ms   B {sum(v[0] for v in B.values())/1e3:.3f} ms")
for k in sorted(set(A) | set(B), key=lambda k: -abs(B[k][0] - A[k][0])):
    d = B[k][0] - A[k][0]
    if abs(d) >= thr:
        print(f"{k[:72]:72s} A {A[k][0]:8.1f} ({A[k][1]:4.1f})  B {B[k][0]:8.1f} ({B[k][1]:4.1f})  {d:+8.1f} us")
