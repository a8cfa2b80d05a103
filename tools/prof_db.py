"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as rocpd sqlite (run_results.db):
python tools/prof_db.py <db> [first_step last_step] -- steps are delimited by pack_kernel launches."""
import collections, re, sqlite3, sys

def clean(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)

def load(path, lo=None, hi=None):
    c = sqlite3.connect(path)
    rows = c.execute("select name,start,end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "pack_kernel" in r[0]]
    lo = 2 if lo is None else lo
    hi = len(idx) - 1 if hi is None else hi
    return rows[idx[lo]:idx[hi]], hi - lo

if __name__ == "__main__":
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else None
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else None
    st, n = load(sys.argv[1], lo, hi)
    agg = collections.defaultdict(lambda: [0.0, 0.0])
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for r in st:
        for d, k in ((agg, clean(r[0])), (fam, re.split(r"[<(]", clean(r[0]))[0])):
            d[k][0] += (r[2] - r[1]) / n
            d[k][1] += 1.0 / n
    tot = sum(v[0] for v in agg.values())
    print(f"# {sys.argv[1]}: {n} steps, kernel time {tot/1e6:.3f} ms/step, {sum(v[1] for v in agg.values()):.0f} launches/step")
    print(f"{'kernel template (all instantiations)':70s} {'calls':>6s} {'ms/step':>8s} {'avg us':>8s} {'%':>6s}")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"{k[:70]:70s} {v[1]:6.1f} {v[0]/1e6:8.3f} {v[0]/v[1]/1e3:8.1f} {100*v[0]/tot:6.1f}")
    print()
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"{k[:70]:70s} {v[1]:6.1f} {v[0]/1e6:8.3f} {v[0]/v[1]/1e3:8.1f} {100*v[0]/tot:6.1f}")
