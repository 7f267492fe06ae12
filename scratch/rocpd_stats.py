"""Summarise a rocprofv3 rocpd sqlite (--kernel-trace) into a per-kernel table: calls, total ms, avg us, %."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels").fetchall() if "start" in cols else None
if rows is None:
    print(cols); sys.exit(1)
agg = {}
for name, s, e in rows:
    if "spin_kernel" in name:  # bench.py's roofline leg queues a spin kernel ahead of its recorded step: not part of the workload
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0] if not n.startswith("at::") else n[:90]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{k[:80]:80s} {v[0]:7d} {v[1]/1e6:10.3f} {v[1]/v[0]/1e3:9.2f} {100*v[1]/tot:6.2f}")
print(f"{'TOTAL':80s} {sum(v[0] for v in agg.values()):7d} {tot/1e6:10.3f}")
