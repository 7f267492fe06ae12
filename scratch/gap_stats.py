"""kernel-boundary accounting of graph replays from a rocprofv3 rocpd sqlite: per replay the number of kernels, the sum of their durations, the
span, and the idle time between consecutive kernels (same queue)"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = [r for r in rows if "spin_kernel" not in r[0]]
# replays: split where the idle gap exceeds 100 us
groups, cur_g = [], [rows[0]]
for prev, r in zip(rows, rows[1:]):
    if r[1] - prev[2] > 100_000: groups.append(cur_g); cur_g = []
    cur_g.append(r)
groups.append(cur_g)
big = [g for g in groups if len(g) > 1000]
print(f"{len(groups)} groups, {len(big)} with > 1000 kernels")
for g in big[-4:]:
    n = len(g); dur = sum(e - s for _, s, e in g); span = g[-1][2] - g[0][1]
    gaps = [max(0, b[1] - a[2]) for a, b in zip(g, g[1:])]
    ov = sum(max(0, a[2] - b[1]) for a, b in zip(g, g[1:]))
    gs = sorted(gaps)
    print(f"replay: {n} kernels, sum of durations {dur/1e6:.3f} ms, span {span/1e6:.3f} ms, idle between kernels {sum(gaps)/1e6:.3f} ms "
          f"(median gap {gs[n//2]/1e3:.2f} us, p90 {gs[int(n*0.9)]/1e3:.2f} us, max {gs[-1]/1e3:.1f} us), overlap {ov/1e6:.3f} ms")
for a, b in zip(big[-5:], big[-4:]):
    print(f"between replays: {(b[0][1] - a[-1][2]) / 1e3:.1f} us idle (first kernel of the next replay: {b[0][0][:40]})")
g = big[-1]
after = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(g, g[1:]):
    k = a[0].split("(")[0][-60:]
    after[k][0] += 1; after[k][1] += max(0, b[1] - a[2])
print("idle time FOLLOWING each kernel (top):")
for k, v in sorted(after.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {v[1]/1e3:8.1f} us over {v[0]:4d} launches ({v[1]/v[0]/1e3:5.2f} us each)  {k}")
