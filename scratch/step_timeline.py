"""One graph replay of the step as a timeline from a rocprofv3 rocpd sqlite: index, start offset, duration, grid, kernel name -- the last
replay that has > 1000 kernels.  usage: python scratch/step_timeline.py <db> [out.txt]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {gx}" if gx else ", 0") + (f", {wx}" if wx else ", 1")
rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
groups, g = [], [rows[0]]
for prev, r in zip(rows, rows[1:]):
    if r[1] - prev[2] > 100_000: groups.append(g); g = []
    g.append(r)
groups.append(g)
big = [x for x in groups if len(x) > 600]
tight = [x for x in big if (x[-1][2] - x[0][1]) < 1.08 * sum(r[2] - r[1] for r in x)]   # graph replays (kernels abut), not the eager roofline leg
g = (tight or big)[-1]
ends = [i for i, r in enumerate(g) if 'renorm_rows_kernel' in r[0]]   # (consecutive replays less than 100 us apart land in one group: keep the last whole step)
if len(ends) >= 2: g = g[ends[-2] + 1:ends[-1] + 1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0 = g[0][1]
def short(n):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:70]
print(f"# {len(g)} kernels, span {(g[-1][2] - t0) / 1e6:.3f} ms, sum {sum(r[2] - r[1] for r in g) / 1e6:.3f} ms; columns: idx start_us dur_us gap_us wgs name", file=out)
prev_end = t0
for i, (name, s, e, gxv, wxv) in enumerate(g):
    wgs = (gxv // max(wxv, 1)) if gxv else 0
    print(f"{i:5d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:6.2f} {wgs:7d}  {short(name)}", file=out)
    prev_end = e
