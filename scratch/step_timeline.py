"""One graph replay of the step as a timeline from a rocprofv3 rocpd sqlite: index, start offset, duration, grid, kernel name -- the last
replay that has > 1000 kernels.  usage: python scratch/step_timeline.py <db> [out.txt]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {gx}" if gx else ", 0") + (f", {wx}" if wx else ", 1")
rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
# a step = the launches between two consecutive launches of the step's last kernel (opt_apply_kernel; renorm_rows_kernel until round 5); graph replays are the steps whose kernels abut
# (span close to the sum of the durations) -- the last such step is printed whole (round 5: independent of the > 100 us gaps inside a replay)
ends = [i for i, r in enumerate(rows) if 'opt_apply_kernel' in r[0] or 'renorm_rows_kernel' in r[0]]   # (round 6: the tail's last kernel is opt_apply_kernel)
steps = [rows[a + 1:b + 1] for a, b in zip(ends, ends[1:])]
steps = [x for x in steps if len(x) > 300 and not any('spin_kernel' in r[0] or 'mfma_peak' in r[0] for r in x)]   # (not the eager roofline leg)
tight = [x for x in steps if (x[-1][2] - x[0][1]) < 1.10 * sum(r[2] - r[1] for r in x)]
g = (tight or steps)[-1]
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
