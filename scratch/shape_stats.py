"""per-(kernel, grid) durations from a rocprofv3 rocpd db: python scratch/shape_stats.py <db> [n]"""
import collections, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = collections.defaultdict(list)
for n, gx, gy, d in cur.execute("select name, grid_x, grid_y, end-start from kernels"):
    agg[(n.split('(')[0].replace('void ', ''), gx, gy)].append(d)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    v = sorted(v)
    print("%-40s grid %5d x %5d  calls %4d  avg %7.2f us  med %7.2f  min %7.2f" % (k[0][:40], k[1], k[2], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3))
