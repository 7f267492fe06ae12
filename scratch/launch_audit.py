"""every recorded launch of one eager step with its algorithmic FLOP / bytes and event time: launches sorted by time in excess of a simple
floor (max(FLOP / 1.0 PFLOP/s, bytes / 4 TB/s) + 5 us) -- a list of where to look for dispatch gaps"""
import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
from textboost_amd import ops
step, _ = build_step()
for _ in range(2): step.step_eager()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); torch.cuda._sleep(2_000_000); t1.record(); torch.cuda.synchronize()
cpm = 2_000_000 / max(t0.elapsed_time(t1), 1e-3)
ops.start_recording()
torch.cuda._sleep(int(cpm * 600.0))
emp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
for a, b in emp: a.record(); b.record()
step.step_eager(); torch.cuda.synchronize()
rec = ops.stop_recording()
ov = sorted(a.elapsed_time(b) for a, b in emp)[32]
rows = []
for name, fl, by, e0, e1 in rec:
    t = max(e0.elapsed_time(e1) - ov, 1e-4) * 1e3   # us
    floor = max(fl / 1.0e15, by / 4.0e12) * 1e6 + 5.0
    rows.append((t - floor, t, floor, fl, by, name))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for ex, t, floor, fl, by, name in rows:
    k = (name, round(fl / 1e9, 1), round(by / 1e6, 1))
    agg[k][0] += 1; agg[k][1] += t; agg[k][2] += ex
print(f"recorded launches {len(rows)}, total {sum(r[1] for r in rows)/1e3:.2f} ms, total excess {sum(r[0] for r in rows)/1e3:.2f} ms")
for (name, gf, mb), (n, t, ex) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:90]:
    print(f"{n:3d} x {t/n:7.1f} us  excess {ex:8.1f} us total  {gf:8.1f} GF {mb:7.1f} MB  ({gf/ (t/n) /1e3 if gf else 0:5.2f} PF/s, {mb/(t/n)/1e3:5.2f} TB/s)  {name}")
