"""A/B of an environment switch read at import / construction time: each variant runs in its own process (fresh import), alternating, on the
same box.  usage: python scratch/ab_env.py TB_FUSE_LN 0 1 [rounds]"""
import os, subprocess, sys
var, vals = sys.argv[1], sys.argv[2:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
code = r'''
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from textboost_amd.workload import build_step
step, _ = build_step()
step.capture(warmup=2)
for _ in range(5): step.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(60): step.replay()
torch.cuda.synchronize()
print("%.3f ms/step  loss %.6f" % ((time.perf_counter() - t0) / 60 * 1e3, step.scalars()["loss"]))
'''
for r in range(rounds):
    for v in vals:
        env = dict(os.environ); env[var] = v
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"{var}={v}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]}", flush=True)
