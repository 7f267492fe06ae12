"""A/B of HIP-runtime environment switches (read when libamdhip64 loads): each variant in its own process, sustained timing (80 warm + 100
timed replays).  usage: python scratch/ab_env2.py VAR=val[,VAR2=val2] ...   ("base" = nothing set)"""
import os, subprocess, sys
code = r'''
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from textboost_amd.workload import build_step
step, _ = build_step()
step.capture(warmup=2)
for _ in range(80): step.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step.replay()
torch.cuda.synchronize()
print("%.3f ms/step  loss %.6f" % ((time.perf_counter() - t0) / 100 * 1e3, step.scalars()["loss"]))
'''
for spec in sys.argv[1:]:
    env = dict(os.environ)
    if spec != "base":
        for kv in spec.split(","):
            k, v = kv.split("=", 1); env[k] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"{spec:48s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else 'FAILED ' + out.stderr[-300:]}", flush=True)
