"""Builds textboost_amd/csrc/*.hip into the in-tree C-ABI library textboost_amd/libtextboost_hip.so for gfx950.

hipcc cross-compiles without a GPU. Objects are rebuilt only when their source (or a header) is newer."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# kernel-tuning A/B: TB_LIB_SUFFIX=_alt TB_CFLAGS="-DTB_SOMETHING=1" builds / loads a second library next to the default one
SUFFIX = os.environ.get("TB_LIB_SUFFIX", "")
LIB = os.path.join(PKG, f"libtextboost_hip{SUFFIX}.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-munsafe-fp-atomics"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


LIB_BF16 = os.path.join(PKG, f"libtextboost_hip_bf16{SUFFIX}.so")


def build(force: bool = False, verbose: bool = True) -> str:
    """both libraries: the fp16 one (returned) and, from the same sources with -DTB_BF16, the bfloat16 one of --mixed_precision bf16"""
    lib = _build_one(LIB, os.path.join(PKG, "build" + SUFFIX), [], force, verbose)
    if os.environ.get("TB_SKIP_BF16", "0") != "1":
        _build_one(LIB_BF16, os.path.join(PKG, "build_bf16" + SUFFIX), ["-DTB_BF16"], force, verbose)
    return lib


def _build_one(LIB: str, objdir: str, defs, force: bool, verbose: bool) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    newest_hdr = max([os.path.getmtime(h) for h in hdrs] + [0.0])
    extra = list(defs) + os.environ.get("TB_CFLAGS", "").split()
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            jobs.append([hipcc, *FLAGS, *extra, "-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    if verbose:
        print(f"[textboost_amd.build] {len(jobs)} object(s) rebuilt -> {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
