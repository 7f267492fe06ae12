"""MFMA-busy and instruction mix of the MFMA kernels from the SQ counter pass (profiles/r02_pmc_sq.txt):
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles, = 32 x N for 32x32x16 MFMAs: MI355X_MICROARCH.md)"""
import json, re, sys
txt = open(sys.argv[1]).read()
out = {}
print(f"{'kernel (dispatches averaged)':66s} {'kcycles':>8s} {'MFMA-busy':>9s} {'MFMA M':>7s} {'VALU/MFMA':>9s} {'coexec':>6s} {'wait':>5s}")
for b in re.split(r'\n(?=\S)', txt):
    lines = b.strip().split('\n')
    c = {}
    for l in lines[1:]:
        f = l.split()
        if len(f) == 2: c[f[0]] = float(f[1])
    if not c.get('SQ_INSTS_MFMA') or not c.get('SQ_BUSY_CYCLES'): continue
    cyc = c['SQ_BUSY_CYCLES'] / 32
    out[lines[0].split("  (x")[0]] = {"mfma_busy": round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), 4), "kernel_kcycles": round(cyc / 1e3),
                                      "valu_per_mfma": round(c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA'], 2), "wait_share": round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 3)}
    print(f"{lines[0][:66]:66s} {cyc/1e3:8.0f} {100*c['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc):8.1f}% {c['SQ_INSTS_MFMA']/1e6:7.2f} "
          f"{c['SQ_INSTS_VALU']/c['SQ_INSTS_MFMA']:9.2f} {100*c['SQ_VALU_MFMA_COEXEC_CYCLES']/c['SQ_VALU_MFMA_BUSY_CYCLES']:5.0f}% "
          f"{100*c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:4.0f}%")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
