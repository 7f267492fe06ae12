"""A/B inside one process: graph-replayed UNet fwd+bwd and whole step under alternating kernel knobs.
usage: ab_step.py name:variant[,variant..] ...   variants are tb_gemm_set_variant codes or g8:<bits>"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
from textboost_amd import ops, _lib as L
lib = L.lib()
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(2): step.step_eager()
def apply(codes):
    for c in codes:
        if c.startswith("g8:"): lib.tb_gemm8_set(int(c[3:]))
        elif c.startswith("attn:"): lib.tb_attention_set_variant(int(c[5:]))
        elif c.startswith("gn:"): lib.tb_groupnorm_set_variant(int(c[3:]))
        elif c.startswith("defer:"): ops.DEFER_SPLITK = bool(int(c[6:]))
        elif c.startswith("lora:"): lib.tb_lora_set_variant(int(c[5:]))
        elif c.startswith("bc:"): lib.tb_boundary_conv_set_variant(int(c[3:]))
        elif c.startswith("chain:"):
            import textboost_amd.text_encoder as _te; _te.CHAIN_LORA_BWD = bool(int(c[6:]))
        elif c.startswith("env:"):
            k, v = c[4:].split("=", 1); os.environ[k] = v
        else: lib.tb_gemm_set_variant(int(c))
configs = []
for a in sys.argv[1:]:
    name, codes = a.split(":", 1)
    configs.append((name, codes.split(",")))
graphs = {}
for name, codes in configs:
    apply(codes)
    step.step_eager(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step.draw(); step.forward_backward(); step.optimizer_step()
    graphs[name] = g
res = {n: [] for n, _ in configs}
SUSTAIN = os.environ.get("TB_AB_SUSTAIN", "1") == "1"   # the driver's bench is a 250-step continuous run: the chip settles at a lower clock than in
                                                        # short bursts, and an A/B has to be taken in that state (round 4: a -1.03 ms burst gain was -0.4 ms sustained)
if SUSTAIN:
    for rnd in range(3):
        for name, _ in configs:
            g = graphs[name]
            for _ in range(80): g.replay()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(100): g.replay()
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / 100)
for rnd in range(0 if SUSTAIN else 5):
    for name, _ in configs:
        g = graphs[name]
        g.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): g.replay()
        e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 10)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:24s} median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f}  ({1000/v[len(v)//2]:.2f} steps/s)")
