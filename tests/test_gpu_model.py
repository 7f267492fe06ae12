"""GPU parity of the HIP executors (UNet fwd + dgrad bwd, text encoder fwd/bwd, full optimizer step) against the
CPU oracle (oracle/) on small configs the oracle finishes in seconds.

Tolerances (fp16 MFMA operands, fp32 accumulation / statistics, vs the fp32 oracle; every check = rel-L2 AND max-abs / max|ref| AND, where
a channel axis exists, the worst per-channel rel-L2 -- tests/parity.py): UNet prediction 3e-3 / 4e-3 / 4e-3 (measured 1.2e-3), d(ehs)
5e-3 / 6e-3 / 2e-2 (1.9e-3), trainable-encoder hidden states 5e-4 (3.5e-5), encoder-only LoRA gradients 1.5e-3 (4.5e-4), whole-step
gradients through the UNet 4e-3 / 6e-3 (1-1.7e-3), fp16 teacher 1.5e-3 (3.7e-4).  The reference's own fp16 UNet, emulated by rounding every
operator result of the oracle to fp16 (oracle/fp16_mode.py), is 4.9e-3 / 9.5e-3 away from the fp32 oracle: asserted as the noise floor."""
import copy

import pytest
import torch

from parity import parity

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def round_fp16_(module):
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(p.half().float())
    return module


def make_unet(B=2, hw=16, cross_dim=64, seed=0, sd2=False, channels=None):
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from textboost_amd.unet import HipUNet, UNetGeometry
    torch.manual_seed(seed)
    cfg = UNetConfig.tiny(cross_dim)
    if channels is not None:
        cfg.block_out_channels = tuple(channels)
    if sd2:  # SD2.x structure: Linear proj_in/out, per-level head counts with a uniform head dim of 64
        cfg.use_linear_projection = True
        cfg.num_heads = (1, 2, 2, 2)
    ref = round_fp16_(UNet2DCondition(cfg))
    with torch.no_grad():   # default-init norms are identity: perturb so affine params matter
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
        round_fp16_(ref)
    geo = UNetGeometry(block_out_channels=cfg.block_out_channels, num_heads=cfg.num_heads, cross_attention_dim=cross_dim,
                       cross_attn_levels=cfg.cross_attn_levels, use_linear_projection=cfg.use_linear_projection)
    hip = HipUNet(geo, ref.state_dict(), B, hw, hw, text_len=77, device=dev)
    return ref, hip, cfg


def test_unet_forward_and_dgrad_backward_match_oracle():
    B, hw, D = 2, 16, 64
    ref, hip, cfg = make_unet(B, hw, D)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, hw, hw, generator=g).half().float()
    t = torch.tensor([17, 801])
    ehs = torch.randn(B, 77, D, generator=g).half().float().requires_grad_(True)
    pred_ref = ref(x, t, ehs)
    dpred = torch.randn(B, 4, hw, hw, generator=g)
    pred_ref.backward(dpred)
    pred = hip.forward(x.half().to(dev), t.to(dev), ehs.detach().half().view(B * 77, D).to(dev).contiguous())
    e_pred = parity("tiny UNet pred", pred, pred_ref, rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)[0]
    d_ehs = hip.backward(dpred.to(dev))
    e_dehs = parity("tiny UNet d_ehs", d_ehs.view(B, 77, D), ehs.grad, rel=5e-3, maxabs=6e-3, ch_dim=2, ch_rel=2e-2)[0]
    # the noise floor of the REFERENCE's own fp16 UNet: the same oracle with every operator result rounded to fp16 (oracle/fp16_mode.py).
    # The HIP kernels (fp32 epilogues, fused norm / activation / residual) must sit inside it.
    from oracle.fp16_mode import fp16_rounding
    ehs16 = ehs.detach().clone().requires_grad_(True)
    with fp16_rounding():
        p16 = ref(x, t, ehs16)
        p16.backward(dpred.half().float())
    f_pred, f_dehs = rel_err(p16, pred_ref), rel_err(ehs16.grad, ehs.grad)
    print(f"[parity] fp16-faithful oracle vs fp32 oracle: pred {f_pred:.3e}, d_ehs {f_dehs:.3e}")
    assert 2e-3 < f_pred < 2e-2 and e_pred < f_pred and e_dehs < f_dehs


def make_encoders(B=2, D=64, r=4, n_added=3, seed=0, act="quick_gelu"):
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from textboost_amd.text_encoder import CLIPGeometry, HipTextEncoder
    torch.manual_seed(seed)
    ccfg = CLIPTextCfg.tiny(D)
    ccfg.act = act
    base = TextBoostEncoder(ccfg, r=0)
    with torch.no_grad():
        for n, p in base.named_parameters():
            if "ln" in n or n.endswith("bias"):
                p.add_(torch.randn_like(p) * 0.1)
        base.token_embedding.weight.mul_(0.5)
        null = base.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    base.set_null_embedding(null)
    teacher = ts.make_teacher(base)
    student = TextBoostEncoder(ccfg, r=r)
    student.load_state_dict(base.state_dict(), strict=False)
    student.set_null_embedding(null)
    with torch.no_grad():
        for n, p in student.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.05)     # non-zero B so every LoRA gradient path is exercised
    added = add_tokens(student, [100, 200, 300][:n_added])
    geo = CLIPGeometry(hidden_size=D, intermediate_size=ccfg.intermediate_size, num_layers=ccfg.num_layers, num_heads=ccfg.num_heads,
                       act=act)
    sd = {hf: dict(base.named_parameters())[ours].detach() for ours, hf in base.hf_key_map().items()}
    hip = HipTextEncoder(geo, sd, B, mode="autocast", lora_rank=r, n_slots=2, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([100, 200, 300][:n_added])
    for i, layer in enumerate(student.layers):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    hip_teacher = HipTextEncoder(geo, sd, B, mode="half", lora_rank=0, device=dev)
    hip_teacher.set_null_embedding(null)
    return student, teacher, hip, hip_teacher, added, null


def lora_grads_from_oracle(student):
    gA = torch.stack([torch.cat([l.q.lora_A.grad, l.k.lora_A.grad, l.v.lora_A.grad]) for l in student.layers])
    gB = torch.stack([torch.cat([l.q.lora_B.grad, l.k.lora_B.grad, l.v.lora_B.grad]) for l in student.layers])
    return gA, gB


def test_text_encoder_forward_backward_match_oracle():
    from oracle import train_step as ts
    B, D = 3, 64
    student, teacher, hip, hip_teacher, added, null = make_encoders(B, D)
    g = torch.Generator().manual_seed(2)
    ids = ts.synthetic_ids(B, added, g)
    ids[1, 1:] = 49407   # a null prompt row -> pinned, zero gradient
    out_ref = student(ids)
    R = torch.randn(B, 77, D, generator=g)
    (out_ref * R).sum().backward()
    hip.pack_lora()
    out = hip.forward(ids.to(dev), slot=0)
    parity("tiny encoder hidden states", out.view(B, 77, D), out_ref, rel=5e-4, maxabs=1e-3, ch_dim=2, ch_rel=2e-3)
    assert torch.equal(out.view(B, 77, D)[1].cpu(), null)
    hip.zero_grad()
    hip.backward(R.view(B * 77, D).to(dev).contiguous(), slot=0)
    gA, gB = lora_grads_from_oracle(student)
    parity("tiny encoder grad lora_A", hip.grad_A, gA, rel=1.5e-3, maxabs=2e-3, ch_dim=0, ch_rel=2e-3)
    parity("tiny encoder grad lora_B", hip.grad_B, gB, rel=1.5e-3, maxabs=2e-3, ch_dim=0, ch_rel=2e-3)
    parity("tiny encoder grad added rows", hip.grad_added, student.token_embedding.weight.grad[added], rel=1e-3, maxabs=1.5e-3)
    # the fp16 teacher
    pids = ts.synthetic_ids(B, added, g, prior=True)   # the teacher never sees added tokens (49408-row table, :650)
    with torch.no_grad():
        t_ref = teacher(pids)
    t_out = hip_teacher.forward(pids.to(dev), slot=0)
    parity("tiny fp16 teacher hidden states", t_out.view(B, 77, D), t_ref, rel=1.5e-3, maxabs=5e-3)
    # the same teacher rows riding along in the student's launches (extra_ids: zero LoRA operand rows, original token table): the student rows
    # must be bit-identical to the pass without them, the extra rows equal to the frozen encoder, and the backward must not see them
    merged = hip.forward(ids.to(dev), slot=0, extra_ids=pids.to(dev), extra_table=hip_teacher.token_table.float())
    assert torch.equal(merged[:B * 77], out)
    parity("teacher rows inside the student pass", merged[B * 77:].view(B, 77, D), t_ref, rel=5e-4, maxabs=1e-3)
    gA0, gB0, gE0 = hip.grad_A.clone(), hip.grad_B.clone(), hip.grad_added.clone()
    hip.zero_grad()
    hip.backward(R.view(B * 77, D).to(dev).contiguous(), slot=0)
    assert torch.equal(hip.grad_A, gA0) and torch.equal(hip.grad_B, gB0) and torch.equal(hip.grad_added, gE0)


def build_step(B=2, hw=16, D=64, use_scaler=True, sd2=False, kpl_type="cos", mixing=None, prediction_type="epsilon"):
    from oracle import train_step as ts
    from textboost_amd.trainer import StepHyper, TextBoostStep
    ref_unet, hip_unet, _ = make_unet(B, hw, D, seed=3, sd2=sd2)
    student, teacher, hip_te, hip_teacher, added, null = make_encoders(B, D, seed=4, act="gelu" if sd2 else "quick_gelu")
    st_ref = ts.TrainState(student, teacher, ref_unet, added, ts.StepConfig(kpl_type=kpl_type, mixing=mixing, prediction_type=prediction_type))
    hp = StepHyper(use_grad_scaler=use_scaler, init_scale=65536.0 if use_scaler else 1.0, kpl_type=kpl_type, mixing=mixing,
                   prediction_type=prediction_type)
    step = TextBoostStep(hip_unet, hip_te, hip_teacher, hp, (B, 4, hw, hw), device=dev)
    step.external_noise = True
    return st_ref, step, added


def test_full_step_matches_oracle_elementwise():
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    st_ref, step, added = build_step(B, hw, D)
    assert abs(step.mean_norm - st_ref.mean_norm) < 1e-4 * st_ref.mean_norm
    g = torch.Generator().manual_seed(5)
    te_ref = st_ref.te
    for it in range(2):
        ids = ts.synthetic_ids(B, added, g)
        pids = ts.synthetic_ids(B, added, g, prior=True)
        x0 = torch.randn(B, 4, hw, hw, generator=g)
        noise = torch.randn(B, 4, hw, hw, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        w_before = te_ref.token_embedding.weight.detach().clone()
        out = st_ref.step(x0, noise, t, ids, pids)
        step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t)
        step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        step.step_eager()
        sc = step.scalars()
        assert sc["found_inf"] == 0.0
        assert abs(sc["loss_mse"] - out["mse"]) < 2e-2 * abs(out["mse"]) + 1e-4, (sc, out["mse"])
        assert abs(sc["loss_kpl"] - out["kpl"]) < 5e-2 * abs(out["kpl"]) + 1e-5, (sc, out["kpl"])
        # gradients (unscaled): flat_grad holds loss_scale * grad
        inv = 1.0 / 65536.0 if it == 0 else 1.0 / sc["loss_scale"]
        nA = step.te.lora_A.numel()
        gA = torch.stack([torch.cat(out["g_lora"][6 * l + 0: 6 * l + 6: 2]) for l in range(len(te_ref.layers))])
        gB = torch.stack([torch.cat(out["g_lora"][6 * l + 1: 6 * l + 6: 2]) for l in range(len(te_ref.layers))])
        # oracle g_lora are post-clip; compare directions through the clip coefficient
        clip = min(1.0, 1.0 / (out["lora_grad_norm"] + 1e-6))
        assert abs(sc["grad_norm"] - out["lora_grad_norm"]) < 5e-2 * out["lora_grad_norm"]
        parity(f"step {it} grad lora_A", step.te.grad_A * inv * clip, gA, rel=4e-3, maxabs=6e-3)
        parity(f"step {it} grad lora_B", step.te.grad_B * inv * clip, gB, rel=4e-3, maxabs=6e-3)
        parity(f"step {it} grad added rows", step.te.grad_added * inv, out["g_emb_added"], rel=4e-3, maxabs=6e-3)
        # parameters after the update, elementwise
        w = step.te.token_table.cpu()
        wr = te_ref.token_embedding.weight.detach()
        torch.testing.assert_close(w[:49408], wr[:49408], rtol=1e-6, atol=1e-7)       # decay-only rows
        assert (w[added] - wr[added]).abs().max().item() < 2.5e-3                     # <= ~2 * emb_lr (Adam step-1 sign flips)
        assert rel_err(w[added], wr[added]) < 2e-3
        A_ref = torch.stack([torch.cat([l.q.lora_A, l.k.lora_A, l.v.lora_A]) for l in te_ref.layers]).detach()
        assert (step.te.lora_A.cpu() - A_ref).abs().max().item() < 1.5e-4             # <= ~2 * lr
    assert step.scalars()["opt_steps"] == 2.0


def test_graph_replay_equals_eager():
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    outs = []
    for mode in ("eager", "graph"):
        st_ref, step, added = build_step(B, hw, D)
        g = torch.Generator().manual_seed(6)
        step.input_ids.copy_(ts.synthetic_ids(B, added, g)); step.prior_ids.copy_(ts.synthetic_ids(B, added, g, prior=True))
        step.x0.copy_(torch.randn(B, 4, hw, hw, generator=g)); step.noise.copy_(torch.randn(B, 4, hw, hw, generator=g))
        step.timesteps.copy_(torch.randint(0, 1000, (B,), generator=g))
        if mode == "graph":
            step.capture(warmup=2)
            step.replay(); step.replay()
        else:
            for _ in range(4):
                step.step_eager()
        torch.cuda.synchronize()
        outs.append((step.te.lora_A.clone(), step.te.lora_B.clone(), step.te.token_table[49408:].clone(), step.state.clone()))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_cli_end_to_end_writes_reference_layout(tmp_path):
    """train_textboost.py with the reference's flags on synthetic latents: 6 steps, a checkpoint, the final adapter + token files."""
    import json
    import os
    import sys
    from safetensors.torch import load_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    out = str(tmp_path / "run")
    args = T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--output_dir", out, "--train_batch_size", "2",
                         "--resolution", "128", "--max_train_steps", "6", "--checkpointing_steps", "4", "--placeholder_token", "<dog>",
                         "--augment_inversion", "--lora_rank", "4", "--mixed_precision", "fp16", "--learning_rate", "5e-5",
                         "--emb_learning_rate", "1e-3", "--seed", "42"])
    T.main(args)
    assert os.path.exists(os.path.join(out, "training.log"))
    files = set(os.listdir(out))
    assert {"text_encoder", "dog.bin", "checkpoint-4", "grayscale_0.bin", "right_2.bin", "crop.bin"} <= files
    assert len([f for f in files if f.endswith(".bin")]) == 17           # 1 placeholder + 16 augmentation vectors
    sd = load_file(os.path.join(out, "text_encoder", "adapter_model.safetensors"))
    assert len(sd) == 72 and sd["base_model.model.text_model.encoder.layers.11.self_attn.v_proj.lora_B.weight"].shape == (768, 4)
    assert any(v.abs().max() > 0 for k, v in sd.items() if "lora_B" in k)   # B left zero-init only if nothing trained
    ck = set(os.listdir(os.path.join(out, "checkpoint-4")))
    assert {"text_encoder", "model.safetensors", "optimizer.bin", "scheduler.bin", "scaler.pt", "random_states_0.pkl", "dog.bin"} <= ck
    cfg = json.load(open(os.path.join(out, "text_encoder", "adapter_config.json")))
    assert cfg["base_model_name_or_path"] == "/nonexistent/sd15"
    d = torch.load(os.path.join(out, "dog.bin"))
    assert d["<dog>"].shape == (768,) and torch.isfinite(d["<dog>"]).all()


def test_collective_paths_with_one_rank_group():
    """The N>1 code paths -- the RCCL all-reduce captured INSIDE the one step graph, and the fallback of an eager all-reduce between two
    graphs -- exercised with a 1-rank nccl group: both must equal the no-collective single-graph path bit for bit (the sum over one
    rank is the identity, and grad_div = 1)."""
    import os
    import torch.distributed as dist
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    outs = []
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        for force, single in ((False, True), (True, True), (True, False)):
            st_ref, step, added = build_step(B, hw, D)
            step.force_dist = force
            g = torch.Generator().manual_seed(8)
            step.input_ids.copy_(ts.synthetic_ids(B, added, g)); step.prior_ids.copy_(ts.synthetic_ids(B, added, g, prior=True))
            step.x0.copy_(torch.randn(B, 4, hw, hw, generator=g)); step.noise.copy_(torch.randn(B, 4, hw, hw, generator=g))
            step.timesteps.copy_(torch.randint(0, 1000, (B,), generator=g))
            step.capture(warmup=1, single_graph=single)
            assert step.graph_mode == ("single" if not force else ("single+rccl" if single else "two+eager-rccl")), step.graph_mode
            assert len(step.graph) == (1 if single else 2)
            step.replay(); step.replay()
            torch.cuda.synchronize()
            outs.append((step.te.lora_A.clone(), step.te.lora_B.clone(), step.te.token_table[49408:].clone()))
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                torch.testing.assert_close(a, b, rtol=0, atol=0)
    finally:
        if created:
            dist.destroy_process_group()


def test_capture_recovers_when_the_collective_cannot_be_captured():
    """capture(single_graph=True) with a collective that breaks the capture (here: one that synchronises the host) must fall back to the two
    graphs around an eager exchange and still train -- the failed attempt leaves torch's CUDA generator flagged as capturing, which the
    fallback has to repair before it can capture again."""
    import os
    import torch.distributed as dist
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        outs = []
        for broken in (False, True):
            st_ref, step, added = build_step(B, hw, D)
            step.force_dist = True
            g = torch.Generator().manual_seed(8)
            step.input_ids.copy_(ts.synthetic_ids(B, added, g)); step.prior_ids.copy_(ts.synthetic_ids(B, added, g, prior=True))
            step.x0.copy_(torch.randn(B, 4, hw, hw, generator=g))
            step.external_noise = False      # draw() uses the CUDA generator inside the graph: the state the failed capture leaves behind
            torch.cuda.manual_seed(1234)
            if broken:
                good = step.all_reduce
                calls = {"n": 0}

                def bad():
                    calls["n"] += 1
                    if calls["n"] == 1:      # the captured attempt: a host synchronisation is illegal inside a capture
                        step.flat_grad.sum().item()
                    good()
                step.all_reduce = bad
            with pytest.warns(UserWarning) if broken else __import__("contextlib").nullcontext():
                step.capture(warmup=0)
            assert step.graph_mode == ("two+eager-rccl" if broken else "single+rccl")
            step.replay(); step.replay()
            torch.cuda.synchronize()
            assert step.scalars()["opt_steps"] == 2.0
            outs.append((step.te.lora_A.clone(), step.te.lora_B.clone(), step.te.token_table[49408:].clone()))
        for a, b in zip(*outs):          # same seed, same noise stream: the repaired generator continues where the snapshot was taken
            torch.testing.assert_close(a, b, rtol=0, atol=0)
    finally:
        if created:
            dist.destroy_process_group()


def test_sd2_style_models_kpl_mse_and_mixing_match_oracle():
    """BASELINE config 4 structure at small size: Linear proj_in/out + hd 64 UNet, erf-GELU text MLP, plus --kpl_type mse and
    --mixing (object): gradients and updated weights vs the oracle."""
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    st_ref, step, added = build_step(B, hw, D, sd2=True, kpl_type="mse", mixing="object")
    g = torch.Generator().manual_seed(9)
    ids = ts.synthetic_ids(B, added, g); pids = ts.synthetic_ids(B, added, g, prior=True)
    x0 = torch.randn(B, 4, hw, hw, generator=g); noise = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    out = st_ref.step(x0, noise, t, ids, pids)
    step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t); step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
    step.step_eager()
    sc = step.scalars()
    assert sc["found_inf"] == 0.0
    assert abs(sc["loss_mse"] - out["mse"]) < 2e-2 * abs(out["mse"]) + 1e-4
    assert abs(sc["loss_kpl"] - out["kpl"]) < 5e-2 * abs(out["kpl"]) + 1e-6
    te_ref = st_ref.te
    clip = min(1.0, 1.0 / (out["lora_grad_norm"] + 1e-6))
    inv = 1.0 / 65536.0
    gB = torch.stack([torch.cat(out["g_lora"][6 * l + 1: 6 * l + 6: 2]) for l in range(len(te_ref.layers))])
    parity("mixing step grad lora_B", step.te.grad_B * inv * clip, gB, rel=4e-3, maxabs=6e-3)
    gBv = step.te.grad_B.view(len(te_ref.layers), 3, D, 4)
    assert gBv[:, :, 1::2].abs().max().item() == 0.0 and gBv[:, :, 0::2].abs().max().item() > 0.0   # :1119-1126, object
    parity("mixing step grad added rows", step.te.grad_added * inv, out["g_emb_added"], rel=4e-3, maxabs=6e-3)


def test_v_prediction_target_matches_oracle():
    """SD2.1-768 style `prediction_type="v_prediction"` (:1070-1075): target = noise_scheduler.get_velocity(x0, noise, t)."""
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    st_ref, step, added = build_step(B, hw, D, sd2=True, prediction_type="v_prediction")
    g = torch.Generator().manual_seed(10)
    ids = ts.synthetic_ids(B, added, g); pids = ts.synthetic_ids(B, added, g, prior=True)
    x0 = torch.randn(B, 4, hw, hw, generator=g); noise = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([37, 911])
    out = st_ref.step(x0, noise, t, ids, pids)
    step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t); step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
    step.step_eager()
    sc = step.scalars()
    torch.testing.assert_close(step.velocity.cpu(), ts.get_velocity(x0, noise, t, st_ref.acp), rtol=1e-5, atol=1e-6)
    assert sc["found_inf"] == 0.0 and abs(sc["loss_mse"] - out["mse"]) < 2e-2 * abs(out["mse"]) + 1e-4, (sc, out["mse"])
    parity("v-prediction step grad added rows", step.te.grad_added / 65536.0, out["g_emb_added"], rel=4e-3, maxabs=6e-3)


def _fill_inputs(step, added, B, hw, seed, rows=None):
    """deterministic step inputs; `rows` selects a slice of a larger global batch (data-parallel shard)"""
    from oracle import train_step as ts
    g = torch.Generator().manual_seed(seed)
    n = B if rows is None else rows[1]
    ids, pids = ts.synthetic_ids(n, added, g), ts.synthetic_ids(n, added, g, prior=True)
    x0, noise = torch.randn(n, 4, hw, hw, generator=g), torch.randn(n, 4, hw, hw, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    sl = slice(0, n) if rows is None else slice(rows[0], rows[0] + B)
    step.input_ids.copy_(ids[sl]); step.prior_ids.copy_(pids[sl]); step.x0.copy_(x0[sl]); step.noise.copy_(noise[sl])
    step.timesteps.copy_(t[sl])


def test_data_parallel_equivalence_of_the_gradient_exchange():
    """DDP semantics without a second GPU: two ranks with per-rank batch B (each normalising its losses over its OWN B samples, as the
    reference does under DDP) whose flat gradients are SUMMED and divided by W -- folded into the unscale coefficient by
    tb_scaler_update(grad_div=W) -- must give the gradient, the clip norm and the parameter update of ONE step over the 2B samples."""
    from textboost_amd import _lib as L
    B, hw, D, W = 2, 16, 64, 2
    _, big, added = build_step(W * B, hw, D)
    _fill_inputs(big, added, W * B, hw, seed=21)
    big.forward_backward()
    shards = []
    for r in range(W):
        _, st, _ = build_step(B, hw, D)
        _fill_inputs(st, added, B, hw, seed=21, rows=(r * B, W * B))
        st.forward_backward()
        shards.append(st)
    summed = shards[0].flat_grad + shards[1].flat_grad
    e = rel_err(summed / W, big.flat_grad)
    assert e < 5e-3, f"mean of the shard gradients vs the 2B-sample gradient: rel err {e}"
    # the step as rank 0 runs it: summed buffer + world = 2, against the big step with world = 1
    r0 = shards[0]
    r0.flat_grad.copy_(summed)
    r0.world = W
    r0.optimizer_step(); big.optimizer_step()
    torch.cuda.synchronize()
    s0, sb = r0.scalars(), big.scalars()
    assert s0["found_inf"] == 0.0 and abs(s0["grad_norm"] - sb["grad_norm"]) < 5e-3 * sb["grad_norm"], (s0, sb)
    assert abs(r0.state[L.ST_COEF_EMB].item() * W - big.state[L.ST_COEF_EMB].item()) < 1e-12
    # Adam's first step is lr * sign(g) (+ decay): identical up to sign flips of near-zero gradients -> compare the update direction
    dA0, dAb = r0.m_lora, big.m_lora
    assert rel_err(dA0, dAb) < 5e-3
    torch.testing.assert_close(r0.te.token_table[:49408], big.te.token_table[:49408], rtol=0, atol=0)   # decay-only rows: bit-equal


def test_resume_from_checkpoint_is_bit_exact_under_an_lr_schedule(tmp_path):
    """with a non-constant --lr_scheduler the never-updated embedding rows decay by a different factor every step; the
    checkpoint therefore carries the whole table.  save after 2 steps -> fresh trainer -> load -> 2 more steps == 4 uninterrupted steps."""
    from textboost_amd import checkpoint as ckpt
    from textboost_amd.trainer import lr_lambda
    B, hw, D = 2, 16, 64
    lam = lr_lambda("linear", 2, 6)
    table = [lam(k) for k in range(7)]

    def fresh():
        _, st, added = build_step(B, hw, D)
        st.set_lr_table(table)
        return st, added

    ref, added = fresh()
    for it in range(4):
        _fill_inputs(ref, added, B, hw, seed=30 + it)
        ref.step_eager()
    a, _ = fresh()
    for it in range(2):
        _fill_inputs(a, added, B, hw, seed=30 + it)
        a.step_eager()
    ckpt.save_trainer_state(a, str(tmp_path / "checkpoint-2"))
    b, _ = fresh()
    ckpt.load_trainer_state(b, str(tmp_path / "checkpoint-2"))
    for it in range(2, 4):
        _fill_inputs(b, added, B, hw, seed=30 + it)
        b.step_eager()
    torch.cuda.synchronize()
    assert ref.scalars()["opt_steps"] == 4.0 and b.scalars()["opt_steps"] == 4.0
    for x, y in ((ref.te.token_table, b.te.token_table), (ref.te.lora_A, b.te.lora_A), (ref.te.lora_B, b.te.lora_B),
                 (ref.m_lora, b.m_lora), (ref.v_emb, b.v_emb), (ref.state, b.state)):
        torch.testing.assert_close(x, y, rtol=0, atol=0)
    # and the schedule really was non-constant: rows below first_added decayed by prod(1 - emb_lr * wd * lambda(k)), k = 0..3
    hp = ref.hp
    expect = 1.0
    for k in range(4):
        expect *= 1.0 - hp.emb_lr * hp.wd * table[k]
    _, untouched, _ = build_step(B, hw, D)
    ratio = (ref.te.token_table[100] / untouched.te.token_table[100]).mean().item()
    assert abs(ratio - expect) < 1e-6 and abs(ratio - (1.0 - hp.emb_lr * hp.wd) ** 4) > 1e-7


def test_split_text_encoder_schedule_equals_the_merged_one():
    """round 3: instance rows / prior + teacher rows of the text encoder as concurrent branches beside the UNet (trainer `split_te`) against all
    rows in one pass on the main stream: the same losses and the same gradients (row-wise arithmetic is identical; only the fp32 order of the
    LoRA-gradient row sums and the two-buffer accumulation differ), eagerly and as a captured graph."""
    B, hw, D = 2, 16, 64
    res = {}
    for split in (False, True):
        _, step, added = build_step(B, hw, D)
        assert step.merge_teacher
        step.split_te = split
        step.te_fwd_side = False
        _fill_inputs(step, added, B, hw, seed=33)
        step.forward_backward()
        torch.cuda.synchronize()
        res[split] = (step.flat_grad.clone(), step.state.clone(), step.d_all.clone())
        if split:  # graph replay of the three-branch step == its eager run, bit for bit
            _, st2, _ = build_step(B, hw, D)
            st2.split_te = True
            _fill_inputs(st2, added, B, hw, seed=33)
            st2.capture(warmup=0)
            st2.replay()
            _, st3, _ = build_step(B, hw, D)
            st3.split_te = True
            _fill_inputs(st3, added, B, hw, seed=33)
            st3.step_eager()
            torch.cuda.synchronize()
            for a, b in ((st2.flat_grad, st3.flat_grad), (st2.te.lora_A, st3.te.lora_A), (st2.te.token_table[49408:], st3.te.token_table[49408:]),
                         (st2.state, st3.state)):
                torch.testing.assert_close(a, b, rtol=0, atol=0)
    (g0, s0, d0), (g1, s1, d1) = res[False], res[True]
    torch.testing.assert_close(s1, s0, rtol=1e-5, atol=1e-7)        # both losses
    assert rel_err(d1, d0) < 1e-5, rel_err(d1, d0)                  # d ehs and d prior rows: the same arithmetic per row (tile choice may follow M)
    assert rel_err(g1, g0) < 1e-5, rel_err(g1, g0)


def test_encoder_forward_as_a_branch_beside_the_unet_head_is_bit_equal():
    """`te_fwd_side` (opt-in A/B knob): the merged encoder forward on a second stream, joined in front of the hoisted K/V projection -- the same
    kernels on the same data, so gradients and losses are bit-equal to the single-stream order, eagerly and replayed."""
    B, hw, D = 2, 16, 64
    res = []
    for side, graph in ((False, False), (True, False), (True, True)):
        _, step, added = build_step(B, hw, D)
        assert step.merge_teacher and not step.te_fwd_side   # opt-in (measured slower: DESIGN section 4)
        step.te_fwd_side = side
        _fill_inputs(step, added, B, hw, seed=34)
        if graph:
            step.capture(warmup=0)
            step.replay()
        else:
            step.step_eager()
        torch.cuda.synchronize()
        res.append((step.flat_grad.clone(), step.state.clone(), step.te.lora_A.clone(), step.te.token_table[49408:].clone()))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_encoder_forward_issued_behind_the_unet_head_is_bit_equal():
    """`te_fwd_late` (round 5, default): the encoder forward issued on the SAME stream where the UNet first needs the text states (in front of the
    hoisted K/V projection) instead of in front of the UNet -- an issue-order change only, so gradients, losses and the updated parameters are
    bit-equal to the reference's order (`train_textboost.py:1054-1067`), eagerly and replayed."""
    B, hw, D = 2, 16, 64
    res = []
    for late, graph in ((False, False), (True, False), (True, True)):
        _, step, added = build_step(B, hw, D)
        assert step.te_fwd_late
        step.te_fwd_late = late
        _fill_inputs(step, added, B, hw, seed=35)
        if graph:
            step.capture(warmup=0)
            step.replay()
        else:
            step.step_eager()
        torch.cuda.synchronize()
        res.append((step.flat_grad.clone(), step.state.clone(), step.te.lora_A.clone(), step.te.token_table[49408:].clone(), step.pred.clone()))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            torch.testing.assert_close(a, b, rtol=0, atol=0)


@pytest.mark.parametrize("mode", ["autocast", "fp32"])
def test_hip_text_encoder_on_the_reference_generated_fixture(mode):
    """tests/golden/clip_textboost_tiny.pt was produced by the REFERENCE's own TextBoostModel (/root/reference/textboost/text_encoder.py:34-87,
    generator tests/golden/make_golden.py): forward incl. both pins, and the gradients of sum(out * R).  The HIP encoder is run on the same
    weights / ids directly (until round 4 the fixture only pinned the CPU oracle, so the HIP <-> reference link was transitive): hidden states,
    the gradient of every token-embedding row the prompts use (from the returned input gradient) and the added-token rows' accumulated
    gradient (what the optimizer consumes).  autocast = the fp16 mixed-precision arithmetic of the metric; fp32 = the no-AMP mode."""
    import os
    from textboost_amd.text_encoder import CLIPGeometry, HipTextEncoder
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_textboost_tiny.pt"))
    c = g["cfg"]
    V0 = 49408                                  # the fixture's table already carries 3 added rows (49408..49410)
    n_added = c["V"] - V0
    emb = torch.zeros(c["V"], c["D"])
    emb[g["emb_rows_idx"]] = g["emb_rows"]
    sd = {"text_model." + k: v for k, v in g["state_dict"].items()}
    sd["text_model.embeddings.token_embedding.weight"] = emb[:V0]
    geo = CLIPGeometry(vocab_size=V0, hidden_size=c["D"], intermediate_size=c["I"], num_layers=c["L"], num_heads=c["H"])
    ids = g["ids"]
    B, T, D = ids.shape[0], ids.shape[1], c["D"]
    hip = HipTextEncoder(geo, sd, B, mode=mode, lora_rank=0, device=dev)
    new_ids = hip.add_tokens([0] * n_added)
    assert new_ids == list(range(V0, c["V"]))
    hip.token_table[V0:] = emb[V0:].to(hip.token_table)
    hip.set_null_embedding(g["null"])
    out = hip.forward(ids.to(dev), slot=0)
    tol = dict(rel=5e-4, maxabs=1e-3, ch_rel=2e-3) if mode == "autocast" else dict(rel=2e-6, maxabs=5e-6, ch_rel=5e-6)
    parity(f"reference fixture, hidden states ({mode})", out.view(B, T, D), g["out"], ch_dim=2, **tol)
    assert torch.equal(out.view(B, T, D)[:, 0].cpu(), g["null"][0].expand(B, -1))   # text_encoder.py:81-86
    assert torch.equal(out.view(B, T, D)[2].cpu(), g["null"])                        # :71-79 (row 2 is the null prompt)
    hip.zero_grad()
    d_in = hip.backward(g["R"].reshape(B * T, D).to(dev).contiguous(), slot=0)       # gradient w.r.t. the embedded inputs, per position
    flat = ids.reshape(-1).to(dev)
    g_rows = torch.zeros(c["V"], D, device=dev).index_add_(0, flat, d_in.float())[g["emb_rows_idx"].to(dev)]
    gtol = dict(rel=1.5e-3, maxabs=2e-3) if mode == "autocast" else dict(rel=5e-6, maxabs=1e-5)
    parity(f"reference fixture, token-row gradients ({mode})", g_rows, g["g_emb_rows"], **gtol)
    added_pos = [int((g["emb_rows_idx"] == t).nonzero()) for t in new_ids]
    parity(f"reference fixture, added-row gradients ({mode})", hip.grad_added, g["g_emb_rows"][added_pos], **gtol)
