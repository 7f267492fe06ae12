"""GPU edge cases: degenerate shapes through the C-ABI, all-null prompts, fp16 overflow -> GradScaler skip (accelerate semantics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def test_degenerate_gemm_and_attention_shapes():
    from textboost_amd import ops
    torch.manual_seed(0)
    for M, N, K in [(1, 8, 64), (3, 1, 128), (129, 257, 192), (64, 64, 64)]:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / 8).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        ops.gemm(A, W, out)
        assert rel_err(out, A.float() @ W.float().T) < 3e-3, (M, N, K)
    for Sq, Skv, hd, causal in [(1, 1, 64, False), (1, 77, 40, False), (5, 3, 8, False), (77, 77, 64, True), (130, 1, 160, False)]:
        B, H = 2, 2
        C = H * hd
        q = torch.randn(B * Sq, C, device=dev).half(); k = torch.randn(B * Skv, C, device=dev).half(); v = torch.randn(B * Skv, C, device=dev).half()
        o = torch.empty_like(q); lse = torch.empty(B, H, Sq, device=dev)
        ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd, causal=causal)
        qh, kh, vh = [t.float().view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v)]
        s = qh @ kh.transpose(-1, -2) * hd ** -0.5
        if causal:
            s = s + torch.full_like(s[0, 0], float("-inf")).triu(1)
        ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * Sq, C)
        assert rel_err(o, ref) < 3e-3, (Sq, Skv, hd)
        do = torch.randn_like(q); delta = torch.empty_like(lse)
        dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Sq, Skv, hd, causal=causal)
        assert torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all()


def test_all_null_prompts_give_zero_encoder_gradient():
    from tests.test_gpu_model import make_encoders
    B, D = 2, 64
    student, teacher, hip, hip_teacher, added, null = make_encoders(B, D)
    ids = torch.full((B, 77), 49407, dtype=torch.int64)
    ids[:, 0] = 49406
    hip.pack_lora()
    out = hip.forward(ids.to(dev)).view(B, 77, D)
    assert torch.equal(out[0].cpu(), null) and torch.equal(out[1].cpu(), null)        # text_encoder.py:71-79
    hip.zero_grad()
    hip.backward(torch.randn(B * 77, D, device=dev))
    assert hip.grad_A.abs().max() == 0 and hip.grad_B.abs().max() == 0 and hip.grad_added.abs().max() == 0


def test_fp16_overflow_skips_the_step_and_halves_the_scale():
    """accelerate fp16 semantics: inf/nan gradients -> optimizer.step skipped, scale *= 0.5, no state change."""
    from oracle import train_step as ts
    from tests.test_gpu_model import build_step
    B, hw, D = 2, 16, 64
    st_ref, step, added = build_step(B, hw, D)
    g = torch.Generator().manual_seed(12)
    step.input_ids.copy_(ts.synthetic_ids(B, added, g)); step.prior_ids.copy_(ts.synthetic_ids(B, added, g, prior=True))
    step.noise.copy_(torch.randn(B, 4, hw, hw, generator=g)); step.timesteps.copy_(torch.randint(0, 1000, (B,), generator=g))
    step.x0.copy_(torch.randn(B, 4, hw, hw, generator=g))
    step.step_eager()
    before = (step.te.lora_A.clone(), step.te.lora_B.clone(), step.te.token_table.clone(), step.m_lora.clone())
    assert step.scalars()["opt_steps"] == 1.0
    step.x0.fill_(1e6)   # overflows fp16 inside the UNet -> non-finite gradients
    step.step_eager()
    sc = step.scalars()
    assert sc["found_inf"] == 1.0 and sc["opt_steps"] == 1.0 and sc["loss_scale"] == 32768.0
    first = step.te.first_added
    assert torch.equal(before[0], step.te.lora_A) and torch.equal(before[1], step.te.lora_B) and torch.equal(before[3], step.m_lora)
    assert torch.equal(before[2][:first], step.te.token_table[:first])          # no decay on a skipped step
    # the norm clamp (:1138-1149) runs on every iteration in the reference too, skipped step or not: it may re-round rows
    # that already sit at mean_norm by an ulp, nothing more
    torch.testing.assert_close(before[2][first:], step.te.token_table[first:], rtol=1e-6, atol=0)
    # and it recovers on the next clean batch
    step.x0.copy_(torch.randn(B, 4, hw, hw, generator=g))
    step.step_eager()
    sc = step.scalars()
    assert sc["found_inf"] == 0.0 and sc["opt_steps"] == 2.0 and sc["loss_scale"] == 32768.0


def test_optimizer_tail_in_two_launches_is_bit_equal_to_the_ten_separate_ones():
    """round 6: `tb_optimizer_tail` (sums of squares of both groups + a snapshot of the scaler state | every workgroup derives the step's scalars
    itself, then AdamW per group, the added rows' norm clamp, the decay of the untouched rows; train_textboost.py:1128-1149) against the entry
    points it replaces -- tb_sumsq x2, tb_lr_from_table, tb_scaler_update, tb_adamw x2, tb_weight_decay, tb_renorm_rows -- over clean steps under
    an lr schedule, an overflow step (skipped update, halved scale, the clamp still runs) and the recovery: state, parameters and both Adam
    moments identical bit for bit."""
    from oracle import train_step as ts
    from tests.test_gpu_model import build_step
    from textboost_amd.trainer import lr_lambda
    B, hw, D = 2, 16, 64
    runs = []
    for fused in (True, False):
        st_ref, step, added = build_step(B, hw, D)
        step.fused_tail = fused
        lam = lr_lambda("linear", 1, 6)
        step.set_lr_table([lam(k) for k in range(8)])
        g = torch.Generator().manual_seed(12)
        snaps = []
        for it in range(5):
            step.input_ids.copy_(ts.synthetic_ids(B, added, g)); step.prior_ids.copy_(ts.synthetic_ids(B, added, g, prior=True))
            step.noise.copy_(torch.randn(B, 4, hw, hw, generator=g)); step.timesteps.copy_(torch.randint(0, 1000, (B,), generator=g))
            step.x0.copy_(torch.randn(B, 4, hw, hw, generator=g))
            if it == 2:
                step.x0.fill_(1e6)            # overflows fp16 inside the UNet -> non-finite gradients -> the step is skipped
            step.step_eager()
            torch.cuda.synchronize()
            snaps.append([t.clone() for t in (step.state, step.flat_lora, step.te.token_table, step.m_lora, step.v_lora, step.m_emb, step.v_emb,
                                              step.added_norms)])
        sc = step.scalars()
        assert sc["opt_steps"] == 4.0 and sc["loss_scale"] == 32768.0
        runs.append(snaps)
    names = ("state", "lora", "token_table", "m_lora", "v_lora", "m_emb", "v_emb", "added_norms")
    for it, (a, b) in enumerate(zip(*runs)):
        for n, x, y in zip(names, a, b):
            same = torch.equal(x, y) or ((torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)))
            assert same, (it, n, (x - y).abs().max().item(), x.flatten()[:16].tolist(), y.flatten()[:16].tolist())


def test_cli_rejects_unbuilt_arithmetic_options(tmp_path):
    """Options that would change the step's arithmetic but are not built must fail loudly (the fp32 no-AMP mode is built: tests/test_gpu_f32.py)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    base = ["--pretrained_model_name_or_path", "/nonexistent/sd15", "--output_dir", str(tmp_path / "o"), "--train_batch_size", "1",
            "--resolution", "128", "--max_train_steps", "1"]
    for extra in (["--mixed_precision", "fp16", "--text_encoder_use_attention_mask"],
                  ["--mixed_precision", "fp16", "--unet_params_to_train", "crossattn_kv"]):   # (bf16 + crossattn_kv is built: tests/test_gpu_bf16.py)
        with pytest.raises(NotImplementedError):
            T.main(T.parse_args(base + extra))


def test_cli_reads_the_architecture_from_the_model_directory(tmp_path, monkeypatch):
    """from_pretrained semantics: unet/config.json + text_encoder/config.json select the geometry (here an SD2.x-style one: Linear proj_in/out,
    per-level head counts, erf-GELU text MLP), scheduler_config.json the prediction type, assets/null_emb_sd21base.pt the null embedding."""
    import json
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    m = tmp_path / "sd2-tiny"
    for sub in ("unet", "text_encoder", "scheduler"):
        (m / sub).mkdir(parents=True)
    json.dump({"attention_head_dim": [2, 4, 8, 8], "block_out_channels": [64, 128, 256, 256], "cross_attention_dim": 128,
               "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
               "in_channels": 4, "out_channels": 4, "layers_per_block": 2, "norm_num_groups": 32, "norm_eps": 1e-5,
               "use_linear_projection": True, "act_fn": "silu"}, open(m / "unet" / "config.json", "w"))
    json.dump({"hidden_act": "gelu", "hidden_size": 128, "intermediate_size": 256, "num_attention_heads": 4, "num_hidden_layers": 2,
               "max_position_embeddings": 77, "vocab_size": 49408, "layer_norm_eps": 1e-5}, open(m / "text_encoder" / "config.json", "w"))
    json.dump({"prediction_type": "v_prediction"}, open(m / "scheduler" / "scheduler_config.json", "w"))
    (tmp_path / "assets").mkdir()
    null = torch.randn(77, 128, generator=torch.Generator().manual_seed(0))
    torch.save(null, str(tmp_path / "assets" / "null_emb_sd21base.pt"))
    monkeypatch.chdir(tmp_path)
    out = str(tmp_path / "run")
    args = T.parse_args(["--pretrained_model_name_or_path", str(m), "--output_dir", out, "--train_batch_size", "2", "--resolution", "128",
                         "--max_train_steps", "3", "--placeholder_token", "<dog>", "--lora_rank", "8", "--mixed_precision", "fp16", "--seed", "5"])
    T.main(args)
    log = open(os.path.join(out, "training.log")).read()
    assert "null embedding loaded from assets/null_emb_sd21base.pt" in log
    from safetensors.torch import load_file
    sd = load_file(os.path.join(out, "text_encoder", "adapter_model.safetensors"))
    assert len(sd) == 2 * 3 * 2 and sd["base_model.model.text_model.encoder.layers.1.self_attn.q_proj.lora_A.weight"].shape == (8, 128)
    d = torch.load(os.path.join(out, "dog.bin"))
    assert d["<dog>"].shape == (128,) and torch.isfinite(d["<dog>"]).all()
