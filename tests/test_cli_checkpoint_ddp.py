"""CPU-only: CLI parity with the reference's flag table, output-layout writer, world_size-2 gloo gradient averaging."""
import json
import os
import types

import pytest
import torch


def test_cli_flag_table_matches_reference(golden_dir):
    from textboost_amd import cli
    golden = json.load(open(os.path.join(golden_dir, "cli_flags.json")))
    assert cli.flag_table() == golden
    assert len(golden) == 66


def test_cli_defaults_abbreviations_and_validation():
    from textboost_amd.cli import parse_args
    a = parse_args(["--pretrained_model_name_or_path", "x"])
    assert a.disable_weighted_sample is True and a.lora_rank == 4 and a.kpl_weight == 0.1 and a.mixed_precision is None
    assert a.emb_learning_rate == 1e-3 and a.learning_rate == 5e-5 and a.adam_weight_decay == 1e-2 and a.null_prob == 0.1
    # README.md:64 of the reference uses the argparse abbreviation --validation_prompt
    a = parse_args(["--pretrained_model_name_or_path", "x", "--validation_prompt", "a", "b", "--class_token", "dog", "cat"])
    assert a.validation_prompts == ["a", "b"] and a.class_token == ["dog", "cat"]
    with pytest.raises(ValueError):
        parse_args(["--pretrained_model_name_or_path", "x", "--with_image_prior"])
    with pytest.raises(ValueError):
        parse_args(["--pretrained_model_name_or_path", "x", "--augment_inversion", "--augment_prompt", "0"])
    with pytest.raises(SystemExit):
        parse_args([])  # --pretrained_model_name_or_path is required


def fake_encoder(L=2, D=16, r=4, V=49408, k=3):
    te = types.SimpleNamespace()
    te.geo = types.SimpleNamespace(hidden_size=D, num_layers=L)
    te.r = r
    te.lora_A = torch.randn(L, 3 * r, D)
    te.lora_B = torch.randn(L, 3 * D, r)
    te.token_table = torch.randn(V + k, D)
    te.first_added = V
    te.n_added = k
    return te


def test_output_layout_matches_reference_readers(tmp_path):
    from safetensors.torch import load_file
    from textboost_amd import checkpoint as ck
    from train_textboost import multi_vector_names
    te = fake_encoder()
    out = str(tmp_path)
    ck.save_text_encoder_adapter(te, os.path.join(out, "text_encoder"), "runwayml/stable-diffusion-v1-5")
    cfg = json.load(open(os.path.join(out, "text_encoder", "adapter_config.json")))
    assert cfg["peft_type"] == "LORA" and cfg["r"] == 4 and cfg["lora_alpha"] == 4 and cfg["lora_dropout"] == 0.0
    assert cfg["target_modules"] == ["q_proj", "k_proj", "v_proj"] and cfg["init_lora_weights"] == "gaussian" and cfg["bias"] == "none"
    sd = load_file(os.path.join(out, "text_encoder", "adapter_model.safetensors"))
    assert len(sd) == 2 * 3 * 2
    a = sd["base_model.model.text_model.encoder.layers.1.self_attn.k_proj.lora_A.weight"]
    b = sd["base_model.model.text_model.encoder.layers.1.self_attn.k_proj.lora_B.weight"]
    assert a.shape == (4, 16) and b.shape == (16, 4) and a.dtype == torch.float32
    assert torch.equal(a, te.lora_A[1, 4:8]) and torch.equal(b, te.lora_B[1, 16:32])
    # token files: placeholder -> 1-D [D]; augmentation -> [1, D]; '<' '>' stripped from the file name only
    added = {"<dog>": 49408}
    aug = {n: 49409 + i for i, n in enumerate(multi_vector_names("<zoom-in>", 2))}
    assert list(aug) == ["<zoom-in_0>", "<zoom-in_1>"]
    ck.save_token_embeddings(te, out, added, aug)
    d = torch.load(os.path.join(out, "dog.bin"))
    assert list(d) == ["<dog>"] and d["<dog>"].shape == (16,) and torch.equal(d["<dog>"], te.token_table[49408])
    z = torch.load(os.path.join(out, "zoom-in_1.bin"))
    assert z["<zoom-in_1>"].shape == (1, 16)
    # eval_dreambooth.py:329-336 counts files starting with the instance name to infer num_vectors
    assert sorted(f for f in os.listdir(out) if f.endswith(".bin")) == ["dog.bin", "zoom-in_0.bin", "zoom-in_1.bin"]
    # lora round trip
    te2 = fake_encoder()
    ck.load_lora_state_dict(te2, sd)
    assert torch.equal(te2.lora_A, te.lora_A) and torch.equal(te2.lora_B, te.lora_B)
    # checkpoint rotation (:1159-1175)
    for s in (50, 100, 150):
        os.makedirs(os.path.join(out, f"checkpoint-{s}"))
    ck.rotate_checkpoints(out, 2)
    assert sorted(d for d in os.listdir(out) if d.startswith("checkpoint")) == ["checkpoint-150"]


FLAT_L, FLAT_R, FLAT_D, FLAT_K = 12, 4, 768, 18       # SD1.5 / CLIP-L flat gradient layout [grad_A | grad_B | added rows]: 0.94 MB
N_LORA = 2 * FLAT_L * 3 * FLAT_R * FLAT_D


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from textboost_amd.trainer import average_gradients, shard_indices, sum_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    average_gradients(g, world)
    idx = shard_indices(1, 4, 3, rank, world) + shard_indices(5, 2, 1, rank, world)
    # --- the step's exchange on the real flat layout with rank-dependent gradients:
    # reference (DDP, train_textboost.py:919-926 + :1109-1117): dense embedding gradient averaged over ranks, THEN rows < first_added zeroed;
    # here: only the added rows travel, summed, and 1/W is folded into the unscale coefficient (tb_scaler_update grad_div)
    gen = torch.Generator().manual_seed(100 + rank)
    first_added, window = 16, 16 + FLAT_K                     # a window of the table: 16 original rows + the 18 added ones
    lora = torch.randn(N_LORA, generator=gen)
    dense_emb = torch.randn(window, FLAT_D, generator=gen)    # this rank's (scaled) embedding gradient, dense like autograd makes it
    ref_lora, ref_emb = lora.clone(), dense_emb.clone()
    dist.all_reduce(ref_lora); dist.all_reduce(ref_emb)
    ref_lora /= world; ref_emb /= world
    ref_emb[:first_added] = 0                                 # :1114-1117
    flat = torch.cat([lora, dense_emb[first_added:].reshape(-1)])
    sum_gradients(flat, world)
    scale = 65536.0
    coef = 1.0 / (scale * world)                              # what scaler_update_kernel writes to state[TB_ST_COEF_EMB]
    err_l = ((flat[:N_LORA] * coef - ref_lora / scale).abs().max() / (ref_lora / scale).abs().max()).item()
    err_e = ((flat[N_LORA:].view(FLAT_K, FLAT_D) * coef - ref_emb[first_added:] / scale).abs().max()).item() / (ref_emb.abs().max().item() / scale)
    norm_ours = (flat[:N_LORA].double().pow(2).sum().sqrt() * coef).item()      # state[TB_ST_GRAD_NORM]
    norm_ref = (ref_lora.double() / scale).norm().item()
    q.put((rank, g.tolist(), idx, err_l, err_e, abs(norm_ours - norm_ref) / norm_ref, flat.numel() * 4))
    dist.destroy_process_group()


def test_gloo_world2_gradient_average_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    ps = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = (torch.arange(10, dtype=torch.float32) * 1.5).tolist()
    assert res[0][1] == expect and res[1][1] == expect            # mean over ranks, identical on every rank
    assert res[0][2][:4] == [0, 0, 0, 0] and res[1][2][:4] == [0, 0, 0, 0]   # one image: every shard non-empty
    assert res[0][2][4:] == [4, 0] and res[1][2][4:] == [1, 2]   # it=1: rank0 -> samples 4,5%5 ; rank1 -> 6%5, 7%5
    for r in res:  # masking and averaging commute; the folded 1/W equals DDP's mean (fp32 rounding only)
        assert r[3] < 1e-6 and r[4] < 1e-6 and r[5] < 1e-6, r[3:]
        assert r[6] == (N_LORA + FLAT_K * FLAT_D) * 4 == 940032   # the 0.94 MB bucket of DESIGN.md section 5


def test_sharding_for_eight_ranks_never_leaves_a_rank_empty():
    """8 ranks, 1..5 instance images (the DreamBooth one-shot case hangs in the reference): every rank gets data, every image
    is used, and the per-epoch multiset is the tiled key list."""
    from textboost_amd.data import IndexStream
    from textboost_amd.trainer import shard_indices
    for n in (1, 3, 5, 8, 11):
        taken = []
        for rank in range(8):
            st = IndexStream(n, seed=42, rank=rank, world=8)
            got = st.take(4)
            assert len(got) == 4 and all(0 <= i < n for i in got)
            taken.append(got[0])
        assert set(taken) == set(range(min(n, 8))) or n > 8       # first pass: ranks cover the (tiled) key list
        for it in range(3):
            cover = [i for rank in range(8) for i in shard_indices(n, 2, it, rank, 8)]
            assert len(cover) == 16 and set(cover) == set(range(n)) or n > 16
