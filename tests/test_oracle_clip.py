"""Pins oracle/clip_text.py against the reference's own TextBoostModel outputs (tests/golden/clip_textboost_tiny.pt,
made by tests/golden/make_golden.py from /root/reference/textboost/text_encoder.py) -- CPU only."""
import os

import torch

from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens


def load_golden_encoder(golden_dir, r=0):
    g = torch.load(os.path.join(golden_dir, "clip_textboost_tiny.pt"))
    c = g["cfg"]
    cfg = CLIPTextCfg(vocab_size=c["V"], hidden_size=c["D"], intermediate_size=c["I"], num_layers=c["L"], num_heads=c["H"])
    torch.manual_seed(0)
    enc = TextBoostEncoder(cfg, r=r)
    sd = dict(g["state_dict"])
    emb = torch.zeros(c["V"], c["D"])
    emb[g["emb_rows_idx"]] = g["emb_rows"]
    sd["embeddings.token_embedding.weight"] = emb
    enc.load_hf_state_dict({"text_model." + k: v for k, v in sd.items()})
    return enc, g


def test_forward_matches_reference_textboost_model(golden_dir):
    enc, g = load_golden_encoder(golden_dir)
    with torch.no_grad():
        raw = enc.transformer(g["ids"])
        torch.testing.assert_close(raw, g["raw"], rtol=1e-5, atol=1e-5)
        enc.set_null_embedding(g["null"])
        out = enc(g["ids"])
    torch.testing.assert_close(out, g["out"], rtol=1e-5, atol=1e-5)
    assert torch.equal(out[:, 0], g["null"][0].expand(out.shape[0], -1))  # text_encoder.py:81-86
    assert torch.equal(out[2], g["null"])                                  # :71-79 (row 2 is the null prompt)


def test_backward_matches_reference_textboost_model(golden_dir):
    enc, g = load_golden_encoder(golden_dir)
    enc.set_null_embedding(g["null"])
    out = enc(g["ids"])
    (out * g["R"]).sum().backward()
    ge = enc.token_embedding.weight.grad
    torch.testing.assert_close(ge[g["emb_rows_idx"]], g["g_emb_rows"], rtol=1e-4, atol=1e-5)
    inv = {v: k for k, v in enc.hf_key_map().items()}
    params = dict(enc.named_parameters())
    assert len(g["grads"]) >= 6
    for hf_name, gref in g["grads"].items():
        ours = inv["text_model." + hf_name]
        torch.testing.assert_close(params[ours].grad, gref, rtol=1e-4, atol=1e-5)


def test_lora_zero_B_is_identity_and_param_count(golden_dir):
    enc0, g = load_golden_encoder(golden_dir, r=0)
    enc4, _ = load_golden_encoder(golden_dir, r=4)
    with torch.no_grad():
        torch.testing.assert_close(enc0.transformer(g["ids"]), enc4.transformer(g["ids"]))
    D, L = g["cfg"]["D"], g["cfg"]["L"]
    assert sum(p.numel() for p in enc4.lora_parameters()) == L * 3 * 2 * 4 * D
    # SD1.x: 12 layers x (q,k,v) x (A[4,768] + B[768,4]) = 221,184  
    assert 12 * 3 * 2 * 4 * 768 == 221_184
    # LoRA linear equals merged weight W + B A (peft semantics, alpha = r)
    lin = enc4.layers[0].q
    with torch.no_grad():
        lin.lora_B.normal_()
        x = torch.randn(5, D)
        merged = torch.nn.functional.linear(x, lin.weight + lin.lora_B @ lin.lora_A, lin.bias)
        torch.testing.assert_close(lin(x), merged, rtol=1e-4, atol=1e-5)


def test_clip_l_param_count():
    with torch.device("meta"):
        enc = TextBoostEncoder(CLIPTextCfg.sd15())
    n = sum(p.numel() for p in enc.parameters())
    assert n == 123_060_480


def test_add_tokens_rows(golden_dir):
    enc, g = load_golden_encoder(golden_dir)
    V = enc.token_embedding.weight.shape[0]
    ids = add_tokens(enc, [5, 17])
    assert ids == [V, V + 1]
    assert torch.equal(enc.token_embedding.weight[V], enc.token_embedding.weight[5])
    assert torch.equal(enc.token_embedding.weight[V + 1], enc.token_embedding.weight[17])


def test_wrapper_order_golden(golden_dir):
    g = torch.load(os.path.join(golden_dir, "wrapper_order.pt"))
    assert g["n5_seed42_rep2"] == [4, 2, 3, 1, 0, 0, 2, 3, 4, 1]
    assert g["n1_seed42_rep4"] == [0, 0, 0, 0]
