"""Generates tests/golden/*.pt from the REFERENCE itself, in the build container only.

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)

What is captured (data only -- inputs and expected outputs, no reference source):
* clip_textboost_tiny.pt : the reference's own `TextBoostModel` (textboost/text_encoder.py:17-87), imported
  from /root/reference with the 2-line shim described in SURVEY.md 8(c) (transformers 5.x removed
  `CLIPTextTransformer` and `.text_model`), random tiny weights, ids incl. a null prompt ->
  last_hidden_state with pins, and gradients of a fixed scalar loss w.r.t. touched embedding rows and
  a few encoder weights.
* wrapper_order.pt : `Wrapper(...).shuffle(seed).repeat()` index order (textboost/dataset.py:838-872).
"""
import functools
import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def ref_textboost_model():
    sys.path.insert(0, REF)
    from transformers.models.clip import modeling_clip as mc
    mc.CLIPTextTransformer = object  # unused import in textboost/text_encoder.py:10-14
    from textboost.text_encoder import TextBoostModel
    from transformers import CLIPTextConfig, CLIPTextModel
    return TextBoostModel, CLIPTextConfig, CLIPTextModel


def make_clip():
    TextBoostModel, CLIPTextConfig, CLIPTextModel = ref_textboost_model()
    D, L, H, I = 64, 2, 2, 128
    V = 49408 + 3
    cfg = CLIPTextConfig(vocab_size=V, hidden_size=D, intermediate_size=I, num_hidden_layers=L, num_attention_heads=H,
                         max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=49406, eos_token_id=49407,
                         pad_token_id=49407)
    torch.manual_seed(1234)
    m = TextBoostModel(cfg)
    m.text_model = functools.partial(CLIPTextModel.forward, m)  # shim: 5.x has no .text_model attribute
    m.float()
    with torch.no_grad():  # make LN affine / biases non-trivial
        for n, p in m.named_parameters():
            if "layer_norm" in n or n.endswith("bias"):
                p.add_(torch.randn_like(p) * 0.1)
    pool = torch.tensor([5, 17, 320, 1000, 2001, 30000, 49405, 49408, 49409, 49410])
    B = 4
    g = torch.Generator().manual_seed(7)
    ids = torch.full((B, 77), 49407)
    ids[:, 0] = 49406
    for b in range(B):
        n = 3 + 2 * b
        ids[b, 1:1 + n] = pool[torch.randint(0, len(pool), (n,), generator=g)]
    ids[2, 1:] = 49407  # null prompt
    null = torch.randn(77, D, generator=g)
    m.set_null_embedding(null)
    R = torch.randn(B, 77, D, generator=g)
    out = m(ids, attention_mask=None, return_dict=False)[0]
    loss = (out * R).sum()
    loss.backward()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k != "null_embedding"}
    emb = sd.pop("embeddings.token_embedding.weight")
    used = torch.unique(ids)
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()
             if p.grad is not None and ("q_proj" in n or "fc2" in n or "position_embedding" in n or "final_layer_norm" in n)}
    g_emb = m.embeddings.token_embedding.weight.grad
    assert g_emb[[i for i in range(V) if i not in set(used.tolist())][:100]].abs().max() == 0
    # also the unpinned transformer output (before TextBoost pins) from the stock CLIPTextModel.forward
    with torch.no_grad():
        raw = CLIPTextModel.forward(m, input_ids=ids, return_dict=False)[0]
    torch.save({"cfg": dict(D=D, L=L, H=H, I=I, V=V), "state_dict": sd, "emb_rows_idx": used, "emb_rows": emb[used].clone(),
                "ids": ids, "null": null, "R": R, "out": out.detach().clone(), "raw": raw.clone(),
                "g_emb_rows": g_emb[used].clone(), "grads": grads}, os.path.join(OUT, "clip_textboost_tiny.pt"))
    print("clip golden:", out.shape, "pins ok:", torch.equal(out[:, 0], null[0].expand(B, -1)), torch.equal(out[2], null))


def make_wrapper():
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms"); v2 = types.ModuleType("torchvision.transforms.v2")
    tv.transforms = tvt; tvt.v2 = v2
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.v2": v2})
    try:
        from textboost.dataset import Wrapper
    except Exception as e:  # pragma: no cover
        print("wrapper import failed:", e)
        return
    it = iter(Wrapper(range(5), drop_last=False).shuffle(seed=42).repeat(2))
    order = [int(x) for x in it]
    it1 = iter(Wrapper(range(1), drop_last=False).shuffle(seed=42).repeat(4))
    order1 = [int(x) for x in it1]
    torch.save({"n5_seed42_rep2": order, "n1_seed42_rep4": order1}, os.path.join(OUT, "wrapper_order.pt"))
    print("wrapper golden:", order, order1)


if __name__ == "__main__":
    make_clip()
    make_wrapper()


def make_cli_flags():
    """Interface facts of the reference CLI (train_textboost.py:49-450): flag names, types, defaults, actions, nargs, choices,
    required -- extracted from the argparse calls by AST (the file itself cannot be imported: diffusers/peft are absent)."""
    import ast
    import json
    src = open(os.path.join(REF, "train_textboost.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args"][0]
    flags = []
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            names = [a.value for a in node.args if isinstance(a, ast.Constant)]
            spec = {"flags": names}
            for kw in node.keywords:
                if kw.arg == "help":
                    continue
                if kw.arg == "type":
                    spec["type"] = kw.value.id
                else:
                    spec[kw.arg] = ast.literal_eval(kw.value)
            flags.append(spec)
    flags.sort(key=lambda s: s["flags"][0])
    with open(os.path.join(OUT, "cli_flags.json"), "w") as f:
        json.dump(flags, f, indent=1, sort_keys=True)
    print("cli golden:", len(flags), "flags")


if __name__ == "__main__":
    make_cli_flags()
