"""Output / checkpoint layout of the reference (train_textboost.py:1157-1209, :1236-1266), so that inference.py:55-68 and
eval_dreambooth.py:193-206 / :329-337 read the results unchanged:

  <out>/text_encoder/adapter_config.json + adapter_model.safetensors     PEFT LoRA adapter, fp32, keys
        base_model.model.text_model.encoder.layers.{i}.self_attn.{q,k,v}_proj.lora_{A,B}.weight
  <out>/{tok}.bin                                                        torch.save({"<tok>": tensor}); placeholder tokens save a
        1-D [D] row (:1189-1193), augmentation tokens a 2-D [1, D] slice (:1201-1205); '<' '>' stripped from the FILE name only
  <out>/checkpoint-{step}/ (same two things) + trainer state: model.safetensors, optimizer.bin, scheduler.bin, scaler.pt,
        random_states_0.pkl (the file names accelerate.save_state uses; the readers skip optimizer.bin / scheduler.bin)
"""
from __future__ import annotations

import json
import os
import pickle
import random
import shutil
from typing import Dict

import numpy as np
import torch
from safetensors.torch import load_file, save_file


def lora_state_dict(te) -> Dict[str, torch.Tensor]:
    """flat [L,3r,D] / [L,3D,r] masters -> PEFT key layout (adapter name stripped on save, like peft does)."""
    sd = {}
    D, r = te.geo.hidden_size, te.r
    if not r:  # --lora_rank 0: no adapter exists
        return sd
    for i in range(te.geo.num_layers):
        for p, name in enumerate(("q_proj", "k_proj", "v_proj")):
            base = f"base_model.model.text_model.encoder.layers.{i}.self_attn.{name}"
            sd[base + ".lora_A.weight"] = te.lora_A[i, p * r:(p + 1) * r].detach().float().cpu().contiguous()
            sd[base + ".lora_B.weight"] = te.lora_B[i, p * D:(p + 1) * D].detach().float().cpu().contiguous()
    return sd


def load_lora_state_dict(te, sd):
    D, r = te.geo.hidden_size, te.r
    if not r:
        return
    for i in range(te.geo.num_layers):
        for p, name in enumerate(("q_proj", "k_proj", "v_proj")):
            base = f"base_model.model.text_model.encoder.layers.{i}.self_attn.{name}"
            te.lora_A[i, p * r:(p + 1) * r].copy_(sd[base + ".lora_A.weight"])
            te.lora_B[i, p * D:(p + 1) * D].copy_(sd[base + ".lora_B.weight"])


def unet_lora_state_dict(unet) -> Dict[str, torch.Tensor]:
    """UNet cross-attention K/V adapters (--unet_params_to_train crossattn_kv, :712-721) under the parameter names diffusers' `unet.add_adapter`
    gives them (peft-injected modules, adapter "default"): `<attn2 path>.to_{k,v}.lora_{A,B}.default.weight`."""
    sd = {}
    r = unet.kv_r
    for l, (p, C) in enumerate(unet.xattn):
        ko = unet.kv_off[p]
        for j, name in enumerate(("to_k", "to_v")):
            sd[f"{p}.{name}.lora_A.default.weight"] = unet.kv_lora_A[l, j * r:(j + 1) * r].detach().float().cpu().contiguous()
            sd[f"{p}.{name}.lora_B.default.weight"] = unet.kv_lora_B[ko + j * C:ko + (j + 1) * C].detach().float().cpu().contiguous()
    return sd


def load_unet_lora_state_dict(unet, sd):
    r = unet.kv_r
    for l, (p, C) in enumerate(unet.xattn):
        ko = unet.kv_off[p]
        for j, name in enumerate(("to_k", "to_v")):
            unet.kv_lora_A[l, j * r:(j + 1) * r].copy_(sd[f"{p}.{name}.lora_A.default.weight"])
            unet.kv_lora_B[ko + j * C:ko + (j + 1) * C].copy_(sd[f"{p}.{name}.lora_B.default.weight"])


def unet_peft_adapter_state_dict(unet) -> Dict[str, torch.Tensor]:
    """the same tensors under the keys peft 0.13.2 writes into `adapter_model.safetensors` (`get_peft_model_state_dict`: adapter name stripped,
    `base_model.model.` prefix), i.e. what `set_peft_model_state_dict` / `PeftModel.from_pretrained` accept -- as `lora_state_dict` does for the encoder"""
    return {"base_model.model." + k.replace(".default.weight", ".weight"): v for k, v in unet_lora_state_dict(unet).items()}


def load_unet_peft_adapter_state_dict(unet, sd):
    """accepts peft's adapter layout (`base_model.model.<param>.lora_{A,B}.weight`, what `save_unet_adapters` writes since round 5) and the
    in-model layout of earlier rounds' files (`<param>.lora_{A,B}.default.weight`, no prefix); anything else is refused by name."""
    pre = "base_model.model."
    keys = list(sd)
    if keys and all(k.startswith(pre) for k in keys):
        bad = [k for k in keys if ".default." in k or not k.endswith((".lora_A.weight", ".lora_B.weight"))]
        if bad:
            raise KeyError(f"not a peft adapter key (expected '{pre}<param>.lora_A|lora_B.weight'): {bad[0]}")
        sd = {k[len(pre):-len(".weight")] + ".default.weight": v for k, v in sd.items()}
    elif keys and all(k.endswith((".lora_A.default.weight", ".lora_B.default.weight")) for k in keys):
        pass   # the in-model layout
    else:
        bad = next(k for k in keys if not k.startswith(pre)) if keys else "<empty state dict>"
        raise KeyError(f"unrecognised UNet adapter layout (neither peft's '{pre}...lora_A.weight' nor '...lora_A.default.weight'): {bad}")
    load_unet_lora_state_dict(unet, sd)


def save_unet_adapters(unet, out_dir: str, base_model_name_or_path: str):
    """<out>/unet/ (:1237-1239).  The reference's `unet.save_pretrained` writes the WHOLE fp32 UNet (3.4 GB: the frozen base weights under
    `.base_layer.` names plus the adapters); no reader of the output layout loads it (inference.py / eval_dreambooth.py never open unet/).
    Written here in peft's adapter layout, so that the file name does not promise a loadable diffusers model: `adapter_model.safetensors` (the
    adapter tensors under peft's adapter keys, `unet_peft_adapter_state_dict`: `base_model.model.<module path>.lora_{A,B}.weight`) +
    `adapter_config.json` naming the base model whose weights are unchanged + a README.txt saying so."""
    os.makedirs(out_dir, exist_ok=True)
    save_file(unet_peft_adapter_state_dict(unet), os.path.join(out_dir, "adapter_model.safetensors"), metadata={"format": "pt"})
    cfg = adapter_config(unet.kv_r, base_model_name_or_path)          # a pure peft LoraConfig: LoraConfig(**json) must accept every key
    cfg["target_modules"] = ["attn2.to_k", "attn2.to_v"]
    with open(os.path.join(out_dir, "adapter_config.json"), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    with open(os.path.join(out_dir, "README.txt"), "w") as f:
        f.write("adapter_model.safetensors holds the cross-attention K/V LoRA tensors only (peft adapter layout). The UNet's base weights are frozen "
                "and unchanged: load them from <base_model_name_or_path>/unet.\n")


def adapter_config(rank: int, base_model_name_or_path: str) -> dict:
    """peft 0.13.2 LoraConfig JSON for LoraConfig(r, lora_alpha=r, init_lora_weights="gaussian", target q/k/v) (:702-709)."""
    return {"alpha_pattern": {}, "auto_mapping": None, "base_model_name_or_path": base_model_name_or_path, "bias": "none",
            "fan_in_fan_out": False, "inference_mode": True, "init_lora_weights": "gaussian", "layer_replication": None,
            "layers_pattern": None, "layers_to_transform": None, "loftq_config": {}, "lora_alpha": rank, "lora_dropout": 0.0,
            "megatron_config": None, "megatron_core": "megatron.core", "modules_to_save": None, "peft_type": "LORA", "r": rank,
            "rank_pattern": {}, "revision": None, "target_modules": ["q_proj", "k_proj", "v_proj"], "task_type": None,
            "use_dora": False, "use_rslora": False}


def save_text_encoder_adapter(te, out_dir: str, base_model_name_or_path: str):
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "adapter_config.json"), "w") as f:
        json.dump(adapter_config(te.r, base_model_name_or_path), f, indent=2, sort_keys=True)
    save_file(lora_state_dict(te), os.path.join(out_dir, "adapter_model.safetensors"), metadata={"format": "pt"})


def save_token_embeddings(te, out_dir: str, added_tokens: Dict[str, int], aug_token_dict: Dict[str, int] | None):
    os.makedirs(out_dir, exist_ok=True)
    W = te.token_table.detach().float().cpu()
    for token, tid in added_tokens.items():
        torch.save({token: W[tid].clone()}, os.path.join(out_dir, token.replace("<", "").replace(">", "") + ".bin"))   # 1-D [D]
    for token, tid in (aug_token_dict or {}).items():
        torch.save({token: W[tid:tid + 1].clone()}, os.path.join(out_dir, token.replace("<", "").replace(">", "") + ".bin"))  # [1, D]


def rotate_checkpoints(output_dir: str, total_limit):
    """:1159-1175"""
    if total_limit is None:
        return
    cks = sorted([d for d in os.listdir(output_dir) if d.startswith("checkpoint")], key=lambda x: int(x.split("-")[1]))
    if len(cks) >= total_limit:
        for d in cks[: len(cks) - total_limit + 1]:
            shutil.rmtree(os.path.join(output_dir, d))


def save_trainer_state(step_obj, ckpt_dir: str):
    te = step_obj.te
    os.makedirs(ckpt_dir, exist_ok=True)
    model = dict(lora_state_dict(te))
    model["token_embedding.added_rows"] = te.token_table[te.first_added:].detach().float().cpu().contiguous()
    # the WHOLE table, as accelerate's model.safetensors holds it: rows below first_added carry the accumulated decoupled weight decay of
    # every optimizer step so far, which depends on the lr schedule and on skipped steps -- saving them makes resume bit-exact
    model["text_model.embeddings.token_embedding.weight"] = te.token_table.detach().float().cpu().contiguous()
    save_file(model, os.path.join(ckpt_dir, "model.safetensors"))
    extra = {}
    if getattr(step_obj, "n_unet", 0):  # accelerate.prepare(text_encoder, unet, ...) (:924-926) saves the second model as model_1.safetensors
        save_file(unet_lora_state_dict(step_obj.unet), os.path.join(ckpt_dir, "model_1.safetensors"))
        extra = {"m_unet": step_obj.m_unet.cpu(), "v_unet": step_obj.v_unet.cpu()}
    torch.save({"m_lora": step_obj.m_lora.cpu(), "v_lora": step_obj.v_lora.cpu(), "m_emb": step_obj.m_emb.cpu(),
                "v_emb": step_obj.v_emb.cpu(), "state": step_obj.state.cpu(), **extra,
                "orig_rows_decay_steps": float(step_obj.state[2].item()),
                "lr_table": step_obj.lr_table.cpu() if getattr(step_obj, "lr_table", None) is not None else None},
               os.path.join(ckpt_dir, "optimizer.bin"))
    torch.save({"last_epoch": float(step_obj.state[2].item()), "lr_multiplier": 1.0 + float(step_obj.state[13].item())},
               os.path.join(ckpt_dir, "scheduler.bin"))  # lambda(step) of --lr_scheduler is recomputed from the step index on resume
    torch.save({"scale": float(step_obj.state[0].item()), "growth_tracker": float(step_obj.state[1].item()), "growth_factor": 2.0,
                "backoff_factor": 0.5, "growth_interval": step_obj.hp.growth_interval}, os.path.join(ckpt_dir, "scaler.pt"))
    with open(os.path.join(ckpt_dir, "random_states_0.pkl"), "wb") as f:
        # accelerate's key names; `random` / numpy feed the augmentation draws of the device feeder (textboost_amd/augment.py)
        pickle.dump({"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
                     "torch_manual_seed": torch.get_rng_state(),
                     "torch_cuda_manual_seed": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else []}, f)


def load_trainer_state(step_obj, ckpt_dir: str):
    te = step_obj.te
    model = load_file(os.path.join(ckpt_dir, "model.safetensors"))
    load_lora_state_dict(te, model)
    opt = torch.load(os.path.join(ckpt_dir, "optimizer.bin"))
    for k in ("m_lora", "v_lora", "m_emb", "v_emb", "state"):
        getattr(step_obj, k).copy_(opt[k])
    if getattr(step_obj, "n_unet", 0):
        load_unet_lora_state_dict(step_obj.unet, load_file(os.path.join(ckpt_dir, "model_1.safetensors")))
        step_obj.m_unet.copy_(opt["m_unet"])
        step_obj.v_unet.copy_(opt["v_unet"])
    full = model.get("text_model.embeddings.token_embedding.weight")
    if full is not None:
        te.token_table.copy_(full)
    else:  # checkpoints written before the full table was saved: constant-lr approximation of the decay of the never-updated rows
        te.token_table[te.first_added:].copy_(model["token_embedding.added_rows"])
        n = opt["orig_rows_decay_steps"]
        te.token_table[: te.first_added].mul_((1.0 - step_obj.hp.emb_lr * step_obj.hp.wd) ** n)
    # the lr schedule is NOT restored: LambdaLR's lambdas always come from the current command line (:911-916); the caller sets the table
    # (or, for a constant schedule, resets the multiplier slot the checkpointed `state` carried)
    with open(os.path.join(ckpt_dir, "random_states_0.pkl"), "rb") as f:
        rs = pickle.load(f)
    if "random_state" in rs:
        random.setstate(rs["random_state"])
    if "numpy_random_seed" in rs:
        np.random.set_state(rs["numpy_random_seed"])
    torch.set_rng_state(rs["torch_manual_seed"])
    if rs["torch_cuda_manual_seed"]:
        torch.cuda.set_rng_state_all(rs["torch_cuda_manual_seed"])
