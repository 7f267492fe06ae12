"""Command-line interface of the TextBoost trainer -- the boundary of /root/reference/train_textboost.py:49-450.

Every flag name / type / default / action / nargs / choices of the reference's `parse_args` is reproduced (66 flags, incl.
the `default=True` store_true of --disable_weighted_sample, :406-411, and argparse prefix abbreviation, which README.md:64
relies on); tests/test_host_logic.py checks this table against tests/golden/cli_flags.json, extracted from the reference
by AST.  Post-parse validation follows :435-448."""
from __future__ import annotations

import argparse
import warnings

_T = {"str": str, "int": int, "float": float}

# (flag, kwargs) -- interface facts of the reference CLI
FLAGS = [
    ("--pretrained_model_name_or_path", dict(type="str", default=None, required=True)),
    ("--revision", dict(type="str", default=None, required=False)),
    ("--variant", dict(type="str", default=None)),
    ("--tokenizer_name", dict(type="str", default=None)),
    ("--instance_data_dir", dict(type="str", default=None)),
    ("--instance", dict(type="str")),
    ("--class_data_dir", dict(type="str", default=None, required=False)),
    ("--instance_token", dict(type="str", default=None)),
    ("--class_token", dict(type="str", nargs="+", default=None)),
    ("--with_image_prior", dict(default=False, action="store_true")),
    ("--image_ppl_weight", dict(type="float", default=1.0)),
    ("--kpl_weight", dict(type="float", default=0.1)),
    ("--kpl_type", dict(type="str", default="cos")),
    ("--num_prior_images", dict(type="int", default=200)),
    ("--output_dir", dict(type="str", default="dreambooth-model")),
    ("--seed", dict(type="int", default=42)),
    ("--resolution", dict(type="int", default=512)),
    ("--center_crop", dict(default=False, action="store_true")),
    ("--train_batch_size", dict(type="int", default=1)),
    ("--sample_batch_size", dict(type="int", default=4)),
    ("--max_train_steps", dict(type="int", default=500)),
    ("--checkpointing_steps", dict(type="int", default=100)),
    ("--checkpoints_total_limit", dict(type="int", default=None)),
    ("--resume_from_checkpoint", dict(type="str", default=None)),
    ("--gradient_accumulation_steps", dict(type="int", default=1)),
    ("--gradient_checkpointing", dict(action="store_true")),
    ("--learning_rate", dict(type="float", default=5e-5)),
    ("--emb_learning_rate", dict(type="float", default=1e-3)),
    ("--scale_lr", dict(action="store_true", default=False)),
    ("--lr_scheduler", dict(type="str", default="constant")),
    ("--lr_warmup_steps", dict(type="int", default=500)),
    ("--dataloader_num_workers", dict(type="int", default=2)),
    ("--adam_beta1", dict(type="float", default=0.9)),
    ("--adam_beta2", dict(type="float", default=0.999)),
    ("--adam_weight_decay", dict(type="float", default=1e-2)),
    ("--adam_epsilon", dict(type="float", default=1e-08)),
    ("--max_grad_norm", dict(default=1.0, type="float")),
    ("--hub_token", dict(type="str", default=None)),
    ("--logging_dir", dict(type="str", default="logs")),
    ("--allow_tf32", dict(action="store_true")),
    ("--report_to", dict(type="str", default="tensorboard")),
    ("--validation_prompts", dict(type="str", nargs="+", default=None)),
    ("--num_validation_images", dict(type="int", default=4)),
    ("--validation_steps", dict(type="int", default=100)),
    ("--mixed_precision", dict(type="str", default=None, choices=["no", "fp16", "bf16"])),
    ("--prior_generation_precision", dict(type="str", default=None, choices=["no", "fp32", "fp16", "bf16"])),
    ("--concepts_list", dict(type="str", default=None)),
    ("--no_safe_serialization", dict(action="store_true")),
    ("--skip_save_text_encoder", dict(action="store_true", required=False)),
    ("--class_labels_conditioning", dict(required=False, default=None)),
    ("--validation_scheduler", dict(type="str", default="DPMSolverMultistepScheduler",
                                    choices=["DPMSolverMultistepScheduler", "DDPMScheduler"])),
    ("--text_encoder_use_attention_mask", dict(action="store_true", required=False)),
    ("--placeholder_token", dict(type="str", default="<dog>")),
    ("--initializer_token", dict(type="str", default="dog")),
    ("--unet_params_to_train", dict(type="str", choices=["none", "crossattn_kv", "crossattn", "attn", "all"], default="none")),
    ("--augment", dict(default="none")),
    ("--augment_ops", dict(type="str", default="object")),
    ("--augment_p", dict(type="float", default=0.8)),
    ("--augment_prompt", dict(type="int", default=1)),
    ("--augment_inversion", dict(action="store_true", default=False)),
    ("--num_samples", dict(type="int", default=None)),
    ("--lora_rank", dict(type="int", default=4)),
    ("--disable_weighted_sample", dict(action="store_true", default=True)),
    ("--null_prob", dict(type="float", default=0.1)),
    ("--template", dict(type="str", default="textboost")),
    ("--mixing", dict(action="store_true", default=False)),
]


def flag_table():
    """Normalised view used by the parity test (same schema as tests/golden/cli_flags.json)."""
    out = []
    for name, kw in FLAGS:
        d = {"flags": [name]}
        d.update(kw)
        out.append(d)
    return sorted(out, key=lambda s: s["flags"][0])


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="TextBoost training on MI355X (HIP kernels behind the reference CLI).")
    for name, kw in FLAGS:
        kw = dict(kw)
        if "type" in kw:
            kw["type"] = _T[kw["type"]]
        p.add_argument(name, **kw)
    return p


def parse_args(input_args=None):
    args = build_parser().parse_args(input_args)
    # train_textboost.py:435-448
    if args.with_image_prior:
        if args.class_data_dir is None:
            raise ValueError("You must specify a data directory for class images.")
        if args.class_token is None:
            raise ValueError("You must specify prompt for class images.")
    else:
        if args.class_data_dir is not None:
            warnings.warn("You need not use --class_data_dir without --with_image_prior.")
        if args.class_token is not None:
            warnings.warn("You need not use --class_token without --with_image_prior.")
    if args.augment_inversion and not bool(args.augment_prompt):
        raise ValueError("You need to use --augment_prompt=1 with --augment_prompt.")
    return args
