#!/usr/bin/env python
"""TextBoost training entry point on MI355X -- drop-in for the reference's `train_textboost.py` CLI.

    python train_textboost.py --pretrained_model_name_or_path <dir> --placeholder_token "<dog>" ... (same flags as the reference)
    python -m torch.distributed.run --nproc-per-node=N train_textboost.py ...                       (README.md:82 of the reference)

What runs here is the reference's hot path (train_textboost.py:1024-1150) on the HIP kernels of libtextboost_hip.so, behind
the reference's flags (:49-450) and output layout (:1157-1209, :1236-1266).  Data sources, in order of preference (for
`--validation_prompts` without a tokenizer: `<instance_data_dir>/validation_input_ids.pt` [P,77] -- the prompts tokenised by the reference's
tokenizer):

  * image files in `<instance_data_dir>` + `<pretrained_model_name_or_path>/tokenizer/` (the reference's own inputs): every image is decoded
    ONCE, stays resident in HBM, and `TextBoostDataset.__getitem__` (dataset.py:353-381: template draw, `--augment pda|paug`
    PairedAugmentation, Resize(LANCZOS), crop, normalise, tokenise) runs per sample on the device (textboost_amd/augment.py), feeding the
    device VAE encoder; or
  * `<instance_data_dir>/pixel_values.pt` -- a [N,3,R,R] fp32 tensor in [-1,1] (what `TextBoostDataset` yields, dataset.py:420-457):
    the SD VAE encoder then runs on the device at the top of every step, exactly where the reference calls it (:1036-1037;
    weights from `<pretrained_model_name_or_path>/vae/diffusion_pytorch_model.safetensors`, else seeded random init), or
  * `<instance_data_dir>/latents.pt`   -- a [N,4,h,w] fp32 tensor of `vae.encode(x).latent_dist.sample()*scaling_factor`
    (+ optional `input_ids.pt` [N,77] / `prior_input_ids.pt` [M,77] int64 from the reference's tokenizer), or
  * synthetic latents and token ids when no such file exists -- there is no network, tokenizer or checkpoint in
    the build image; weights then come from a seeded random init of the exact SD1.5 / CLIP-L shapes unless
    `<pretrained_model_name_or_path>/{unet/diffusion_pytorch_model,text_encoder/model}.safetensors` exist locally.
"""
import json
import logging
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from textboost_amd import checkpoint as ckpt  # noqa: E402
from textboost_amd import models  # noqa: E402
from textboost_amd.cli import parse_args  # noqa: E402

logger = logging.getLogger("textboost_amd")

# number of embedding vectors per augmentation token = BPE pieces of its initialiser (textboost/utils.py:180-199); the
# tokenizer is not available offline, the counts are the `_i` suffixes the reference's augmentation code emits
AUG_TOKENS_OBJECT = [("<grayscale>", 2), ("<zoom-in>", 2), ("<zoom-out>", 2), ("<collage>", 2), ("<crop>", 1), ("<hflip>", 1),
                     ("<left>", 3), ("<right>", 3)]
AUG_TOKENS_STYLE = [("<hflip>", 1)]


def multi_vector_names(token: str, n: int):
    """textboost/utils.py:133-141: `<x>` -> `<x_0>`, `<x_1>`, ... when the initialiser has several pieces."""
    if n == 1:
        return [token]
    stem = token[:-1] if token.endswith(">") else token
    tail = ">" if token.endswith(">") else ""
    return [f"{stem}_{i}{tail}" for i in range(n)]


def load_tokenizer(model_dir, tokenizer_name=None):
    """`AutoTokenizer.from_pretrained(..., subfolder="tokenizer", use_fast=False)` (:629-641) from LOCAL files only; None when absent.
    --tokenizer_name (:630-633) names the tokenizer directory itself."""
    tdir = tokenizer_name if tokenizer_name else os.path.join(model_dir or "", "tokenizer")
    if not os.path.exists(os.path.join(tdir, "vocab.json")):
        if tokenizer_name:
            raise FileNotFoundError(f"--tokenizer_name {tokenizer_name}: no vocab.json there (only local tokenizer files can be loaded)")
        return None
    from transformers import CLIPTokenizer
    return CLIPTokenizer.from_pretrained(tdir, local_files_only=True)


def load_local_state_dict(path):
    from safetensors.torch import load_file
    return load_file(path) if os.path.exists(path) else None


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train_textboost.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.gradient_accumulation_steps > 1:  # :573-577
            raise ValueError("Gradient accumulation is not supported when training with multiple processes.")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    is_main = rank == 0
    if is_main:
        os.makedirs(args.output_dir, exist_ok=True)
        fh = logging.FileHandler(os.path.join(args.output_dir, "training.log"))  # <output_dir>/training.log, :583-588
        fh.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(name)s - %(message)s"))
        logger.addHandler(fh)
        logger.setLevel(logging.INFO)
    if args.seed is not None:
        torch.manual_seed(args.seed)  # set_seed(args.seed): the same seed on every rank (:601)
    from textboost_amd import _lib
    _lib.lib()
    from textboost_amd.text_encoder import HipTextEncoder
    from textboost_amd.trainer import StepHyper, TextBoostStep, shard_indices
    from textboost_amd.unet import HipUNet
    from textboost_amd.workload import BOS, EOS, synthetic_ids

    # ---- models (:629-656)
    mdir = args.pretrained_model_name_or_path
    usd = load_local_state_dict(os.path.join(mdir, "unet", "diffusion_pytorch_model.safetensors")) if os.path.isdir(mdir) else None
    csd = load_local_state_dict(os.path.join(mdir, "text_encoder", "model.safetensors")) if os.path.isdir(mdir) else None
    unet_geo, clip_geo = models.SD15_UNET, models.SD15_CLIP
    ucfg, ccfg = os.path.join(mdir, "unet", "config.json"), os.path.join(mdir, "text_encoder", "config.json")
    if os.path.exists(ucfg):  # from_pretrained reads the architecture from the model directory (SD1.x, SD2.x): so does this
        unet_geo = models.unet_geometry_from_config(json.load(open(ucfg)))
    if os.path.exists(ccfg):
        clip_geo = models.clip_geometry_from_config(json.load(open(ccfg)))
    if usd is None or csd is None:
        logger.warning("no local weights under %s: seeded random-init SD1.5 / CLIP-L shapes are used", mdir)
        usd = models.random_state_dict(models.unet_shapes(unet_geo), 1234, device=dev)
        csd = models.random_state_dict(models.clip_shapes(clip_geo), 1235, device=dev)
    B = args.train_batch_size
    latent = args.resolution // 8
    lat_path = os.path.join(args.instance_data_dir or "", "latents.pt")
    latents = torch.load(lat_path) if args.instance_data_dir and os.path.exists(lat_path) else None
    px_path = os.path.join(args.instance_data_dir or "", "pixel_values.pt")
    pixels = torch.load(px_path) if args.instance_data_dir and os.path.exists(px_path) else None
    tokenizer = load_tokenizer(mdir, getattr(args, "tokenizer_name", None))
    from textboost_amd import data as tbdata
    # :602-615: one concept from the flags, or the --concepts_list JSON (a list of such dicts: multi-concept training)
    if getattr(args, "concepts_list", None):
        with open(args.concepts_list, "r") as f:
            concepts = json.load(f)
        for c in concepts:
            c.setdefault("initializer_token", args.initializer_token)
    else:
        concepts = [{"instance_data_dir": args.instance_data_dir, "placeholder_token": args.placeholder_token,
                     "initializer_token": args.initializer_token}]
    use_images = tokenizer is not None and all(tbdata.has_instance_images(c.get("instance_data_dir")) for c in concepts)
    if use_images:
        latents, pixels, latent = None, None, args.resolution // 8
    if pixels is not None:
        latents, latent = None, pixels.shape[-1] // 8
    if latents is not None:
        latent = latents.shape[-1]
    # --mixed_precision (:298-308): "fp16" = fp16 UNet / teacher, autocast text encoder with fp32 masters, dynamic loss scaling (:930-939);
    # unset / "no" = the default: weight_dtype float32, everything in fp32, no GradScaler (accelerate launch would otherwise take the mode from its
    # own config file; there is none here, so the flag alone decides)
    fp32_mode = args.mixed_precision in (None, "no")
    bf16_mode = args.mixed_precision == "bf16"   # (:930-934) weight_dtype = bf16; accelerate's GradScaler exists for fp16 only
    from textboost_amd import _lib as _L
    _L.set_half("bf16" if bf16_mode else "fp16")   # which build of the kernel library this process computes with
    adt = torch.float32 if fp32_mode else _L.half_dtype()
    unet = HipUNet(unet_geo, usd, B, latent, latent, text_len=clip_geo.max_pos, device=dev, dtype=adt)
    # ---- validation sampling (:453-531, :1212-1228): prompts must arrive tokenised (no tokenizer offline)
    val_ids_path = os.path.join(args.instance_data_dir or "", "validation_input_ids.pt")
    val_ids = torch.load(val_ids_path) if args.validation_prompts and os.path.exists(val_ids_path) else None
    if args.validation_prompts and tokenizer is not None:
        val_ids = torch.zeros(len(args.validation_prompts), clip_geo.max_pos, dtype=torch.int64)  # filled once the tokens are registered
    sampler = None
    if args.validation_prompts and val_ids is None:
        logger.warning("--validation_prompts given but neither a tokenizer nor %s exists: validation is skipped", val_ids_path)
    if val_ids is not None and is_main:
        from textboost_amd.sampler import HipSampler
        from textboost_amd.vae import HipVAEDecoder, VAEGeometry, vae_decoder_shapes
        nv = val_ids.shape[0] * args.num_validation_images
        dsd = load_local_state_dict(os.path.join(mdir, "vae", "diffusion_pytorch_model.safetensors")) if os.path.isdir(mdir) else None
        if dsd is None:
            logger.warning("no local VAE weights under %s: seeded random-init SD VAE decoder shapes are used for validation images", mdir)
            dsd = models.random_state_dict(vae_decoder_shapes(VAEGeometry()), 1237, device=dev)
        # `scheduler_class.from_config(pipeline.scheduler.config)` (:493-495): betas, prediction_type, timestep_spacing ("leading": the PNDM /
        # DDIM instance's value) and steps_offset are inherited from the model's own scheduler
        scfg_path = os.path.join(mdir, "scheduler", "scheduler_config.json")
        scfg = json.load(open(scfg_path)) if os.path.exists(scfg_path) else None
        sampler = HipSampler(HipUNet(unet_geo, usd, 2 * nv, latent, latent, text_len=clip_geo.max_pos, device=dev),  # (validation images: fp16 pipeline in both modes)
                             HipVAEDecoder(VAEGeometry(), dsd, nv, latent, latent, device=dev), steps=25, guidance=7.5,
                             scheduler_config=scfg, scheduler=args.validation_scheduler)
        del dsd
    del usd
    te_mode = "fp32" if fp32_mode else "autocast"
    teacher = HipTextEncoder(clip_geo, csd, B, mode="fp32" if fp32_mode else "half", device=dev) if args.kpl_weight > 0 else None
    frozen = HipTextEncoder(clip_geo, csd, 1, mode=te_mode, device=dev)
    null_ids = torch.full((1, clip_geo.max_pos), EOS, dtype=torch.int64, device=dev)
    null_ids[0, 0] = BOS
    null = frozen.forward(null_ids, pins=False).clone()  # SD1.x null embedding (the reference ships one only for SD2.1: SURVEY 0.5)
    del frozen
    shipped = os.path.join("assets", "null_emb_sd21base.pt")  # the reference's hard-coded relative path (:649), a [77, 1024] tensor
    if os.path.exists(shipped):
        t = torch.load(shipped, map_location="cpu")
        t = t[0] if t.dim() == 3 else t
        if tuple(t.shape) == tuple(null.shape[-2:]):
            null = t.to(dev, null.dtype).reshape(null.shape)
            logger.info("null embedding loaded from %s", shipped)
    te = HipTextEncoder(clip_geo, csd, B, mode=te_mode, lora_rank=args.lora_rank, n_slots=1, device=dev, seed=args.seed)
    del csd
    te.set_null_embedding(null)
    if teacher is not None:
        teacher.set_null_embedding(null)

    # ---- tokens (:658-694; utils.py:117-214): placeholder vectors first, then augmentation vectors; ids contiguous from 49408
    g = torch.Generator().manual_seed(args.seed or 0)
    added_tokens, aug_token_dict = {}, {}
    n_place = 1
    concept_names = []  # per concept: the list of its placeholder vector names (`concept["instance_token"] = placeholder_tokens`, :691-692)
    if tokenizer is not None:  # the reference's own registration: one vector per BPE piece of the initialiser, rows copied from the pieces
        for c in concepts:  # :665-679
            names, ids = tbdata.add_token(te, tokenizer, c["placeholder_token"], c["initializer_token"])
            concept_names.append(names)
            added_tokens.update(zip(names, ids))
        if args.augment_inversion:
            _, aug_token_dict = tbdata.add_augmentation_tokens(te, tokenizer, "style" if args.augment_ops == "style" else "object")
    else:
        for c in concepts:
            names = multi_vector_names(c["placeholder_token"], n_place)
            concept_names.append(names)
            for name in names:
                added_tokens[name] = te.add_tokens(torch.randint(0, 49406, (1,), generator=g).tolist())[0]
        if args.augment_inversion:
            for tok, n in (AUG_TOKENS_OBJECT if args.augment_ops == "object" else AUG_TOKENS_STYLE):
                for name in multi_vector_names(tok, n):
                    aug_token_dict[name] = te.add_tokens(torch.randint(0, 49406, (1,), generator=g).tolist())[0]
    added_ids = list(added_tokens.values()) + list(aug_token_dict.values())
    if args.validation_prompts and tokenizer is not None:  # log_validation (:502-505): `<i>` -> the i-th concept's placeholder vectors
        for j, prompt in enumerate(args.validation_prompts):
            for ci, names in enumerate(concept_names):
                prompt = prompt.replace(f"<{ci}>", " ".join(names))
            val_ids[j] = tokenizer(prompt, truncation=True, padding="max_length", max_length=tokenizer.model_max_length,
                                   return_tensors="pt").input_ids[0]

    # noise_scheduler = DDPMScheduler.from_pretrained(..., subfolder="scheduler") (:644): prediction type from its config (SD2.1-768: v)
    pred_type = "epsilon"
    sched_cfg = os.path.join(mdir, "scheduler", "scheduler_config.json")
    if os.path.exists(sched_cfg):
        pred_type = json.load(open(sched_cfg)).get("prediction_type", "epsilon")
        if pred_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"Unknown prediction type {pred_type}")  # :1075
    hp = StepHyper(prediction_type=pred_type, use_grad_scaler=not (fp32_mode or bf16_mode), init_scale=1.0 if (fp32_mode or bf16_mode) else 65536.0,
                   grad_accum=args.gradient_accumulation_steps,
                   lr=args.learning_rate * (args.train_batch_size * world if args.scale_lr else 1), emb_lr=args.emb_learning_rate,
                   beta1=args.adam_beta1, beta2=args.adam_beta2, wd=args.adam_weight_decay, eps=args.adam_epsilon,
                   max_grad_norm=args.max_grad_norm, kpl_weight=args.kpl_weight, kpl_type="cos" if args.kpl_type == "cos" else "mse",
                   mixing=(args.augment_ops if args.augment_ops == "object" else "style") if args.mixing else None)
    if args.with_image_prior:
        raise NotImplementedError("--with_image_prior is broken in the reference itself (generate_prior_images is called with 6 of its 7 "
                                  "arguments, SURVEY 0.6): no runnable reference behaviour to match")
    if args.unet_params_to_train not in ("none", "crossattn_kv"):
        raise NotImplementedError(f"--unet_params_to_train {args.unet_params_to_train}: the reference adds adapters only for 'crossattn_kv' (:712); "
                                  "every other value trains nothing in the UNet and only writes an unchanged copy of it to <output_dir>/unet")
    if args.unet_params_to_train == "crossattn_kv":
        if not fp32_mode and args.mixed_precision != "bf16":   # (bf16: the reference trains the bf16-cast adapters, no GradScaler; built in round 6)
            raise NotImplementedError("--unet_params_to_train crossattn_kv needs the fp32 (no --mixed_precision) or the bf16 mode: under fp16 the "
                                      "reference casts the UNet's freshly added LoRA parameters to fp16 (:937) and GradScaler.unscale_ rejects them")
        if args.lora_rank <= 0:
            raise ValueError("--unet_params_to_train crossattn_kv is only reached with --lora_rank > 0 (:700-721)")
        unet.enable_kv_lora(args.lora_rank, seed=None if args.seed is None else args.seed + 1)
        logger.info("Added LoRA to U-Net")  # :721
    # options that change the step's arithmetic and are not built fail loudly instead of silently training something else
    if args.text_encoder_use_attention_mask:
        raise NotImplementedError("--text_encoder_use_attention_mask has no runnable reference behaviour: the collate function hands "
                                  "encode_prompt a Python LIST of masks (textboost/dataset.py:427-454) and `attention_mask.to(device)` "
                                  "(textboost/utils.py:14-15) raises AttributeError in the reference itself")
    step = TextBoostStep(unet, te, teacher, hp, (B, 4, latent, latent), device=dev, world_size=world)
    if pixels is not None or use_images:  # :651-656, :938: the (frozen) VAE; the step then starts from pixel_values (:1027-1037)
        from textboost_amd.vae import HipVAEEncoder, VAEGeometry, vae_encoder_shapes
        vsd = load_local_state_dict(os.path.join(mdir, "vae", "diffusion_pytorch_model.safetensors")) if os.path.isdir(mdir) else None
        if vsd is None:
            logger.warning("no local VAE weights under %s: seeded random-init SD VAE encoder shapes are used", mdir)
            vsd = models.random_state_dict(vae_encoder_shapes(VAEGeometry()), 1236, device=dev)
        step.attach_vae(HipVAEEncoder(VAEGeometry(), vsd, B, 8 * latent, 8 * latent, device=dev))
        del vsd

    # ---- data: latents (+ ids) from disk, else synthetic; rank r takes samples r, r+W, ... (every shard non-empty: SURVEY 0.6)
    dg = torch.Generator().manual_seed(1000 + rank)
    ids_path = os.path.join(args.instance_data_dir or "", "input_ids.pt")
    pids_path = os.path.join(args.instance_data_dir or "", "prior_input_ids.pt")
    inst_ids = torch.load(ids_path) if os.path.exists(ids_path) else None
    prior_ids = torch.load(pids_path) if os.path.exists(pids_path) else None

    feeder, index_stream = None, None
    if use_images:  # :856-890: PairedAugmentation + TextBoostDataset + Wrapper(...).shuffle(seed).repeat(), on the device
        import random as pyrandom

        import numpy as np
        from textboost_amd import augment as tbaug
        if args.augment not in ("none", "pda", "paug"):
            raise NotImplementedError(f"--augment {args.augment}: only PairedAugmentation ('pda' / 'paug') is built")
        pipe = None
        if args.augment in ("pda", "paug"):
            pipe = tbaug.PairedAugmentation(hflip="inversion" if args.augment_inversion else "false", augment_prompt=args.augment_prompt,
                                            inversion=args.augment_inversion, p=args.augment_p, ops=args.augment_ops)
        # the reference formats the template with the LIST of placeholder names (`concept["instance_token"] = placeholder_tokens`, :691-692)
        images = []
        for c, names in zip(concepts, concept_names):  # dataset.py:302-308: every concept's images, each with that concept's token list
            paths = [p for p in tbdata.get_images_path(c["instance_data_dir"], args.num_samples) if p.lower().endswith(tbdata.IMAGE_EXTENSIONS)]
            images += [(tbaug.to_device_image(tbdata.decode_rgb(p)), names) for p in paths]
        feeder = tbaug.DeviceFeeder(images, tokenizer, tbdata.load_templates(args.template), size=args.resolution,
                                    center_crop=args.center_crop, augment_pipe=pipe)
        index_stream = tbdata.IndexStream(len(images), args.seed or 0, rank, world)
        if args.seed is not None:  # set_seed(args.seed) (:601) also seeds `random` and numpy, which the augmentation draws from
            pyrandom.seed(args.seed)
            np.random.seed(args.seed)
        logger.info("device feeder: %d resident instance image(s), template set %r (%d prompts), augment %s", len(images), args.template,
                    len(feeder.templates), args.augment)

    # ---- knowledge-preservation prompts (:892-907): InstructPix2Pix edit prompts + template x class-token prompts + null prompts
    prior_feeder = None
    edit_prompts = os.path.join("data", "human-written-prompts.jsonl")  # the reference's hard-coded relative path (:892)
    if tokenizer is not None and os.path.exists(edit_prompts):
        from textboost_amd.augment import PromptFeeder
        prior_feeder = tbdata.PriorPromptFeeder(tbdata.read_edit_prompts(edit_prompts), feeder.tokenize if feeder is not None else
                                                PromptFeeder(tokenizer), additional_template=args.template,
                                                additional_category=args.class_token, null_prob=args.null_prob, seed=args.seed or 0,
                                                rank=rank, world=world)
        logger.info("prior prompts: %d edit prompts + %d template prompts, null_prob %.2f", len(prior_feeder.data),
                    len(prior_feeder.template_data), args.null_prob)

    prefetcher = None

    def next_batch(it):
        if prefetcher is not None:
            prefetcher.commit()
        elif pixels is not None:
            idx = shard_indices(pixels.shape[0], B, it, rank, world)
            step.pixel_values.copy_(pixels[idx])
            if inst_ids is not None:
                step.input_ids.copy_(inst_ids[[i % inst_ids.shape[0] for i in idx]])
        elif latents is not None:
            idx = shard_indices(latents.shape[0], B, it, rank, world)
            step.x0.copy_(latents[idx])
            if inst_ids is not None:
                step.input_ids.copy_(inst_ids[[i % inst_ids.shape[0] for i in idx]])
        else:
            step.x0.copy_(torch.randn(B, 4, latent, latent, generator=dg))
        if feeder is None and ((latents is None and pixels is None) or inst_ids is None):
            step.input_ids.copy_(synthetic_ids(B, added_ids, dg))
        if prior_feeder is not None:
            step.prior_ids.copy_(prior_feeder.batch(B)["input_ids"])
        elif prior_ids is not None:
            j = torch.randint(0, prior_ids.shape[0], (B,), generator=dg)
            step.prior_ids.copy_(prior_ids[j])
        else:
            step.prior_ids.copy_(synthetic_ids(B, added_ids, dg, prior=True, null_prob=args.null_prob))

    # ---- resume (:959-981)
    first_step = 0
    if args.resume_from_checkpoint:
        path = args.resume_from_checkpoint
        if path == "latest":
            dirs = sorted([d for d in os.listdir(args.output_dir) if d.startswith("checkpoint")], key=lambda x: int(x.split("-")[1]))
            path = os.path.join(args.output_dir, dirs[-1]) if dirs else None
            if path is None:
                print(f"Checkpoint '{args.resume_from_checkpoint}' does not exist. Starting a new training run.")  # :968-972
        else:  # :961-981: only the basename counts, the checkpoint is looked up under --output_dir
            path = os.path.join(args.output_dir, os.path.basename(path.rstrip("/")))
            if not os.path.isdir(path):  # the reference's soft "Starting a new training run" is only reachable for "latest" (:968-972); an
                # explicit name that does not exist fails in accelerator.load_state -- a typo must not silently retrain from step 0
                raise FileNotFoundError(f"--resume_from_checkpoint {args.resume_from_checkpoint!r}: no such checkpoint directory {path}")
        if path:
            ckpt.load_trainer_state(step, path)  # also restores the torch / numpy / `random` generator states the feeder draws from
            first_step = int(os.path.basename(path.rstrip("/")).split("-")[1])
            consumed = first_step * args.gradient_accumulation_steps * B   # every micro batch of every optimizer step drew B samples
            if index_stream is not None:          # the Wrapper streams are pure functions of (seed, position): fast-forward them
                index_stream.take(consumed)
            if prior_feeder is not None:
                prior_feeder.stream.take(consumed)
    if is_main:
        logger.info("mean_norm %.6f | added tokens %s | world %d | per-GPU batch %d | precision %s", step.mean_norm, list(added_tokens) +
                    list(aug_token_dict), world, B, "fp32 (no mixed precision, no GradScaler)" if fp32_mode else ("bf16 mixed precision (no GradScaler)" if bf16_mode else "fp16 mixed precision"))
        print("Mean norm:", step.mean_norm)

    def run_validation(done):
        """log_validation (:453-531): 25 DPM-Solver++ steps, guidance 7.5, num_validation_images per prompt -> validation_{step}.jpg"""
        import copy
        from textboost_amd.sampler import make_image_grid
        te_val = copy.copy(te)       # same parameters, private activation buffers
        te_val._bufs = {}
        te_val.pack_lora()           # the fp16 LoRA operands were packed BEFORE the last optimizer step: refresh them from the fp32 masters,
        #                              so that the images show the weights the checkpoint saves
        ids = val_ids.to(dev).repeat_interleave(args.num_validation_images, dim=0)
        empty = torch.full_like(ids, EOS)
        empty[:, 0] = BOS
        cond = te_val.forward(ids).clone()
        uncond = te_val.forward(empty).clone()
        if getattr(unet, "kv_r", 0):
            # --unet_params_to_train crossattn_kv: log_validation samples with the TRAINED unet (:453-531).  The sampler is a separate fp16 UNet
            # without adapter support: it gets the attn2.to_k / to_v weights with the current adapters folded in (W + scaling * B A in fp32,
            # rounded once to fp16)
            sampler.unet.load_kv_weight(unet.merged_kv_weight())
        if args.seed is not None:
            sampler.generator = torch.Generator(device=dev).manual_seed(args.seed)
        images = sampler.sample(cond, uncond)
        grid = make_image_grid(images, val_ids.shape[0], args.num_validation_images)
        grid.save(os.path.join(args.output_dir, f"validation_{done}.jpg"))
        logger.info("Running validation... wrote validation_%d.jpg (%d images)", done, images.shape[0])

    from textboost_amd.trainer import lr_lambda
    lam = lr_lambda(args.lr_scheduler, args.lr_warmup_steps, args.max_train_steps, lr_init=hp.lr)  # :911-916
    if feeder is None:
        next_batch(0)  # the device feeder consumes random draws per batch: there the graph is captured over the (zero) input buffers
    if args.lr_scheduler != "constant":
        # lambda(k) for k = 0 .. max_train_steps lives on the device and is indexed by the count of SUCCESSFUL optimizer steps, like
        # accelerate's AcceleratedScheduler, which does not advance on an overflow-skipped step
        step.set_lr_table([lam(k) for k in range(args.max_train_steps + 1)])
    else:
        step.set_lr_multiplier(1.0)  # a resumed checkpoint's `state` may carry another schedule's multiplier: the lambdas are this run's (:911-916)
    step.capture(warmup=0)
    if feeder is not None:  # batch k+1 is produced on a side stream while step k runs (augment.PrefetchFeeder)
        from textboost_amd.augment import PrefetchFeeder
        prefetcher = PrefetchFeeder(feeder, B, step.pixel_values, step.input_ids)
        prefetcher.prefetch(index_stream.take(B))
    t0 = time.perf_counter()
    G = args.gradient_accumulation_steps
    for mit in range(first_step * G, args.max_train_steps * G):  # one loop iteration = one batch (:1024-1033); `step` counts optimizer steps
        next_batch(mit)
        synced = step.replay()
        if prefetcher is not None and not synced:
            prefetcher.prefetch(index_stream.take(B))
        if not synced:  # accelerator.sync_gradients is False: gradients keep accumulating (:1039), nothing below runs (:1153)
            continue
        it = mit // G
        done = it + 1
        if is_main and (done % 50 == 0 or done == args.max_train_steps):  # scalars are read off the hot loop
            sc = step.scalars()
            logger.info("step %d loss %.6f mse %.6f kpl %.6f scale %.0f grad_norm %.4f", done, sc["loss"], sc["loss_mse"],
                        sc["loss_kpl"], sc["loss_scale"], sc["grad_norm"])
            print(f"step {done}: loss {sc['loss']:.5f} lr {args.learning_rate}", flush=True)
        if sampler is not None and done % args.validation_steps == 0:  # :1212-1228
            run_validation(done)
        if is_main and done % args.checkpointing_steps == 0:  # :1157-1209
            ckpt.rotate_checkpoints(args.output_dir, args.checkpoints_total_limit)
            cdir = os.path.join(args.output_dir, f"checkpoint-{done}")
            ckpt.save_trainer_state(step, cdir)
            if args.lora_rank > 0:  # :1178-1182
                ckpt.save_text_encoder_adapter(te, os.path.join(cdir, "text_encoder"), mdir)
            ckpt.save_token_embeddings(te, cdir, added_tokens, aug_token_dict if args.augment_inversion else None)
        if prefetcher is not None and done < args.max_train_steps:
            # after the checkpoint, so that the generator states saved in it are those BEFORE the next batch's draws (resume replays them)
            prefetcher.prefetch(index_stream.take(B))
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()  # accelerator.wait_for_everyone() :1235
    if is_main:  # :1236-1266
        if args.unet_params_to_train != "none":  # :1237-1239
            ckpt.save_unet_adapters(unet, os.path.join(args.output_dir, "unet"), mdir)
        if args.lora_rank > 0:
            ckpt.save_text_encoder_adapter(te, os.path.join(args.output_dir, "text_encoder"), mdir)
        ckpt.save_token_embeddings(te, args.output_dir, added_tokens, aug_token_dict if args.augment_inversion else None)
        dt = time.perf_counter() - t0
        logger.info("Training took %.2f seconds", dt)  # :1268-1269
        print(json.dumps({"steps": args.max_train_steps - first_step, "seconds": round(dt, 3),
                          "steps_per_s": round((args.max_train_steps - first_step) / max(dt, 1e-9), 3)}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
