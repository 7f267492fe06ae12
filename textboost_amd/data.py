"""Host side of the instance-image data path: what `train_textboost.py` of the reference sets up around
`TextBoostDataset` -- template sets (textboost/dataset.py:13-76, selected by `--template`, :205-222 / :289-298), the image list
(`get_images_path`, :96-105), token registration with a real tokenizer (`add_token` / `add_augmentation_tokens`, textboost/utils.py:117-214)
and the `Wrapper` index stream (:828-872).  Pixels never pass through here: images are decoded once and handed to augment.DeviceFeeder."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# initialiser words of the augmentation tokens (textboost/utils.py:180-197): part of the CLI contract of --augment_inversion
AUGMENTATION_INITIALIZERS = {
    "object": {"<grayscale>": "grayscale", "<zoom-in>": "zoom in", "<zoom-out>": "far away", "<collage>": "photo collage", "<crop>": "crop",
               "<hflip>": "ktn", "<left>": "pll", "<right>": "ucd"},
    "style": {"<hflip>": "ktn"},
}


def load_templates(name: str) -> List[str]:
    """`--template`: one of the named sets, else the string itself is the only template (dataset.py:289-298)."""
    with open(os.path.join(_HERE, "prompt_templates.json")) as f:
        sets = json.load(f)
    return list(sets[name]) if name in sets else [name]


def get_images_path(data_root: str, max_samples: Optional[int] = None) -> List[str]:
    """dataset.py:96-105: every entry of the directory, sorted, optionally the first `max_samples`."""
    if not os.path.isdir(data_root):
        raise ValueError("Data root doesn't exists.")
    paths = sorted(os.path.join(data_root, p) for p in os.listdir(data_root))
    return paths[:max_samples] if max_samples is not None else paths


IMAGE_EXTENSIONS = (".jpg", ".jpeg", ".png", ".webp", ".bmp")


def has_instance_images(data_root: Optional[str]) -> bool:
    return bool(data_root) and os.path.isdir(data_root) and any(p.lower().endswith(IMAGE_EXTENSIONS) for p in os.listdir(data_root))


def decode_rgb(path: str) -> np.ndarray:
    """dataset.py:357-361: `Image.open` + `exif_transpose` + RGB -> uint8 [H, W, 3].  The only Pillow call left on the path, once per image."""
    from PIL import Image, ImageOps
    image = ImageOps.exif_transpose(Image.open(path))
    if image.mode != "RGB":
        image = image.convert("RGB")
    return np.asarray(image)


def multi_vector_names(token: str, n: int) -> List[str]:
    """utils.py:133-141: `<x>` -> `<x_0>`, `<x_1>`, ... when the initialiser has several BPE pieces."""
    if n == 1:
        return [token]
    if token.endswith(">"):
        return [f"{token[:-1]}_{i}>" for i in range(n)]
    return [token] + [f"{token}_{i}" for i in range(1, n)]


def add_token(text_encoder, tokenizer, placeholder_token: str, initializer_token: str) -> Tuple[List[str], List[int]]:
    """utils.py:117-166 with a real tokenizer: one placeholder vector per BPE piece of the initialiser, rows copied from the pieces."""
    init_ids = tokenizer.encode(initializer_token, add_special_tokens=False)
    names = multi_vector_names(placeholder_token, len(init_ids))
    if tokenizer.add_tokens(names) != len(init_ids):
        raise ValueError(f"The tokenizer already contains the token {placeholder_token}. Please pass a different"
                         " `placeholder_token` that is not already in the tokenizer.")
    ids = tokenizer.convert_tokens_to_ids(names)
    new = text_encoder.add_tokens(list(init_ids))
    if list(new) != list(ids):
        raise ValueError(f"tokenizer ids {ids} and embedding rows {new} disagree (tokenizer vocabulary != embedding table size)")
    return names, list(ids)


def add_augmentation_tokens(text_encoder, tokenizer, aug_type: str = "object") -> Tuple[List[int], Dict[str, int]]:
    """utils.py:169-214."""
    assert aug_type in ("object", "style"), f"aug_type must be either 'object' or 'style', but is {aug_type}"
    ids: List[int] = []
    table: Dict[str, int] = {}
    for placeholder, init in AUGMENTATION_INITIALIZERS[aug_type].items():
        n = len(tokenizer.encode(init, add_special_tokens=False))
        _, new = add_token(text_encoder, tokenizer, placeholder, init)
        ids += new
        if n > 1:
            for i, t in enumerate(new):
                table[placeholder.replace(">", f"_{i}>")] = t
        else:
            table[placeholder] = new[0]
    return ids, table


class IndexStream:
    """`Wrapper(dataset, drop_last=False).shuffle(seed).repeat()` (dataset.py:828-872) for one consumer per rank: an endless index stream;
    epoch e reshuffles the SAME key array in place with `default_rng(seed + e)` (cumulative, as the reference does), pads to a multiple of
    the world size with the head of the epoch, and rank r takes every `world`-th index starting at r."""

    def __init__(self, n: int, seed: int, rank: int = 0, world: int = 1, shuffle: bool = True):
        self.keys = np.arange(n)
        self.seed, self.rank, self.world, self.shuffle = seed, rank, world, shuffle
        self.epoch = 0
        self._queue: List[int] = []

    def _refill(self):
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(self.keys)
        keys = self.keys
        # pad to a multiple of the world size by TILING the keys (the reference's Wrapper pads with keys[:world - rem], which with fewer
        # images than ranks leaves ranks without data -- its one-image multi-GPU runs hang): every rank gets >= 1 index
        n = -(-len(keys) // self.world) * self.world
        idx = keys if n == len(keys) else np.resize(keys, n)
        self._queue = [int(i) for i in idx[self.rank::self.world]]
        self.epoch += 1

    def take(self, n: int) -> List[int]:
        out = []
        while len(out) < n:
            if not self._queue:
                self._refill()
            out.append(self._queue.pop(0))
        return out


def read_edit_prompts(json_file: str, num_samples: Optional[int] = None) -> List[str]:
    """`InstructPix2PixDataset.__init__` (dataset.py:162-175): one JSON object per line; its "input" and, unless null / "NONE", its "output"."""
    prompts: List[str] = []
    with open(json_file, "r") as f:
        for line in f.readlines():
            rec = json.loads(line)
            prompts.append(rec["input"])
            out = rec["output"]
            if out is not None and out != "NONE":
                prompts.append(out)
    return prompts[:num_samples] if num_samples is not None else prompts


class PriorPromptFeeder:
    """`PriorDataset` + `collate_fn` behind `Wrapper(drop_last=True).shuffle(seed).repeat()` (dataset.py:196-269, train_textboost.py:892-907):
    the knowledge-preservation prompts.  Per sample one `random.random()` draw: < null_prob -> "", < null_prob + template_prob -> a
    `random.choice` of template x class-token prompts, else the source prompt at the stream's index.  Token ids come from the shared cache."""

    def __init__(self, source_prompts: Sequence[str], tokenize, additional_template=None, additional_category=None, template_prob=0.1,
                 null_prob=0.1, seed: int = 0, rank: int = 0, world: int = 1):
        import random
        self._random = random
        self.data = list(source_prompts)
        if len(self.data) < world:
            raise ValueError("fewer prior prompts than ranks: Wrapper(drop_last=True) would leave a rank without data")
        self.tokenize = tokenize
        self.template_prob, self.null_prob = template_prob, null_prob
        categories = additional_category if isinstance(additional_category, list) else [additional_category]
        self.template_data = [t.format(c) for t in load_templates(additional_template) for c in categories]
        self.stream = _DropLastIndexStream(len(self.data), seed, rank, world)

    def batch(self, n: int):
        import torch
        prompts, ids = [], []
        for index in self.stream.take(n):
            r = self._random.random()
            if r < self.null_prob:
                prompt = ""
            elif r < self.null_prob + self.template_prob:
                prompt = self._random.choice(self.template_data)
            else:
                prompt = self.data[index]
            prompts.append(prompt)
            ids.append(self.tokenize(prompt))
        return {"prompt": prompts, "input_ids": torch.cat(ids, dim=0)}


class _DropLastIndexStream(IndexStream):
    """Wrapper(drop_last=True): the tail that does not divide by the world size is dropped instead of padded (dataset.py:863-865)."""

    def _refill(self):
        if self.shuffle:
            np.random.default_rng(self.seed + self.epoch).shuffle(self.keys)
        rem = len(self.keys) % self.world
        idx = self.keys if rem == 0 else self.keys[:-rem]
        self._queue = [int(i) for i in idx[self.rank::self.world]]
        self.epoch += 1
