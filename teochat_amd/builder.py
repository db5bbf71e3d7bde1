"""Checkpoint loading (mirror of the reference's videollava/model/builder.py:27-171, merged-checkpoint branch).

load_pretrained_model(model_path, model_base, model_name, ...) -> (tokenizer, model, {'image': proc, 'video': None}, context_len)

Supported sources:
  * a directory with config.json + *.safetensors shards (HF layout; key names as in SURVEY.md section 8a row H17);
  * "synthetic:<preset>" -- random weights of a named architecture (no network here for real ones):
        synthetic:teochat-7b   LLaMA-2-7B + CLIP-ViT-L/14 shapes
        synthetic:tiny         a KB-scale model for smoke tests
        synthetic:teochat-7b-anchored   the same random stack with 8 restructured lm_head rows so that greedy decisions have
                               margins above the bf16 noise (full-size token-stream tests; see synthetic.anchor_gains)
  * the reference's other two branches (builder.py:37-72 and :73-88):
        LoRA       model_name contains "lora" and model_base is given: base weights from model_base, then
                   non_lora_trainables.bin (prefix rules of builder.py:58-61), then the peft adapter of model_path merged
                   as W += (lora_alpha / r) * B @ A  (= PeftModel.merge_and_unload, builder.py:63-70);
        projector  model_base given without "lora": base weights + mm_projector.bin (builder.py:84-86).
    A peft-wrapped image tower inside a checkpoint (modeling_image.py:775-793: `...encoder.base_model.model.layers...
    q_proj.base_layer.weight` + `lora_A/B.default.weight`) is merged the same way with the vision config's lora_r / lora_alpha.
bitsandbytes is CUDA-only: `load_8bit/load_4bit` are accepted and ignored with a warning (bf16 weights are loaded).
"""
import json
import re
import glob
import math
import os
import warnings

import torch

from .config import LlavaConfig, VisionConfig, teochat_7b_config, vision_config_from_tower_dir
from .engine import TeoEngine
from .model import LlavaLlamaForCausalLM
from .processor import TeoImageProcessor
from .tokenizer_stub import ByteTokenizer


def tiny_config():
    return LlavaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=2, vocab_size=512, mm_hidden_size=128, max_position_embeddings=2048,
                       vision_config=VisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                                  num_attention_heads=2, hidden_act="gelu"))


class CheckpointDir:
    """dict-like, lazy view over every tensor file of an HF checkpoint directory: *.safetensors shards and
    pytorch_model*.bin (the reference's checkpoints come in either form)."""

    def __init__(self, model_dir, device):
        from safetensors import safe_open
        self._open = safe_open
        self.device = device
        self.where = {}
        self._bins = {}
        for f in sorted(glob.glob(os.path.join(model_dir, "*.safetensors"))):
            if os.path.basename(f).startswith("adapter_model"):
                continue
            with safe_open(f, framework="pt", device="cpu") as h:
                for k in h.keys():
                    self.where[k] = f
        for f in sorted(glob.glob(os.path.join(model_dir, "pytorch_model*.bin"))):
            blob = torch.load(f, map_location="cpu", weights_only=True)
            self._bins[f] = blob
            for k in blob:
                self.where.setdefault(k, f)
        if not self.where:
            raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {model_dir}")

    def __contains__(self, k):
        return k in self.where

    def keys(self):
        return self.where.keys()

    def __getitem__(self, k):
        if k not in self.where:
            raise KeyError(k)
        f = self.where[k]
        if f in self._bins:
            return self._bins[f][k].to(self.device)
        with self._open(f, framework="pt", device="cpu") as h:
            return h.get_tensor(k).to(self.device)


def _load_tensor_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def _strip_non_lora_prefixes(blob):
    """builder.py:58-61 / merge_lora_weights.py:16-18."""
    blob = {(k[11:] if k.startswith("base_model.") else k): v for k, v in blob.items()}
    if any(k.startswith("model.model.") for k in blob):
        blob = {(k[6:] if k.startswith("model.") else k): v for k, v in blob.items()}
    return blob


_LORA_RE = re.compile(r"^(?P<stem>.*)\.lora_(?P<ab>[AB])(?:\.(?P<adapter>[^.]+))?\.weight$")


class MergedCheckpoint:
    """A base tensor source + replacement tensors + LoRA pairs, resolved lazily per key:
    W[k] = (override[k] or base[k]) + scale * B @ A   (fp32 product, rounded once to the stored dtype)."""

    def __init__(self, base, overrides=None, lora=None, scale=1.0, rename=None):
        self.base = base
        self.overrides = dict(overrides or {})
        self.lora = dict(lora or {})          # clean key -> (A [r, in], B [out, r], scale)
        self.scale = scale
        self.rename = dict(rename or {})      # clean key -> key in the base source

    def keys(self):
        ks = set(self.overrides) | set(self.rename)
        ks |= {k for k in self.base.keys() if ".lora_" not in k and ".base_layer." not in k and "base_model.model." not in k}
        return ks

    def __contains__(self, k):
        return k in self.overrides or k in self.rename or k in self.base

    def __getitem__(self, k):
        if k in self.overrides:
            w = self.overrides[k]
            dev = getattr(self.base, "device", w.device)
            w = w.to(dev)
        else:
            w = self.base[self.rename.get(k, k)]
        pair = self.lora.get(k)
        if pair is not None:
            A, B, sc = pair
            delta = (B.to(w.device, torch.float32) @ A.to(w.device, torch.float32)) * sc
            w = (w.to(torch.float32) + delta).to(w.dtype)
        return w


def merge_lora_adapter(base, adapter_dir):
    """peft adapter directory (adapter_config.json + adapter_model.{safetensors,bin}) over `base`:
    what PeftModel.from_pretrained(...).merge_and_unload() leaves in the model (builder.py:63-70)."""
    with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
        acfg = json.load(f)
    r, alpha = int(acfg["r"]), float(acfg["lora_alpha"])
    if acfg.get("fan_in_fan_out"):
        raise NotImplementedError("fan_in_fan_out LoRA adapters (Conv1D layers) do not occur in LLaMA")
    scale = alpha / math.sqrt(r) if acfg.get("use_rslora") else alpha / r
    files = [os.path.join(adapter_dir, n) for n in ("adapter_model.safetensors", "adapter_model.bin")]
    files = [f for f in files if os.path.exists(f)]
    if not files:
        raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin under {adapter_dir}")
    blob = _load_tensor_file(files[0])
    halves = {}
    for k, v in blob.items():
        m = _LORA_RE.match(k)
        if not m:
            continue
        stem = m.group("stem")
        if stem.startswith("base_model.model."):
            stem = stem[len("base_model.model."):]
        halves.setdefault(stem + ".weight", {})[m.group("ab")] = v
    lora = {}
    for k, ab in halves.items():
        if "A" not in ab or "B" not in ab:
            raise ValueError(f"incomplete LoRA pair for {k}")
        if k not in base:
            raise KeyError(f"LoRA target {k} is not in the base checkpoint")
        lora[k] = (ab["A"], ab["B"], scale)
    return lora


def unwrap_peft_tower(source, vision_cfg):
    """Keys of a peft-wrapped LanguageBind encoder saved inside a checkpoint (modeling_image.py:775-793):
    `<enc>.base_model.model.layers.N.self_attn.q_proj.base_layer.weight`, `...lora_A.default.weight`, ... ->
    plain `<enc>.layers.N.self_attn.q_proj.weight` with the adapter merged (scale lora_alpha / lora_r)."""
    keys = list(source.keys())
    if not any(".lora_A." in k for k in keys):
        return source
    r = int(getattr(vision_cfg, "lora_r", 0) or 0)
    alpha = float(getattr(vision_cfg, "lora_alpha", r) or r)
    if r <= 0:
        raise ValueError("checkpoint holds LoRA tensors for the image tower but vision_config.lora_r is 0")
    scale = alpha / r

    def clean(k):
        return k.replace(".base_model.model.", ".").replace(".base_layer.", ".")

    rename, halves = {}, {}
    for k in keys:
        m = _LORA_RE.match(k)
        if m:
            halves.setdefault(clean(m.group("stem")) + ".weight", {})[m.group("ab")] = k
        elif clean(k) != k:
            rename[clean(k)] = k
    lora = {}
    for k, ab in halves.items():
        if "A" not in ab or "B" not in ab:
            raise ValueError(f"incomplete LoRA pair for {k}")
        lora[k] = (source[ab["A"]], source[ab["B"]], scale)
    return MergedCheckpoint(source, lora=lora, rename=rename)


def open_checkpoint(model_path, model_base, model_name, device):
    """The tensor source for the three branches of builder.py:33-112 (LoRA / projector-only / merged)."""
    lname = model_name.lower()
    if "lora" in lname and model_base is None:
        warnings.warn("There is `lora` in model name but no `model_base` is provided. If you are loading a LoRA model, "
                      "please provide the `model_base` argument.")
    if "lora" in lname and model_base is not None:
        base = CheckpointDir(model_base, device)
        overrides = {}
        nl = os.path.join(model_path, "non_lora_trainables.bin")
        if os.path.exists(nl):
            overrides = _strip_non_lora_prefixes(torch.load(nl, map_location="cpu", weights_only=True))
        probe = MergedCheckpoint(base, overrides)
        return MergedCheckpoint(base, overrides, lora=merge_lora_adapter(probe, model_path))
    if model_base is not None:
        base = CheckpointDir(model_base, device)
        proj = torch.load(os.path.join(model_path, "mm_projector.bin"), map_location="cpu", weights_only=True)
        return MergedCheckpoint(base, proj)
    return CheckpointDir(model_path, device)


TOWER_KEY = "model.image_tower.image_tower.embeddings.class_embedding"


class TowerBackedCheckpoint:
    """`main` for every key it holds; `model.image_tower.image_tower.X` falls back to `vision_model.X` of the image
    tower's own checkpoint -- what `image_tower.load_model()` does in the reference when the LLM checkpoint carries no
    tower weights (builder.py:136-147 -> languagebind/__init__.py:112-119)."""
    PRE = "model.image_tower.image_tower."

    def __init__(self, main, tower):
        self.main, self.tower = main, tower
        self.device = getattr(main, "device", None)

    def _map(self, k):
        return "vision_model." + k[len(self.PRE):] if k.startswith(self.PRE) else None

    def keys(self):
        ks = set(self.main.keys())
        ks |= {self.PRE + k[len("vision_model."):] for k in self.tower.keys() if k.startswith("vision_model.")}
        return ks

    def __contains__(self, k):
        return k in self.main or (self._map(k) is not None and self._map(k) in self.tower)

    def __getitem__(self, k):
        if k in self.main:
            return self.main[k]
        m = self._map(k)
        if m is not None and m in self.tower:
            return self.tower[m]
        raise KeyError(k)


def find_tower_dir(cfg, model_path, model_base=None):
    """Local directory of the `mm_image_tower` checkpoint (there is no hub access): the name itself if it is a directory,
    $TEOCHAT_IMAGE_TOWER, or a sibling / child directory named like the repo (LanguageBind/LanguageBind_Image ->
    LanguageBind_Image)."""
    name = getattr(cfg, "mm_image_tower", None)
    cands = [os.environ.get("TEOCHAT_IMAGE_TOWER")]
    if name:
        base = os.path.basename(str(name).rstrip("/"))
        cands += [name]
        for root in (model_path, model_base):
            if root:
                cands += [os.path.join(root, base), os.path.join(os.path.dirname(os.path.abspath(root)), base)]
    for c in cands:
        if c and os.path.isdir(c) and os.path.exists(os.path.join(c, "config.json")):
            return c
    return None


def resolve_image_tower(cfg, source, model_path, model_base, device):
    """Fill in what the reference takes from the tower repo: the vision config (when config.json has none) and the tower
    weights (when the checkpoint has none).  Fails loudly instead of defaulting the activation."""
    need_cfg = not getattr(cfg, "vision_config_resolved", True)
    need_w = TOWER_KEY not in source and TOWER_KEY.replace(".embeddings.", ".embeddings.base_layer.") not in source
    if not need_cfg and not need_w:
        return source
    tower_dir = find_tower_dir(cfg, model_path, model_base)
    if tower_dir is None:
        what = " and ".join(x for x, n in (("a `vision_config` block in config.json", need_cfg),
                                            ("`model.image_tower.image_tower.*` weights in the checkpoint", need_w)) if n)
        raise FileNotFoundError(
            f"cannot resolve the image tower {getattr(cfg, 'mm_image_tower', None)!r}: the checkpoint lacks {what} and no local "
            "copy of the tower was found (set TEOCHAT_IMAGE_TOWER=<dir with config.json + weights>, or place it next to the "
            "checkpoint).  The tower's hidden_act / LoRA rank are never guessed.")
    if need_cfg:
        cfg.vision_config = vision_config_from_tower_dir(tower_dir)
        cfg.vision_config_resolved = True
    if need_w:
        source = TowerBackedCheckpoint(source, CheckpointDir(tower_dir, device))
    return source


def load_generation_config(model, model_dir):
    """generation_config.json of the checkpoint -> model.generation_config (what GenerationMixin.generate falls back to for
    every knob the caller leaves unset: LLaMA-2 ships do_sample / temperature 0.6 / top_p 0.9; top_k stays HF's default 50).
    The reference's call (eval/inference.py:64-72) passes do_sample, temperature and max_new_tokens only, so top_p / top_k
    of a real checkpoint come from this file."""
    path = os.path.join(model_dir, "generation_config.json") if os.path.isdir(str(model_dir)) else None
    if not path or not os.path.exists(path):
        return False
    with open(path) as f:
        d = json.load(f)
    for k in ("do_sample", "temperature", "top_k", "top_p", "eos_token_id", "bos_token_id", "pad_token_id", "max_length"):
        if k in d:
            setattr(model.generation_config, k, d[k])
    return True


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", cache_dir=None, dtype=None, max_seq=None, seed=2, weight_format=None):
    """dtype None = the reference's choice where it has one: a real checkpoint runs in torch.float16 (builder.py:105 passes
    torch_dtype=torch.float16 whatever the file holds; eval/inference.py:53 casts the frames to match) -- unless fp8 decode weights are
    requested, which go with bfloat16 -- and the synthetic presets in torch.bfloat16 (BASELINE.json's headline dtype)."""
    if device in (None, "cuda"):
        device = "cuda:0"
    if dtype is None:
        dtype = torch.bfloat16 if (model_path.startswith("synthetic:") or weight_format == "fp8") else torch.float16
    if load_8bit or load_4bit:
        warnings.warn("bitsandbytes int8/nf4 loading is CUDA-only and out of scope; loading bf16 weights instead")
    if model_path.startswith("synthetic:"):
        from .synthetic import synthetic_state_dict
        preset = model_path.split(":", 1)[1]
        anchored = preset.endswith("-anchored")          # decisive-margin variant for full-size token checks (synthetic.py)
        if anchored:
            preset = preset[:-len("-anchored")]
        realistic = preset.endswith("-realistic")        # trained-checkpoint statistics: heavy tails, massive activations, non-unit gains
        if realistic:
            preset = preset[:-len("-realistic")]
        if preset in ("teochat-7b", "teochat", "llava-7b"):
            cfg = teochat_7b_config()
        elif preset == "tiny":
            cfg = tiny_config()
        else:
            raise ValueError(f"unknown synthetic preset {preset!r}")
        sd = synthetic_state_dict(cfg, seed=seed, std=0.02 if preset != "tiny" else 0.08, dtype=dtype, device=device, anchored=anchored, realistic=realistic)
        tokenizer = ByteTokenizer()
    else:
        if "llava" not in model_name.lower() and "teochat" not in model_name.lower():
            raise ValueError(f"Unsupported model name {model_name!r}: expected a llava/teochat checkpoint (builder.py:33)")
        cfg = LlavaConfig.from_pretrained(model_path)           # the LoRA / projector branches read the config of model_path too
        src = resolve_image_tower(cfg, open_checkpoint(model_path, model_base, model_name, device), model_path, model_base, device)
        sd = unwrap_peft_tower(src, cfg.vision_config)
        tok_dir = model_base if model_base is not None else model_path       # builder.py:40,82 vs :111
        if os.path.exists(os.path.join(tok_dir, "tokenizer.model")):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tok_dir, use_fast=False)
        else:
            tokenizer = ByteTokenizer()
    engine = TeoEngine(sd, cfg, dtype=dtype, device=device, max_seq=max_seq, weight_format=weight_format)
    del sd
    image_processor = TeoImageProcessor(size=cfg.vision_config.image_size, engine=engine)     # uint8 frames -> device kernel
    model = LlavaLlamaForCausalLM(cfg, engine, image_processor)
    # GenerationMixin's fallback knobs belong to the model object, which the LoRA / projector branches build with
    # from_pretrained(model_base, ...) (builder.py:51,83): generation_config.json is read where the tokenizer is (tok_dir)
    gen_dir = model_base if (model_base is not None and not model_path.startswith("synthetic:")) else model_path
    if not load_generation_config(model, gen_dir) and gen_dir != model_path:
        load_generation_config(model, model_path)
    context_len = getattr(cfg, "max_sequence_length", 2048)
    return tokenizer, model, {"image": image_processor, "video": None}, context_len
