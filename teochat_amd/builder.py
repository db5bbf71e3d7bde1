"""Checkpoint loading (mirror of the reference's videollava/model/builder.py:27-171, merged-checkpoint branch).

load_pretrained_model(model_path, model_base, model_name, ...) -> (tokenizer, model, {'image': proc, 'video': None}, context_len)

Supported sources:
  * a directory with config.json + *.safetensors shards (HF layout; key names as in SURVEY.md section 8a row H17);
  * "synthetic:<preset>" -- random weights of a named architecture (no network here for real ones):
        synthetic:teochat-7b   LLaMA-2-7B + CLIP-ViT-L/14 shapes
        synthetic:tiny         a KB-scale model for smoke tests
LoRA / bitsandbytes branches of the reference are out of scope: `load_8bit/load_4bit` are accepted and
ignored with a warning (bf16 weights are loaded instead); a LoRA checkpoint raises NotImplementedError.
"""
import glob
import os
import warnings

import torch

from .config import LlavaConfig, VisionConfig, teochat_7b_config
from .engine import TeoEngine
from .model import LlavaLlamaForCausalLM
from .processor import TeoImageProcessor
from .tokenizer_stub import ByteTokenizer


def tiny_config():
    return LlavaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=2, vocab_size=512, mm_hidden_size=128, max_position_embeddings=2048,
                       vision_config=VisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                                  num_attention_heads=2, hidden_act="gelu"))


class LazySafetensors:
    """dict-like view over the shards of an HF checkpoint directory."""

    def __init__(self, model_dir, device):
        from safetensors import safe_open
        self._open = safe_open
        self.device = device
        self.where = {}
        for f in sorted(glob.glob(os.path.join(model_dir, "*.safetensors"))):
            with safe_open(f, framework="pt", device="cpu") as h:
                for k in h.keys():
                    self.where[k] = f
        if not self.where:
            raise FileNotFoundError(f"no *.safetensors under {model_dir}")

    def __contains__(self, k):
        return k in self.where

    def keys(self):
        return self.where.keys()

    def __getitem__(self, k):
        if k not in self.where:
            raise KeyError(k)
        with self._open(self.where[k], framework="pt", device="cpu") as h:
            return h.get_tensor(k).to(self.device)


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto",
                          device="cuda", cache_dir=None, dtype=torch.bfloat16, max_seq=None, seed=2, weight_format=None):
    if device in (None, "cuda"):
        device = "cuda:0"
    if load_8bit or load_4bit:
        warnings.warn("bitsandbytes int8/nf4 loading is CUDA-only and out of scope; loading bf16 weights instead")
    if "lora" in model_name.lower() and model_base is not None:
        raise NotImplementedError("LoRA checkpoints must be merged first (scripts/merge_lora_weights.py in the reference)")
    if model_path.startswith("synthetic:"):
        from .synthetic import synthetic_state_dict
        preset = model_path.split(":", 1)[1]
        if preset in ("teochat-7b", "teochat", "llava-7b"):
            cfg = teochat_7b_config()
        elif preset == "tiny":
            cfg = tiny_config()
        else:
            raise ValueError(f"unknown synthetic preset {preset!r}")
        sd = synthetic_state_dict(cfg, seed=seed, std=0.02 if preset != "tiny" else 0.08, dtype=dtype, device=device)
        tokenizer = ByteTokenizer()
    else:
        if "llava" not in model_name.lower() and "teochat" not in model_name.lower():
            raise ValueError(f"Unsupported model name {model_name!r}: expected a llava/teochat checkpoint (builder.py:33)")
        cfg = LlavaConfig.from_pretrained(model_path)
        sd = LazySafetensors(model_path, device)
        if any(".lora_A." in k for k in sd.keys()):
            raise NotImplementedError("peft-wrapped tower / LoRA tensors found; merge them first")
        if os.path.exists(os.path.join(model_path, "tokenizer.model")):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False)
        else:
            tokenizer = ByteTokenizer()
    engine = TeoEngine(sd, cfg, dtype=dtype, device=device, max_seq=max_seq, weight_format=weight_format)
    del sd
    image_processor = TeoImageProcessor(size=cfg.vision_config.image_size)
    model = LlavaLlamaForCausalLM(cfg, engine, image_processor)
    context_len = getattr(cfg, "max_sequence_length", 2048)
    return tokenizer, model, {"image": image_processor, "video": None}, context_len
