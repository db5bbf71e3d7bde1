"""Configuration objects carrying the reference's HF config attribute names.

LlavaConfig mirrors language_model/llava_llama.py:29-30 (a LlamaConfig with model_type "llava") plus the mm_* knobs
the reference persists on it (llava_arch.py:97-107, train.py:1061-1086).  VisionConfig mirrors the vision half of
languagebind/image/configuration_image.py:128-250 (only the fields the forward path reads).
"""
import json
import os
from types import SimpleNamespace


class VisionConfig(SimpleNamespace):
    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                 num_channels=3, image_size=224, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, **kw):
        super().__init__(hidden_size=hidden_size, intermediate_size=intermediate_size,
                         num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                         num_channels=num_channels, image_size=image_size, patch_size=patch_size,
                         hidden_act=hidden_act, layer_norm_eps=layer_norm_eps)
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_patches(self):
        return (self.image_size // self.patch_size) ** 2


class LlavaConfig(SimpleNamespace):
    model_type = "llava"

    def __init__(self, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                 num_key_value_heads=None, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0,
                 max_position_embeddings=4096, bos_token_id=1, eos_token_id=2, pad_token_id=None,
                 mm_image_tower="LanguageBind/LanguageBind_Image", mm_video_tower=None, mm_hidden_size=1024,
                 mm_projector_type="mlp2x_gelu", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                 mm_use_im_start_end=False, mm_use_im_patch_token=False, tokenizer_model_max_length=None,
                 tokenizer_padding_side="right", image_aspect_ratio=None, vision_config=None, **kw):
        super().__init__()
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        self.vocab_size = vocab_size
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.max_position_embeddings = max_position_embeddings
        self.bos_token_id, self.eos_token_id, self.pad_token_id = bos_token_id, eos_token_id, pad_token_id
        self.mm_image_tower = mm_image_tower
        self.mm_video_tower = mm_video_tower
        self.mm_hidden_size = mm_hidden_size
        self.mm_projector_type = mm_projector_type
        self.mm_vision_select_layer = mm_vision_select_layer
        self.mm_vision_select_feature = mm_vision_select_feature
        self.mm_use_im_start_end = mm_use_im_start_end
        self.mm_use_im_patch_token = mm_use_im_patch_token
        self.tokenizer_model_max_length = tokenizer_model_max_length
        self.tokenizer_padding_side = tokenizer_padding_side
        self.image_aspect_ratio = image_aspect_ratio
        if isinstance(vision_config, dict):
            vision_config = VisionConfig(**vision_config)
        self.vision_config = vision_config or VisionConfig(hidden_size=mm_hidden_size)
        self.vision_config_resolved = True        # False only for a config.json without a vision_config (from_json_file)
        self.pretraining_tp = kw.pop("pretraining_tp", 1)
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.pop("architectures", None)
        cfg = cls(**d)
        # A llava/teochat config.json carries no vision_config: the reference takes the tower's architecture from the
        # `mm_image_tower` repo (LanguageBindImageConfig, languagebind/__init__.py:108,113).  Never default it silently
        # (hidden_act decides the MLP activation): builder.resolve_image_tower() fills it in or fails loudly.
        cfg.vision_config_resolved = "vision_config" in d
        return cfg

    @classmethod
    def from_pretrained(cls, model_dir):
        return cls.from_json_file(os.path.join(model_dir, "config.json"))

    def to_dict(self):
        d = dict(self.__dict__)
        d.pop("vision_config_resolved", None)
        d["vision_config"] = dict(self.vision_config.__dict__)
        d["model_type"] = self.model_type
        return d


def vision_config_from_tower_dir(tower_dir):
    """VisionConfig from a LanguageBind_Image checkpoint directory: the `vision_config` block of its config.json
    (configuration_image.py:105-123), incl. hidden_act and the LoRA rank/alpha of a peft-wrapped encoder (:73-75)."""
    with open(os.path.join(tower_dir, "config.json")) as f:
        d = json.load(f)
    v = d.get("vision_config", d)
    if "hidden_size" not in v or "hidden_act" not in v:
        raise ValueError(f"{tower_dir}/config.json has no usable vision_config (hidden_size / hidden_act missing)")
    keep = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_channels", "image_size",
            "patch_size", "hidden_act", "layer_norm_eps", "lora_r", "lora_alpha", "add_time_attn", "num_frames")
    return VisionConfig(**{k: v[k] for k in keep if k in v})


def teochat_7b_config(**over):
    """LLaMA-2-7B + CLIP-ViT-L/14 (LanguageBind_Image) shapes; `hidden_act` of the tower is checkpoint-dependent."""
    cfg = LlavaConfig(vision_config=VisionConfig(hidden_act=over.pop("vision_hidden_act", "gelu")))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg
