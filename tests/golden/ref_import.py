"""Import shim for the read-only reference at /root/reference (build container ONLY).

Test infrastructure: lets tests/golden/make_golden.py import the reference's own
Python modules unmodified under the transformers version installed here, so that
golden vectors can be produced by the reference itself.  Nothing here ships to the
GPU box and nothing in the product imports it.  Recipe: SURVEY.md section 8(c).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("TEO_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "videollava"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def install():
    """Make `videollava.*` leaf modules importable without running the package __init__ chains."""
    if "videollava" in sys.modules and getattr(sys.modules["videollava"], "_teo_ref_shim", False):
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for k in [k for k in sys.modules if k == "videollava" or k.startswith("videollava.")]:
        del sys.modules[k]
    base = os.path.join(REF_ROOT, "videollava")
    top = _pkg("videollava", base)
    top._teo_ref_shim = True
    _pkg("videollava.eval", os.path.join(base, "eval"))
    _pkg("videollava.model", os.path.join(base, "model"))
    _pkg("videollava.model.language_model", os.path.join(base, "model", "language_model"))
    _pkg("videollava.model.multimodal_projector", os.path.join(base, "model", "multimodal_projector"))
    enc = os.path.join(base, "model", "multimodal_encoder")
    _pkg("videollava.model.multimodal_encoder", enc)
    _pkg("videollava.model.multimodal_encoder.languagebind", os.path.join(enc, "languagebind"))
    _pkg("videollava.model.multimodal_encoder.languagebind.image", os.path.join(enc, "languagebind", "image"))

    # peft is absent here; the vision path never calls it in eval with lora_r == 0
    if "peft" not in sys.modules:
        peft = types.ModuleType("peft")
        peft.LoraConfig = type("LoraConfig", (), {"__init__": lambda self, *a, **k: None})
        peft.PeftModel = type("PeftModel", (), {})
        peft.get_peft_model = lambda model, cfg: model
        sys.modules["peft"] = peft

    import transformers.models.clip.modeling_clip as mc
    if not hasattr(mc, "_expand_mask"):
        mc._expand_mask = lambda *a, **k: None
    if not hasattr(mc, "clip_loss"):
        mc.clip_loss = lambda *a, **k: None

    # encoder builder: the real one imports every LanguageBind modality; give the factory a
    # hook that returns a tower assembled from the reference's own CLIPVisionTransformer.
    fake = types.ModuleType("videollava.model.multimodal_encoder.builder")
    fake._tower_factory = None

    def build_image_tower(cfg, **kw):
        if fake._tower_factory is None:
            raise RuntimeError("set builder._tower_factory first")
        return fake._tower_factory(cfg)

    fake.build_image_tower = build_image_tower
    fake.build_video_tower = lambda cfg, **kw: None
    sys.modules["videollava.model.multimodal_encoder.builder"] = fake


def import_llava_llama():
    """Import llava_llama.py with AutoConfig/AutoModel registration no-op'd (model_type 'llava' is taken here)."""
    install()
    from transformers import AutoConfig, AutoModelForCausalLM
    r1, r2 = AutoConfig.register, AutoModelForCausalLM.register
    AutoConfig.register = staticmethod(lambda *a, **k: None)
    AutoModelForCausalLM.register = staticmethod(lambda *a, **k: None)
    try:
        mod = importlib.import_module("videollava.model.language_model.llava_llama")
    finally:
        AutoConfig.register, AutoModelForCausalLM.register = r1, r2
    return mod
