"""load_model (mirror of the reference's videollava/eval/eval.py:15-34)."""
from .builder import load_pretrained_model
from .mm_utils import get_model_name_from_path


def load_model(model_path, model_base, load_8bit=False, load_4bit=False, cache_dir=None, device=None, **engine_kwargs):
    model_name = get_model_name_from_path(model_path)
    tokenizer, model, processor, _ = load_pretrained_model(model_path, model_base, model_name, load_4bit=load_4bit,
                                                           load_8bit=load_8bit, device=device, cache_dir=cache_dir,
                                                           **engine_kwargs)
    model.model.video_tower = None        # as the reference does: only the image tower is used at inference
    return tokenizer, model, processor["image"]
