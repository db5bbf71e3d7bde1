"""Single-example inference driver (mirror of the reference's videollava/eval/inference.py:11-77).

run_inference_single keeps the reference's positional order and defaults; `do_sample` is an extra trailing
keyword (the reference hard-codes do_sample=True, inference.py:67) so parity runs can ask for greedy decoding.
"""
import re
from datetime import datetime

import torch

from .constants import DEFAULT_IMAGE_TOKEN, DEFAULT_VIDEO_TOKEN, IMAGE_TOKEN_INDEX
from .conversation import SeparatorStyle, conv_templates
from .mm_utils import KeywordsStoppingCriteria, tokenizer_image_token


def replace_video_token(prompt, image_paths, prompt_strategy):
    n = len(image_paths)
    if prompt_strategy is None:
        expansion = DEFAULT_IMAGE_TOKEN * n
    elif prompt_strategy == "interleave":
        expansion = "".join(f"Image {k}: {DEFAULT_IMAGE_TOKEN}" for k in range(1, n + 1))
    else:
        raise ValueError(f"Unknown prompt strategy: {prompt_strategy}")
    return prompt.replace(DEFAULT_VIDEO_TOKEN, expansion)


def build_prompt(inp, image_paths, conv_mode="v1", prompt_strategy="interleave", chronological_prefix=True):
    conv = conv_templates[conv_mode].copy()
    conv.append_message(conv.roles[0], inp)
    conv.append_message(conv.roles[1], None)
    prompt = conv.get_prompt()
    if chronological_prefix:
        prompt = prompt.replace("times:", "times in chronological order:")
    stop_str = conv.sep2 if conv.sep_style == SeparatorStyle.TWO else conv.sep
    return replace_video_token(prompt, image_paths, prompt_strategy), stop_str


def run_inference_single(model, processor, tokenizer, inp, image_paths, conv_mode="v1", timestamps=[],
                         prompt_strategy="interleave", chronological_prefix=True, temperature=0.2, max_new_tokens=256,
                         do_sample=True):
    if len(timestamps) > 0:
        order = sorted(range(len(image_paths)), key=lambda i: datetime.strptime(timestamps[i], "%Y-%m-%d"))
        image_paths = [image_paths[i] for i in order]
        timestamps = [timestamps[i] for i in order]
    frames = [processor.preprocess(p, return_tensors="pt")["pixel_values"][0] for p in image_paths]
    frames = [f.to(model.device, dtype=model.dtype) for f in frames]
    prompt, stop_str = build_prompt(inp, image_paths, conv_mode, prompt_strategy, chronological_prefix)
    input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).to(model.device)
    stopping = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
    with torch.inference_mode():
        output_ids = model.generate(input_ids=input_ids, images=frames, do_sample=do_sample, temperature=temperature,
                                    max_new_tokens=max_new_tokens, use_cache=True, stopping_criteria=[stopping])
    return tokenizer.decode(output_ids[0, input_ids.shape[1]:]).replace("</s>", "").strip()


_BBOX = re.compile(r"\[(\d+), (\d+), (\d+), (\d+)\]")
_POLYGON_DATASETS = ["xbd_loc", "xbd_dmg_cls", "s2_det", "qfabric_rqa2", "qfabric_rqa5", "xbd_sre_qa_rqa", "s2_sre_qa", "s2_rqa"]


def extract_bboxes(bbox_str):
    """'[x1, y1, x2, y2]' groups of integers (exactly ', '-separated) -> [[x1, y1, x2, y2], ...]  (inference.py:80-85)."""
    return [[int(v) for v in m.groups()] for m in _BBOX.finditer(bbox_str)]


def run_inference(dataset, model, tokenizer, processor, prompt_strategy, chronological_prefix, conv_mode, temperature,
                  max_new_tokens):
    """Dataset loop with the bookkeeping the metrics need (inference.py:88-137): response / ground truth / task per example,
    the example's polygon when it has one, and the integer boxes quoted in the question and in the reference answer."""
    outputs = []
    for example in dataset:
        question, answer = example["conversations"][0]["value"], example["conversations"][1]["value"]
        response = run_inference_single(model, processor, tokenizer, question, example["video"], conv_mode=conv_mode,
                                        timestamps=example["timestamp"], prompt_strategy=prompt_strategy,
                                        chronological_prefix=chronological_prefix, temperature=temperature,
                                        max_new_tokens=max_new_tokens)
        record = {"response": response, "ground_truth": answer, "task": example["task"]}
        polygon = example.get("polygon", None)
        if polygon is not None:
            record["polygon"] = polygon
        elif dataset in _POLYGON_DATASETS:            # as in the reference: only a dataset passed by NAME can trip this
            raise ValueError(f"Polygons not found for dataset {dataset}. The TEOChatlas dataset was updated to include these "
                             "polygons on 25 Mar 2025. Please re-download the json files for these splits.")
        boxes_in, boxes_out = extract_bboxes(question), extract_bboxes(answer)
        if boxes_in:
            record["input_bboxes"] = boxes_in
        if boxes_out:
            record["output_bboxes"] = boxes_out
        outputs.append(record)
    return outputs
