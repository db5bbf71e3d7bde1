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


def run_inference_batch(model, processor, tokenizer, inps, image_paths_list, conv_mode="v1", timestamps_list=None,
                        prompt_strategy="interleave", chronological_prefix=True, temperature=0.2, max_new_tokens=256,
                        do_sample=True):
    """run_inference_single for up to 16 examples at once (not in the reference, whose loop is one example at a time,
    inference.py:100-113): the same prompt construction, frame order, tokenisation and stop keyword per example, then ONE
    batched generation -- every frame of every example through the tower together, one prefill per example, and a decode loop
    that streams each weight matrix once per step for all examples (LlavaLlamaForCausalLM.generate_batch).  Greedy decoding
    gives the single-example answers; sampling draws from per-example Philox streams seeded from torch's global generator,
    so a sampled run is reproducible under torch.manual_seed but is not the single-example loop's stream."""
    B = len(inps)
    if timestamps_list is None:
        timestamps_list = [[] for _ in range(B)]
    if len(image_paths_list) != B or len(timestamps_list) != B:
        raise ValueError("run_inference_batch: one image list and one timestamp list per question")
    ids_list, frames_list, crits, n_prompt = [], [], [], []
    for inp, image_paths, timestamps in zip(inps, image_paths_list, timestamps_list):
        if len(timestamps) > 0:
            order = sorted(range(len(image_paths)), key=lambda i: datetime.strptime(timestamps[i], "%Y-%m-%d"))
            image_paths = [image_paths[i] for i in order]
        frames = [processor.preprocess(p, return_tensors="pt")["pixel_values"][0] for p in image_paths]
        frames_list.append([f.to(model.device, dtype=model.dtype) for f in frames])
        prompt, stop_str = build_prompt(inp, image_paths, conv_mode, prompt_strategy, chronological_prefix)
        input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).to(model.device)
        ids_list.append(input_ids[0])
        n_prompt.append(input_ids.shape[1])
        crits.append([KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)])
    with torch.inference_mode():
        outs = model.generate_batch(ids_list, frames_list, do_sample=do_sample, temperature=temperature,
                                    max_new_tokens=max_new_tokens, stopping_criteria=crits)
    return [tokenizer.decode(o[n:]).replace("</s>", "").strip() for o, n in zip(outs, n_prompt)]


_BBOX = re.compile(r"\[(\d+), (\d+), (\d+), (\d+)\]")
_POLYGON_DATASETS = ["xbd_loc", "xbd_dmg_cls", "s2_det", "qfabric_rqa2", "qfabric_rqa5", "xbd_sre_qa_rqa", "s2_sre_qa", "s2_rqa"]


def extract_bboxes(bbox_str):
    """'[x1, y1, x2, y2]' groups of integers (exactly ', '-separated) -> [[x1, y1, x2, y2], ...]  (inference.py:80-85)."""
    return [[int(v) for v in m.groups()] for m in _BBOX.finditer(bbox_str)]


def _record(example, response, dataset):
    question, answer = example["conversations"][0]["value"], example["conversations"][1]["value"]
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
    return record


def run_inference(dataset, model, tokenizer, processor, prompt_strategy, chronological_prefix, conv_mode, temperature,
                  max_new_tokens, batch_size=1):
    """Dataset loop with the bookkeeping the metrics need (inference.py:88-137): response / ground truth / task per example,
    the example's polygon when it has one, and the integer boxes quoted in the question and in the reference answer.
    batch_size (extra trailing keyword, default 1 = the reference's loop): answer that many consecutive examples per
    generation (run_inference_batch, <= 16); the records come out in dataset order either way."""
    if not 1 <= int(batch_size) <= 16:
        raise ValueError(f"batch_size {batch_size}: 1..16 examples per generation")
    outputs = []
    if batch_size == 1:
        for example in dataset:
            response = run_inference_single(model, processor, tokenizer, example["conversations"][0]["value"], example["video"],
                                            conv_mode=conv_mode, timestamps=example["timestamp"], prompt_strategy=prompt_strategy,
                                            chronological_prefix=chronological_prefix, temperature=temperature,
                                            max_new_tokens=max_new_tokens)
            outputs.append(_record(example, response, dataset))
        return outputs
    group = []

    def flush():
        if not group:
            return
        responses = run_inference_batch(model, processor, tokenizer, [e["conversations"][0]["value"] for e in group],
                                        [e["video"] for e in group], conv_mode=conv_mode,
                                        timestamps_list=[e["timestamp"] for e in group], prompt_strategy=prompt_strategy,
                                        chronological_prefix=chronological_prefix, temperature=temperature,
                                        max_new_tokens=max_new_tokens)
        outputs.extend(_record(e, r, dataset) for e, r in zip(group, responses))
        group.clear()

    for example in dataset:
        group.append(example)
        if len(group) == batch_size:
            flush()
    flush()
    return outputs
