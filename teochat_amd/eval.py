"""Dataset evaluation harness (mirror of the reference's videollava/eval/eval.py): `load_model` (:15-34) and the `eval` driver
(:37-171) -- same arguments, defaults, output-file naming and metric dispatch.  The TEOChatlas splits come from the Hugging
Face hub in the reference (`load_dataset("jirvin16/TEOChatlas", split=...)`); there is no network here, so `eval` also accepts
an in-memory `dataset=` (any iterable of examples) and otherwise makes the same `load_dataset` call."""
import json
from pathlib import Path

from .builder import load_pretrained_model
from .detection import detection_metrics
from .inference import run_inference
from .metrics import classification_metrics
from .mm_utils import get_model_name_from_path

CLASSIFICATION_DATASETS = ["fmow_high_res", "fmow_low_res", "abcd", "cdvqa", "aid", "ucm", "lrben", "hrben"]
DETECTION_DATASETS = ["xbd_loc", "xbd_dmg_cls", "s2_det", "xbd_sre_qa_rqa", "s2_sre_qa", "s2_rqa", "qfabric_rqa2",
                      "qfabric_rqa5_rtqa5", "qfabric_tre_rtqa"]
HF_SPLIT = {
    "fmow_high_res": "fMoW_High_Res", "fmow_low_res": "fMoW_Low_Res", "abcd": "ABCD", "cdvqa": "CDVQA", "aid": "AID",
    "ucm": "UCMerced", "lrben": "LRBEN", "hrben": "HRBEN",
    "xbd_loc": "xBD_Change_Detection_Localization", "xbd_dmg_cls": "xBD_Change_Detection_Classification",
    "s2_det": "S2Looking_Change_Detection", "xbd_sre_qa_rqa": "xBD_SRE_QA_RQA", "s2_sre_qa": "S2Looking_SRE_QA",
    "s2_rqa": "S2Looking_RQA", "qfabric_rqa2": "QFabric_RQA2", "qfabric_rqa5_rtqa5": "QFabric_RQA5_RTQA5",
    "qfabric_tre_rtqa": "QFabric_TRE_RTQA",
}


def load_model(model_path, model_base, load_8bit=False, load_4bit=False, cache_dir=None, device=None, **engine_kwargs):
    model_name = get_model_name_from_path(model_path)
    tokenizer, model, processor, _ = load_pretrained_model(model_path, model_base, model_name, load_4bit=load_4bit,
                                                           load_8bit=load_8bit, device=device, cache_dir=cache_dir,
                                                           **engine_kwargs)
    model.model.video_tower = None        # as the reference does: only the image tower is used at inference
    return tokenizer, model, processor["image"]


def output_path(dataset_name, model_path, out_name=None, out_dir=None, prompt_strategy=None, chronological_prefix=True):
    """results/<dataset>/<model>[_prompt_strategy_<s>][_chronological_prefix_<b>].json  (eval.py:101-126)."""
    out_dir = Path("results") if out_dir is None else Path(out_dir)
    out_dir.mkdir(exist_ok=True)
    sub = out_dir / dataset_name
    sub.mkdir(exist_ok=True)
    if out_name is None:
        out_name = f"{get_model_name_from_path(model_path)}.json"
    if ".json" not in out_name:
        out_name = f"{out_name}.json"
    for arg, val in (("prompt_strategy", prompt_strategy), ("chronological_prefix", chronological_prefix)):
        if val is not None:
            out_name = out_name.replace(".json", f"_{arg}_{val}.json")
    return sub / out_name


def eval(dataset_name, model_path, model_base, load_8bit=False, load_4bit=False, cache_dir=None, data_cache_dir=None,
         out_name=None, out_dir=None, prompt_strategy=None, chronological_prefix=True, conv_mode="v1", device="cuda",
         force_rerun=False, temperature=0.2, max_new_tokens=256, dataset=None, model_bundle=None, batch_size=1):
    """Run (or re-use) the model's answers on one TEOChatlas evaluation split and print / return its task metrics.
    Extra keywords (not in the reference): `dataset` = examples to use instead of the hub download, `model_bundle` =
    (tokenizer, model, processor) already loaded, `batch_size` = examples answered per generation (1 = the reference's loop;
    up to 16 share every weight read of a decode step, inference.run_inference_batch)."""
    print("Arguments passed to eval:")
    for k, v in (("dataset_name", dataset_name), ("model_path", model_path), ("model_base", model_base), ("out_name", out_name),
                 ("out_dir", out_dir), ("prompt_strategy", prompt_strategy), ("chronological_prefix", chronological_prefix),
                 ("conv_mode", conv_mode), ("device", device), ("force_rerun", force_rerun), ("temperature", temperature),
                 ("max_new_tokens", max_new_tokens)):
        print(f"\t{k} ({type(v).__name__}): {v}")
    if dataset_name in CLASSIFICATION_DATASETS:
        metrics_fn = classification_metrics
    elif dataset_name in DETECTION_DATASETS:
        metrics_fn = detection_metrics
    else:
        raise ValueError(f"Unsupported dataset: {dataset_name}")
    out_path = output_path(dataset_name, model_path, out_name, out_dir, prompt_strategy, chronological_prefix)
    if out_path.exists() and not force_rerun:
        print(f"Output file {out_path} already exists. Computing metrics without running inference.")
        with open(out_path, "r") as f:
            outputs = json.load(f)
    else:
        tokenizer, model, processor = model_bundle if model_bundle is not None else load_model(
            model_path, model_base, load_8bit=load_8bit, load_4bit=load_4bit, cache_dir=cache_dir, device=device)
        if dataset is None:
            from datasets import load_dataset
            dataset = load_dataset("jirvin16/TEOChatlas", split=f"eval_{HF_SPLIT[dataset_name]}", cache_dir=data_cache_dir,
                                   trust_remote_code=True)
        outputs = run_inference(dataset, model, tokenizer, processor, prompt_strategy, chronological_prefix, conv_mode,
                                temperature, max_new_tokens, **({"batch_size": batch_size} if batch_size != 1 else {}))
        print(f"Saving outputs to {out_path}")
        with open(out_path, "w") as f:
            json.dump(outputs, f, indent=4)
    metrics = metrics_fn(outputs, dataset_name=dataset_name)
    print(f"Metrics for dataset {dataset_name}:")
    for key, value in metrics.items():
        print(f"\t{key}: {value}")
    return metrics


def str_or_none(value):
    """CLI helper of the reference (eval/eval.py:172-175): "" / "none" (any case) -> None."""
    return None if value.strip().lower() in ("", "none") else value


# The command line of the reference's driver (eval/eval.py:178-199; launched by scripts/eval_teochat.sh as
# `python videollava/eval/eval.py --dataset_name ... --prompt_strategy interleave --chronological_prefix`): flag -> argparse
# options.  Names and defaults are interface; every parsed value is a keyword of eval().
_CLI = (
    ("dataset_name", dict(type=str, required=True)),
    ("model_path", dict(type=str, required=True)),
    ("model_base", dict(type=str_or_none, default=None)),
    ("load_8bit", dict(action="store_true")),
    ("load_4bit", dict(action="store_true")),
    ("cache_dir", dict(type=str, default=None)),
    ("data_cache_dir", dict(type=str, default=None)),
    ("out_name", dict(type=str, default=None)),
    ("out_dir", dict(type=str, default=None)),
    ("prompt_strategy", dict(type=str, default="interleave")),
    ("chronological_prefix", dict(action="store_true")),
    ("device", dict(type=str, default="cuda")),
    ("force_rerun", dict(action="store_true")),
    ("temperature", dict(type=float, default=0.2)),
    ("max_new_tokens", dict(type=int, default=256)),
    # not in the reference: examples answered per generation (1 = its one-at-a-time loop)
    ("batch_size", dict(type=int, default=1)),
)


def cli_parser():
    import argparse
    ap = argparse.ArgumentParser(description="TEOChat dataset evaluation on the MI355X path (same flags as the reference's eval.py)")
    for flag, opts in _CLI:
        ap.add_argument("--" + flag, **opts)
    return ap


def main(argv=None):
    return eval(**vars(cli_parser().parse_args(argv)))


if __name__ == "__main__":
    main()
