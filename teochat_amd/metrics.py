"""Task metrics of the reference's dataset evaluation (host-side string work).

classification_metrics follows videollava/eval/classification.py:6-41: per-task accuracy of exact matches after
optional lower-casing / punctuation stripping; with `keywords`, a response also counts when it and the ground truth
share one of the keywords.  As in the reference, a task without a single hit is absent from the result.
The pixel-level detection metrics live in teochat_amd/detection.py.
"""
import string
from collections import Counter

_PUNCT_TABLE = str.maketrans("", "", string.punctuation)


def get_string_cleaner(ignore_casing, ignore_punctuation):
    def clean_string(text):
        if ignore_casing:
            text = text.lower()
        if ignore_punctuation:
            text = text.translate(_PUNCT_TABLE)
        return text
    return clean_string


def classification_metrics(outputs, ignore_casing=True, ignore_punctuation=True, keywords=None, **kwargs):
    clean = get_string_cleaner(ignore_casing, ignore_punctuation)
    hits, totals = Counter(), Counter()
    for rec in outputs:
        task = rec["task"]
        response, truth = clean(rec["response"]), clean(rec["ground_truth"])
        totals[task] += 1
        shared_keyword = keywords is not None and any(k in response and k in truth for k in keywords)
        if shared_keyword or response == truth:
            hits[task] += 1
    return {f"{task}_accuracy": n / totals[task] for task, n in hits.items()}
