"""Detection / change-detection task metrics of the reference's dataset evaluation.

Mirrors videollava/eval/detection.py: `Evaluator` (:12-113, confusion-matrix metrics), `get_classes` (:116-134), `create_mask`
(:137-158), `evaluate_masks` (:161-222), `change_detection_classification` (:225-301), `detection_metrics` (:304-412) -- same
names, arguments, result keys and error behaviour.  Host-side numpy / PIL work on the model's text responses; nothing here is
on the GPU hot path.

The reference parses its polygons with shapely (absent from this image and only used for WKT parsing there: `wkt.loads`,
`.exterior.coords`); `parse_wkt` below reads the POLYGON / MULTIPOLYGON text itself.  Rasterisation is the very PIL call the
reference makes (`ImageDraw.polygon(exterior, outline=1, fill=1)`), so masks are identical pixel for pixel.
"""
import re
from collections import defaultdict

import numpy as np
from PIL import Image, ImageDraw

from .metrics import classification_metrics, get_string_cleaner

_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)"
_RING = re.compile(r"\(\s*(" + _NUM + r"\s+" + _NUM + r"(?:\s+" + _NUM + r")?(?:\s*,\s*" + _NUM + r"\s+" + _NUM + r"(?:\s+" + _NUM + r")?)*)\s*\)")


def _ring_coords(text):
    pts = []
    for pair in text.split(","):
        vals = pair.split()
        pts.append((float(vals[0]), float(vals[1])))
    return pts


def parse_wkt(text):
    """'POLYGON ((x y, ...), (hole ...))' or 'MULTIPOLYGON (((...)), ((...)))' -> list of exterior rings [[(x, y), ...], ...]
    (interior rings are dropped: the reference only draws `polygon.exterior`).  A list / tuple / array of WKT strings gives the
    concatenation (shapely.wkt.loads on a sequence, detection.py:206).  EMPTY geometries give no rings."""
    if not isinstance(text, str):
        rings = []
        for t in text:
            rings.extend(parse_wkt(t))
        return rings
    s = text.strip()
    head = s.split("(", 1)[0].strip().upper()
    if head.endswith("EMPTY") or "(" not in s:
        if head.replace("EMPTY", "").strip() in ("POLYGON", "MULTIPOLYGON"):
            return []
        raise ValueError(f"unsupported WKT: {text[:60]!r}")
    kind = head.split()[0]
    body = s[s.index("("):]
    if kind == "POLYGON":
        m = _RING.search(body)
        if not m:
            raise ValueError(f"malformed POLYGON: {text[:60]!r}")
        return [_ring_coords(m.group(1))]
    if kind == "MULTIPOLYGON":
        rings = []
        depth, start = 0, None
        for i, ch in enumerate(body):                      # polygons are the depth-2 parenthesised groups
            if ch == "(":
                depth += 1
                if depth == 2:
                    start = i
            elif ch == ")":
                if depth == 2 and start is not None:
                    m = _RING.search(body[start:i + 1])
                    if m:
                        rings.append(_ring_coords(m.group(1)))
                    start = None
                depth -= 1
        return rings
    raise ValueError(f"unsupported WKT geometry {kind!r} (POLYGON / MULTIPOLYGON expected)")


class Evaluator(object):
    """Pixel-level confusion matrix and the scores derived from it (rows = ground truth, columns = prediction)."""

    def __init__(self, num_class):
        self.num_class = num_class
        self.confusion_matrix = np.zeros((num_class, num_class), dtype=np.longlong)

    # --- scores; names as in the reference (callers look them up by name)
    def Pixel_Accuracy(self):
        cm = self.confusion_matrix
        return np.diag(cm).sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        cm = self.confusion_matrix
        per_class = np.diag(cm) / (cm.sum(axis=1) + 1e-7)
        return np.nanmean(per_class), per_class

    def _binary(self):
        cm = self.confusion_matrix
        assert cm.shape[0] == 2
        return cm[1, 1], cm[0, 1], cm[1, 0]                 # tp, fp, fn

    def Pixel_Precision_Rate(self):
        tp, fp, _ = self._binary()
        return tp / (fp + tp)

    def Pixel_Recall_Rate(self):
        tp, _, fn = self._binary()
        return tp / (fn + tp)

    def Pixel_F1_score(self):
        assert self.confusion_matrix.shape[0] == 2
        rec, pre = self.Pixel_Recall_Rate(), self.Pixel_Precision_Rate()
        return 2 * rec * pre / (rec + pre)

    def calculate_per_class_metrics(self):
        cm = self.confusion_matrix
        tps = np.diag(cm)[1:]                               # class 0 (background) is excluded
        return tps, np.sum(cm, axis=1)[1:] - tps, np.sum(cm, axis=0)[1:] - tps

    def _per_class_f1(self):
        tps, fns, fps = self.calculate_per_class_metrics()
        precisions = tps / (tps + fps + 1e-7)
        recalls = tps / (tps + fns + 1e-7)
        return 2 * (precisions * recalls) / (precisions + recalls + 1e-7)

    def Damage_F1_socore(self):                             # sic: the reference's spelling is the interface
        return self._per_class_f1()

    def Mean_Intersection_over_Union(self):
        cm = self.confusion_matrix
        iou = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm) + 1e-7)
        return np.nanmean(iou)

    def Intersection_over_Union(self):
        tp, fp, fn = self._binary()
        return tp / (fp + fn + tp)

    def Kappa_coefficient(self):
        cm = self.confusion_matrix
        n = np.sum(cm)
        observed = np.trace(cm) / n
        expected = np.sum(np.sum(cm, axis=0) / n * np.sum(cm, axis=1) / n)
        return (observed - expected) / (1 - expected)

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        freq = np.sum(cm, axis=1) / np.sum(cm)
        iu = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        return (freq[freq > 0] * iu[freq > 0]).sum()

    def Class_Weighted_F1_score(self):
        f1 = self._per_class_f1()
        w = 1 / np.sum(self.confusion_matrix, axis=1)[1:]
        w = w / np.sum(w)
        return np.sum(w * f1)

    # --- accumulation
    def _generate_matrix(self, gt_image, pre_image):
        valid = (gt_image >= 0) & (gt_image < self.num_class)
        pairs = self.num_class * gt_image[valid].astype("int64") + pre_image[valid]
        return np.bincount(pairs, minlength=self.num_class ** 2).reshape(self.num_class, self.num_class)

    def add_batch(self, gt_image, pre_image):
        assert gt_image.shape == pre_image.shape
        self.confusion_matrix += self._generate_matrix(gt_image, pre_image)

    def reset(self):
        self.confusion_matrix = np.zeros((self.num_class,) * 2)


_QFABRIC_STATUS = ["prior-construction", "greenland ", "land-cleared", "excavation", "materials-dumped", "construction-started",
                   "construction-midway", "construction-done", "operational"]
_QFABRIC_TYPES = ["residential", "commercial", "industrial", "road", "demolition", "mega-projects"]
_CLASS_TABLE = {
    "qfabric": {
        "temporal_region_based_question_answering: What is the development status in this region [bbox] in image N?": _QFABRIC_STATUS,
        "region_based_question_answering: Identify the type of urban development that has occurred in this area [bbox].": _QFABRIC_TYPES,
    },
    "xbd": {
        "classification: Classify the level of damage experienced by the building at location [bbox] in the second image. "
        "Choose from: No damage, Minor Damage, Major Damage, Destroyed.": ["No damage", "Minor damage", "Major damage", "Destroyed"],
    },
}


def get_classes(dataset, task):
    return _CLASS_TABLE.get(dataset, {}).get(task)


def create_mask(polygons, im_size):
    """Exterior rings -> uint8 mask (1 inside / on the outline), PIL's polygon fill as in detection.py:137-158.
    `polygons`: WKT text (or a sequence of WKT texts), or already parsed rings."""
    rings = parse_wkt(polygons) if (isinstance(polygons, str) or (len(polygons) and isinstance(polygons[0], str))) else polygons
    img = Image.new("L", im_size, 0)
    draw = ImageDraw.Draw(img)
    for ring in rings:
        draw.polygon([tuple(p) for p in ring], outline=1, fill=1)
    return np.array(img)


_BOX = re.compile(r"\[(.*?)\]")


def boxes_from_response(text, width, height):
    """'[x1, y1, x2, y2], ...' with coordinates in percent of the image -> pixel-space rectangle rings (detection.py:189-205);
    bracket groups that do not parse as numbers are skipped."""
    rings = []
    for grp in _BOX.findall(text):
        try:
            b = list(map(float, grp.split(",")))
        except ValueError:
            continue
        x1, y1, x2, y2 = b[0] / 100 * width, b[1] / 100 * height, b[2] / 100 * width, b[3] / 100 * height
        # the reference formats the corners into WKT text with f-strings and parses them back: repr round trip of a float
        x1, y1, x2, y2 = (float(f"{v}") for v in (x1, y1, x2, y2))
        rings.append([(x1, y1), (x1, y2), (x2, y2), (x2, y1), (x1, y1)])
    return rings


def evaluate_masks(results, dataset, height=256, width=256):
    ev = Evaluator(num_class=2)
    for rec in results:
        if "[" not in rec["ground_truth"]:
            gt = np.zeros((height, width), dtype="uint8")
        else:
            gt = create_mask(parse_wkt(rec["polygon"]), (height, width))
        if "[" not in rec["response"]:
            pred = np.zeros((height, width), dtype="uint8")
        else:
            pred = create_mask(boxes_from_response(rec["response"], width, height), (height, width))
        ev.add_batch(gt, pred)
    return {"oa": ev.Pixel_Accuracy(), "mIoU": ev.Mean_Intersection_over_Union(), "kappa": ev.Kappa_coefficient(),
            "fwIoU": ev.Frequency_Weighted_Intersection_over_Union(), "precision": ev.Pixel_Precision_Rate(),
            "recall": ev.Pixel_Recall_Rate(), "f1": ev.Pixel_F1_score(), "IoU": ev.Intersection_over_Union()}


def change_detection_classification(outputs, classes, skip_classes=[], height=256, width=256, ignore_casing=True,
                                    ignore_punctuation=True):
    """Per-pixel F1 of the predicted vs true class painted on each example's polygon: plain mean over classes ('f1'),
    prevalence-weighted ('w_f1') and inverse-prevalence-weighted ('inv_w_f1')."""
    stats = defaultdict(lambda: {"tp": 0, "fp": 0, "fn": 0, "count": 0})
    clean = get_string_cleaner(ignore_casing, ignore_punctuation)
    for rec in outputs:
        pred_cls, true_cls = clean(rec["response"]), clean(rec["ground_truth"])
        region = create_mask(parse_wkt(rec["polygon"]), im_size=(height, width))
        if true_cls in skip_classes:
            continue
        if pred_cls not in classes:
            fn = 0                                          # the reference sums an all-zero mask here
        else:
            pred_label, true_label = classes.index(pred_cls) + 1, classes.index(true_cls) + 1
            inside = region > 0
            pred_img = np.where(inside, pred_label, 0)
            true_img = np.where(inside, true_label, 0)
            tp = (pred_img == true_label).sum()
            fp = (pred_img == pred_label).sum() - tp
            fn = (true_img == true_label).sum() - tp
            stats[pred_cls]["tp"] += tp
            stats[pred_cls]["fp"] += fp
        stats[true_cls]["fn"] += fn
        stats[true_cls]["count"] += np.sum(region)
    total = sum(s["count"] for s in stats.values())
    per_class = {}
    weighted = inv_weighted = inv_total = 0
    for name in classes:
        tp, fp, fn = stats[name]["tp"], stats[name]["fp"], stats[name]["fn"]
        precision = 0.0 if tp + fp == 0 else tp / (tp + fp)
        recall = 0.0 if tp + fn == 0 else tp / (tp + fn)
        f1 = 0.0 if precision + recall == 0 else 2 * (precision * recall) / (precision + recall)
        per_class[name] = f1
        prevalence = stats[name]["count"] / total
        weighted += f1 * prevalence
        if prevalence != 0:
            inv_weighted += f1 / prevalence
            inv_total += 1 / prevalence
    inv_weighted = inv_weighted / inv_total if inv_total > 0 else 0.0
    return {"f1": np.mean(list(per_class.values())), "w_f1": weighted, "inv_w_f1": inv_weighted}


_XBD_QA_KEYWORDS = ["yes", "no", "top left", "top center", "top right", "center left", "center", "center right", "bottom left",
                    "bottom center", "bottom right"]
_QFABRIC_TYPES_CLEAN = ["residential", "commercial", "industrial", "road", "demolition", "mega projects"]
_QFABRIC_STATUS_CLEAN = ["prior construction", "greenland", "land cleared", "excavation", "materials dumped", "construction started",
                         "construction midway", "construction done", "operational"]


def detection_metrics(outputs, dataset_name, ignore_casing=True, ignore_punctuation=True):
    by_task = defaultdict(list)
    for rec in outputs:
        by_task[rec["task"]].append(rec)
    kw = dict(ignore_casing=ignore_casing, ignore_punctuation=ignore_punctuation)

    def accuracy(task, **extra):
        return classification_metrics(by_task[task], **kw, **extra)[f"{task}_accuracy"]

    def masks_f1(task):
        return evaluate_masks(by_task[task], dataset_name)["f1"]

    res = {}
    for task in by_task:
        if "xbd" in dataset_name:
            if task == "change_detection_classification":
                assert dataset_name == "xbd_dmg_cls"
                res[f"{task}_f1"] = change_detection_classification(
                    by_task[task], ["no damage", "minor damage", "major damage", "destroyed"], skip_classes=["unclassified"], **kw)["inv_w_f1"]
            elif task == "change_detection_localization":
                res[f"{task}_f1"] = masks_f1(task)
            elif task == "spatial_referring_expression":
                assert dataset_name == "xbd_sre_qa_rqa"
                res[f"{task}_f1"] = masks_f1(task)
            elif task == "region_based_question_answering":
                assert dataset_name == "xbd_sre_qa_rqa"
                res[f"{task}_accuracy"] = accuracy(task)
            elif task == "question_answering":
                assert dataset_name == "xbd_sre_qa_rqa"
                res[f"{task}_accuracy"] = accuracy(task, keywords=_XBD_QA_KEYWORDS)
            else:
                raise ValueError(f"Unsupported task {task} for dataset {dataset_name}")
        elif "s2" in dataset_name:
            if task == "change_detection_detection" and dataset_name == "s2_det":
                res[f"{task}_f1"] = masks_f1(task)
            elif task == "region_based_question_answering":
                assert dataset_name == "s2_rqa"
                res[f"{task}_accuracy"] = accuracy(task)
            elif task == "spatial_referring_expression":
                assert dataset_name == "s2_sre_qa"
                res[f"{task}_f1"] = masks_f1(task)
            elif task == "question_answering":
                assert dataset_name == "s2_sre_qa"
                res[f"{task}_accuracy"] = accuracy(task)
            else:
                raise ValueError(f"Unsupported task {task} for dataset {dataset_name}")
        elif "qfabric" in dataset_name:
            if task == "region_based_question_answering":
                res[f"{task}_f1"] = change_detection_classification(by_task[task], _QFABRIC_TYPES_CLEAN, skip_classes=[], **kw)["w_f1"]
            elif task == "region_based_temporal_question_answering":
                if dataset_name == "qfabric_tre_rtqa":
                    res[f"{task}_accuracy"] = accuracy(task)
                elif dataset_name == "qfabric_rqa5_rtqa5":
                    res[f"{task}_f1"] = change_detection_classification(by_task[task], _QFABRIC_STATUS_CLEAN, skip_classes=[], **kw)["w_f1"]
                else:
                    raise ValueError(f"Unsupported dataset {dataset_name} for task {task}")
            elif task == "temporal_referring_expression":
                assert dataset_name == "qfabric_tre_rtqa"
                res[f"{task}_accuracy"] = accuracy(task)
            else:
                raise ValueError(f"Unsupported task: {task} for dataset {dataset_name}")
        else:
            raise ValueError(f"Unsupported dataset: {dataset_name}")
    return res
