"""N4: the dataset-evaluation side of the path -- detection / change-detection metrics, bbox
bookkeeping of run_inference and the eval() driver -- against tests/golden/detection.json, which the REFERENCE's own code
produced (tests/golden/make_golden.py::gen_detection runs videollava/eval/detection.py and eval/inference.py unmodified;
shapely, absent here, is replaced there by a WKT-text stand-in)."""
import json
import os

import numpy as np
import pytest

from teochat_amd import detection as D
from teochat_amd import inference as I
from tests import _tiny as TY

G = TY.load_json("detection")


@pytest.mark.parametrize("dataset", sorted(G["cases"]))
def test_detection_metrics_match_reference(dataset):
    case = G["cases"][dataset]
    got = D.detection_metrics(case["outputs"], dataset)
    assert set(got) == set(case["result"])
    for k, v in case["result"].items():
        assert float(got[k]) == pytest.approx(v, rel=1e-12, abs=1e-15), k


def test_mask_rasterisation_and_pixel_metrics_match_reference():
    polys = {"sq": G["cases"]["xbd_loc"]["outputs"][0]["polygon"], "tri": G["cases"]["xbd_loc"]["outputs"][1]["polygon"],
             "multi": G["cases"]["xbd_loc"]["outputs"][2]["polygon"]}
    for nm, w in polys.items():
        m = D.create_mask(w, (256, 256))
        ys, xs = np.nonzero(m)
        assert list(m.shape) == G["masks"][nm]["shape"] and m.dtype == np.uint8
        assert int(m.sum()) == G["masks"][nm]["sum"] and int((ys * 257 + xs).sum()) == G["masks"][nm]["weighted"], nm
    em = D.evaluate_masks(G["cases"]["xbd_loc"]["outputs"], "xbd_loc")
    assert list(em) == ["oa", "mIoU", "kappa", "fwIoU", "precision", "recall", "f1", "IoU"]
    for k, v in G["evaluate_masks_xbd_loc"].items():
        assert float(em[k]) == pytest.approx(v, rel=1e-12), k
    cdc = D.change_detection_classification(G["cases"]["xbd_dmg_cls"]["outputs"], ["no damage", "minor damage", "major damage", "destroyed"],
                                            skip_classes=["unclassified"])
    for k, v in G["change_detection_classification_xbd"].items():
        assert float(cdc[k]) == pytest.approx(v, rel=1e-12), k


def test_evaluator_matches_reference():
    e = G["evaluator3"]
    ev = D.Evaluator(3)
    rng = np.random.RandomState(0)
    gt = rng.randint(0, 3, size=(32, 32))
    pr = rng.randint(0, 3, size=(32, 32))
    ev.add_batch(gt, pr)
    assert ev.confusion_matrix.tolist() == e["cm"]
    assert float(ev.Pixel_Accuracy()) == pytest.approx(e["oa"], rel=1e-13)
    assert float(ev.Pixel_Accuracy_Class()[0]) == pytest.approx(e["macc"], rel=1e-13)
    assert float(ev.Mean_Intersection_over_Union()) == pytest.approx(e["miou"], rel=1e-13)
    assert float(ev.Kappa_coefficient()) == pytest.approx(e["kappa"], rel=1e-13)
    assert float(ev.Frequency_Weighted_Intersection_over_Union()) == pytest.approx(e["fwiou"], rel=1e-13)
    assert [float(v) for v in ev.Damage_F1_socore()] == pytest.approx(e["damage_f1"], rel=1e-13)
    assert float(ev.Class_Weighted_F1_score()) == pytest.approx(e["cw_f1"], rel=1e-13)
    with pytest.raises(AssertionError):
        ev.Pixel_F1_score()                                  # binary-only scores refuse a 3-class matrix, as the reference
    ev.reset()
    assert ev.confusion_matrix.sum() == 0
    assert D.get_classes("xbd", next(iter(D._CLASS_TABLE["xbd"]))) == G["get_classes"]["xbd"]
    assert D.get_classes("fmow", "x") is None


def test_wkt_parser_edge_cases():
    assert D.parse_wkt("POLYGON ((0 0, 0 1.5, 2e0 1, 0 0))") == [[(0.0, 0.0), (0.0, 1.5), (2.0, 1.0), (0.0, 0.0)]]
    assert D.parse_wkt("POLYGON((0 0,0 1,1 1,0 0),(0.1 0.1,0.1 0.2,0.2 0.2,0.1 0.1))") == [[(0.0, 0.0), (0.0, 1.0), (1.0, 1.0), (0.0, 0.0)]]
    mp = D.parse_wkt("MULTIPOLYGON (((5 5, 5 30, 30 30, 5 5)), ((200 200, 200 240, 250 240, 200 200), (210 210, 210 220, 220 220, 210 210)))")
    assert len(mp) == 2 and mp[1][0] == (200.0, 200.0) and len(mp[1]) == 4          # holes dropped
    assert D.parse_wkt("POLYGON Z ((0 0 9, 0 1 9, 1 1 9, 0 0 9))")[0][1] == (0.0, 1.0)
    assert D.parse_wkt("POLYGON EMPTY") == [] and D.parse_wkt([]) == []
    assert len(D.parse_wkt(["POLYGON ((0 0, 0 1, 1 1, 0 0))", "POLYGON ((2 2, 2 3, 3 3, 2 2))"])) == 2
    for bad in ("POINT (1 2)", "LINESTRING (0 0, 1 1)", "garbage"):
        with pytest.raises(ValueError):
            D.parse_wkt(bad)
    # responses: percent boxes -> pixel rectangles; unparsable groups skipped; too few numbers is an IndexError like the reference
    assert D.boxes_from_response("[0, 50, 40, 100] and [bad]", 256, 256) == [[(0.0, 128.0), (0.0, 256.0), (102.4, 256.0), (102.4, 128.0), (0.0, 128.0)]]
    with pytest.raises(IndexError):
        D.boxes_from_response("[1, 2, 3]", 256, 256)
    with pytest.raises(ValueError):
        D.detection_metrics([{"task": "x", "response": "", "ground_truth": ""}], "fmow_high_res")
    with pytest.raises(ValueError):
        D.detection_metrics([{"task": "unknown_task", "response": "", "ground_truth": ""}], "xbd_loc")


def test_run_inference_bookkeeping_matches_reference(monkeypatch):
    r = G["run_inference"]
    for s, want in r["extract"].items():
        assert I.extract_bboxes(s) == want, s
    calls = []

    def fake_single(model, processor, tokenizer, inp, image_paths, **kw):
        calls.append({"inp": inp, "image_paths": list(image_paths), "kw": dict(kw)})
        return f"answer {len(calls)}"

    monkeypatch.setattr(I, "run_inference_single", fake_single)
    outs = I.run_inference(r["examples"], "M", "T", "P", "interleave", True, "v1", 0.2, 64)
    assert outs == r["outputs"]
    assert [{"inp": c["inp"], "image_paths": c["image_paths"], "kw": c["kw"]} for c in calls] == r["calls"]
    # an example without a polygon, dataset passed by NAME as in the reference's check
    with pytest.raises(KeyError):
        I.run_inference([{"conversations": [{"value": "q"}, {"value": "a"}], "video": [], "timestamp": []}], None, None, None, None, True, "v1", 0.2, 8)


def test_run_inference_batched_loop_groups_in_order(monkeypatch):
    """batch_size > 1: consecutive examples are answered together (run_inference_batch), records stay in dataset order with
    the same bookkeeping; the last group may be short; out-of-range sizes are refused."""
    r = G["run_inference"]
    groups = []

    def fake_batch(model, processor, tokenizer, inps, image_paths_list, **kw):
        groups.append({"inps": list(inps), "paths": [list(p) for p in image_paths_list], "ts": [list(t) for t in kw["timestamps_list"]],
                       "kw": {k: v for k, v in kw.items() if k != "timestamps_list"}})
        base = sum(len(g["inps"]) for g in groups[:-1])
        return [f"answer {base + i + 1}" for i in range(len(inps))]

    monkeypatch.setattr(I, "run_inference_batch", fake_batch)
    n = len(r["examples"])
    for bs in (2, 16):
        groups.clear()
        outs = I.run_inference(r["examples"], "M", "T", "P", "interleave", True, "v1", 0.2, 64, batch_size=bs)
        assert outs == r["outputs"]                       # same records as the one-at-a-time loop (fake answers are numbered alike)
        assert [len(g["inps"]) for g in groups] == [bs] * (n // bs) + ([n % bs] if n % bs else [])
        flat = [(q, p, t) for g in groups for q, p, t in zip(g["inps"], g["paths"], g["ts"])]
        assert flat == [(c["inp"], c["image_paths"], c["kw"]["timestamps"]) for c in r["calls"]]
        assert all(g["kw"] == {k: v for k, v in r["calls"][0]["kw"].items() if k != "timestamps"} for g in groups)
    for bad in (0, 17):
        with pytest.raises(ValueError, match="batch_size"):
            I.run_inference(r["examples"], "M", "T", "P", "interleave", True, "v1", 0.2, 64, batch_size=bad)


def test_eval_driver_reuses_saved_outputs_and_dispatches_metrics(tmp_path, capsys):
    from teochat_amd import eval as E
    recs = G["cases"]["xbd_dmg_cls"]["outputs"]
    p = E.output_path("xbd_dmg_cls", "/models/teochat-7b/checkpoint-100", None, tmp_path, prompt_strategy="interleave", chronological_prefix=True)
    assert p == tmp_path / "xbd_dmg_cls" / "teochat-7b_checkpoint-100_prompt_strategy_interleave_chronological_prefix_True.json"
    assert E.output_path("abcd", "m", "run1", tmp_path, None, None).name == "run1.json"
    json.dump(recs, open(p, "w"))
    m = E.eval("xbd_dmg_cls", "/models/teochat-7b/checkpoint-100", None, out_dir=tmp_path, prompt_strategy="interleave")
    assert m == pytest.approx(G["cases"]["xbd_dmg_cls"]["result"])
    assert "already exists. Computing metrics without running inference." in capsys.readouterr().out
    with pytest.raises(ValueError, match="Unsupported dataset"):
        E.eval("imagenet", "m", None, out_dir=tmp_path)
    # the inference branch with an injected bundle and dataset: outputs are written where the reference writes them
    seen = []

    def fake_single(model, processor, tokenizer, inp, image_paths, **kw):
        seen.append(kw)
        return "yes"

    import teochat_amd.inference as I2
    orig = I2.run_inference_single
    I2.run_inference_single = fake_single
    try:
        ds = [{"conversations": [{"value": "<video> q?"}, {"value": "Yes"}], "video": ["a"], "timestamp": [], "task": "qa"}]
        m2 = E.eval("cdvqa", "m", None, out_dir=tmp_path, out_name="fresh", dataset=ds, model_bundle=("tok", "model", "proc"), temperature=0.7,
                    max_new_tokens=5)
    finally:
        I2.run_inference_single = orig
    assert m2 == {"qa_accuracy": 1.0} and seen[0]["temperature"] == 0.7 and seen[0]["max_new_tokens"] == 5
    saved = json.load(open(tmp_path / "cdvqa" / "fresh_chronological_prefix_True.json"))
    assert saved == [{"response": "yes", "ground_truth": "Yes", "task": "qa"}]


def test_dropin_exposes_the_eval_modules():
    import teochat_amd.dropin as dropin
    dropin.install()
    from videollava.eval.detection import detection_metrics, Evaluator       # noqa: F401
    from videollava.eval.eval import eval as ref_eval, load_model           # noqa: F401
    from videollava.eval.inference import extract_bboxes, run_inference     # noqa: F401
    assert detection_metrics is D.detection_metrics
