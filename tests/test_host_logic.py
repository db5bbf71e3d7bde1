"""CPU tests of the host-side mirror (teochat_amd) against the golden vectors the reference produced, and of the
C-ABI library's export table.  No GPU needed, no compute calls into the library."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from teochat_amd import _lib as L
from teochat_amd import constants, conversation, inference, mm_utils
from teochat_amd.model import build_splice_plan
from teochat_amd.tokenizer_stub import ByteTokenizer
from tests import _tiny as TY

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constants():
    assert constants.IMAGE_TOKEN_INDEX == -200 and constants.IGNORE_INDEX == -100
    assert constants.DEFAULT_IMAGE_TOKEN == "<image>" and constants.DEFAULT_VIDEO_TOKEN == "<video>"
    assert constants.MAX_IMAGE_LENGTH == 16


def test_run_inference_single_plumbing_matches_reference():
    """Same fake model/processor as the golden generator: the ids handed to generate() and the decoded text must be
    identical to what the reference's run_inference_single produced."""
    g = TY.load_json("host")
    reply = ByteTokenizer(add_bos=False)("Two new buildings.</s>").input_ids

    class FakeModel:
        device = torch.device("cpu")
        dtype = torch.float16

        def generate(self, input_ids=None, images=None, **kw):
            self.seen = (input_ids.clone(), len(images), kw)
            return torch.cat([input_ids, torch.tensor([reply])], dim=1)

    class FakeProc:
        def preprocess(self, p, return_tensors=None):
            return {"pixel_values": [torch.zeros(3, 2, 2)]}

    for c in g["run_inference_single"]:
        fm = FakeModel()
        paths = ["frame_%d.png" % i for i in range(c["T"])]
        text = inference.run_inference_single(fm, FakeProc(), ByteTokenizer(), c["inp"], paths, conv_mode="v1",
                                              timestamps=list(c["timestamps"]), prompt_strategy=c["strategy"],
                                              chronological_prefix=c["chrono"], temperature=0.2, max_new_tokens=16)
        assert fm.seen[0][0].tolist() == c["input_ids"], c
        assert fm.seen[1] == c["n_images"]
        assert text == c["output_text"]
        kw = fm.seen[2]
        for k, v in c["generate_kwargs"].items():
            assert kw[k] == v, (k, kw[k], v)
        assert len(kw["stopping_criteria"]) == 1


def test_tokenizer_image_token_edge_cases():
    g = TY.load_json("host")
    for c in g["tokenizer_image_token"]:
        ids = mm_utils.tokenizer_image_token(c["prompt"], ByteTokenizer(add_bos=c["add_bos"]), -200)
        assert ids == c["ids"], c
    t = mm_utils.tokenizer_image_token("a<image>b", ByteTokenizer(), -200, return_tensors="pt")
    assert t.dtype == torch.long
    with pytest.raises(ValueError):
        mm_utils.tokenizer_image_token("a", ByteTokenizer(), -200, return_tensors="np")


def test_replace_video_token_errors():
    with pytest.raises(ValueError):
        inference.replace_video_token("<video>", [1], "bogus")
    assert inference.replace_video_token("x <video> y", [1, 2], None) == "x <image><image> y"


def test_stopping_criteria_truth_table():
    g = TY.load_json("host")
    tok = ByteTokenizer()
    for c in g["stopping"]:
        crit = mm_utils.KeywordsStoppingCriteria(c["keywords"], tok, torch.zeros(1, c["prompt_len"], dtype=torch.long))
        assert bool(crit(torch.tensor([c["row"]]), None)) == c["stop"], c
    b = g["stopping_batch"]
    crit = mm_utils.KeywordsStoppingCriteria(b["keywords"], tok, torch.zeros(1, b["prompt_len"], dtype=torch.long))
    assert bool(crit(torch.tensor(b["rows"]), None)) == b["stop"]


def _decode_plan(plan_row, NV):
    out = []
    for p in plan_row:
        if p == L.INT32_MIN:
            out.append(0)                       # zero pad row: the coded embedding of a pad is 0
        elif p >= 0:
            out.append(int(p))
        else:
            g = -(int(p) + 1)
            out.append(-(1000 * (g // NV) + (g % NV) + 1))
    return out


def test_splice_plan_bit_exact():
    g = TY.load_json("splice")
    NV = g["NV"]
    for name, c in g.items():
        if not isinstance(c, dict) or "plan" not in c:
            continue
        ids = np.array(c["ids"], dtype=np.int64)
        mask = np.array(c["mask"], dtype=bool) if "mask" in c else np.ones_like(ids, dtype=bool)
        labels = np.array(c["labels"], dtype=np.int64) if "labels" in c else np.full_like(ids, -100)
        plan, lab, m, pos, _ = build_splice_plan(ids, mask, labels, [NV] * c["n_images"], c.get("max_len"),
                                                 c.get("padding_side", "right"))
        assert [_decode_plan(r, NV) for r in plan.tolist()] == c["plan"], name
        # zero rows in the reference output: the pad rows, plus rows gathered from vocab id 0 (coded value 0)
        zero_ref = np.array(c["embeds_is_zero_row"])
        assert ((plan == L.INT32_MIN) | (plan == 0)).tolist() == zero_ref.tolist(), name
        if c["position_ids"] is not None:
            assert pos.tolist() == c["position_ids"], name
        if c["attention_mask"] is not None:
            assert m.astype(np.int64).tolist() == c["attention_mask"], name
        if c["labels_out"] is not None:
            assert lab.tolist() == c["labels_out"], name
    with pytest.raises(IndexError):
        ids = np.array([[1, -200, -200]])
        build_splice_plan(ids, np.ones_like(ids, dtype=bool), np.full_like(ids, -100), [NV], None, "right")


def test_conversation_v1():
    c = conversation.conv_templates["v1"].copy()
    c.append_message(c.roles[0], "hi")
    c.append_message(c.roles[1], None)
    assert c.get_prompt().endswith("USER: hi ASSISTANT:")
    assert c.sep_style == conversation.SeparatorStyle.TWO and c.sep2 == "</s>"
    assert conversation.conv_templates["v1"].messages == []          # copy() did not alias the template


def test_processor_identity_at_224():
    from oracle import teo_oracle as O
    from teochat_amd.processor import TeoImageProcessor
    g = torch.Generator().manual_seed(3)
    raw = torch.randint(0, 256, (224, 224, 3), generator=g, dtype=torch.uint8)
    p = TeoImageProcessor()
    assert p.image_mean[0] == pytest.approx(0.48145466) and p.crop_size == {"height": 224, "width": 224}
    assert p.image_mean == O.OPENAI_DATASET_MEAN and p.image_std == O.OPENAI_DATASET_STD
    # the product has no host implementation of the transform: without an engine (= without the GPU) it refuses
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p.preprocess(raw.numpy(), return_tensors="pt")
    with pytest.raises(TypeError, match="uint8"):
        p.preprocess(torch.rand(3, 224, 224))
    with pytest.raises(ValueError):
        p(images=None)
    import teochat_amd.processor as P
    assert not hasattr(P.TeoImageProcessor, "transform") and "interpolate" not in open(P.__file__).read()


def test_library_exports_every_declared_symbol():
    """Every function include/teo_hip.h declares must be exported by the built .so and bound in _lib.py."""
    hdr = open(os.path.join(ROOT, "include", "teo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(teo_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"teo_graph"}
    assert declared, "no declarations parsed"
    assert os.path.exists(L.LIB_PATH), "libteo_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} not exported"
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    m = re.search(r"#define TEO_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "teo_hip.h")).read())
    assert L.load().teo_version() == int(m.group(1)) == L.ABI_VERSION          # header, library and binding agree (and load() checked sizeof)


def test_gate_up_interleave_layout():
    from teochat_amd.engine import interleave_gate_up, rope_tables
    F_, D = 48, 4
    gate = torch.arange(F_ * D, dtype=torch.float32).view(F_, D)
    up = -gate
    gu = interleave_gate_up(gate, up)
    for j in range(F_):
        r = (j // 16) * 32 + (j % 16)
        assert torch.equal(gu[r], gate[j]) and torch.equal(gu[r + 16], up[j])
    from oracle import teo_oracle as O
    cs, sn = rope_tables(16, 10000.0, 600)
    c2, s2 = O.rope_cos_sin(torch.arange(600), 16, 10000.0, torch.float32)
    # the same fp32 expressions; one ulp of slack: torch's vectorised cos / sin may pick a different lane path for the two shapes
    torch.testing.assert_close(cs, c2[:, :8], atol=1.2e-7, rtol=0)
    torch.testing.assert_close(sn, s2[:, :8], atol=1.2e-7, rtol=0)


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle (or the reference)."""
    pkg = os.path.join(ROOT, "teochat_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn
            assert "/root/reference" not in src, fn


def test_classification_metrics_match_reference_truth_table():
    """N4: teochat_amd.metrics.classification_metrics == videollava/eval/classification.py on tests/golden/metrics.json."""
    from teochat_amd.metrics import classification_metrics
    g = TY.load_json("metrics")
    for name, case in g["cases"].items():
        assert classification_metrics(g["outputs"], **case["kwargs"]) == case["result"], name


def test_eval_cli_has_the_reference_flags_and_defaults():
    """scripts/eval_teochat.sh launches `eval.py --dataset_name .. --model_path .. --prompt_strategy interleave
    --chronological_prefix ...` (eval/eval.py:178-199): same flag names, defaults and str_or_none semantics here."""
    import inspect
    from teochat_amd import eval as E
    a = vars(E.cli_parser().parse_args(["--dataset_name", "fmow_high_res", "--model_path", "ckpt"]))
    assert a == {"dataset_name": "fmow_high_res", "model_path": "ckpt", "model_base": None, "load_8bit": False, "load_4bit": False,
                 "cache_dir": None, "data_cache_dir": None, "out_name": None, "out_dir": None, "prompt_strategy": "interleave",
                 "chronological_prefix": False, "device": "cuda", "force_rerun": False, "temperature": 0.2, "max_new_tokens": 256,
                 "batch_size": 1}                     # batch_size: the one flag the reference does not have (default = its loop)
    b = vars(E.cli_parser().parse_args(["--dataset_name", "x", "--model_path", "y", "--model_base", "NONE", "--load_8bit",
                                         "--chronological_prefix", "--temperature", "0.5"]))
    assert b["model_base"] is None and b["load_8bit"] and b["chronological_prefix"] and b["temperature"] == 0.5
    assert E.str_or_none("") is None and E.str_or_none("base") == "base"
    params = inspect.signature(E.eval).parameters
    assert all(k in params for k in a)               # every flag is a keyword of eval(), as eval(**vars(args)) needs


def test_every_tune_key_is_documented_in_the_header():
    """include/teo_hip.h lists the perf-only knobs of a teo_tune block; every key the library accepts (teo_tune_keys(), generated from
    the one table in csrc/tune.h) must be named there and every key named there must exist (round-3 review: 'list every key')."""
    table = open(os.path.join(ROOT, "teochat_amd", "csrc", "tune.h")).read()
    keys = set(re.findall(r"^\s*X\((\w+),", table, flags=re.M))
    assert len(keys) > 20
    from teochat_amd import _lib as L
    assert set(L.tune_keys()) == keys
    hdr = open(os.path.join(ROOT, "include", "teo_hip.h")).read()
    block = hdr[hdr.index("Performance tuning knobs"):hdr.index("typedef struct teo_tune teo_tune;")]
    named = set(re.findall(r'"([a-z0-9_]+)"', block))
    assert keys - named == set(), f"accepted by the library but not documented: {sorted(keys - named)}"
    assert named - keys == set(), f"documented but not accepted by any source: {sorted(named - keys)}"


def test_no_process_wide_knob_is_left_in_the_library():
    """'no globals except an opaque teo_ctx*'.  No .hip file keeps a file-scope mutable int / bool (the knobs live in
    teo_tune blocks; what remains static is per-device caches and thread-local error / profiling state)."""
    import glob
    bad = []
    for f in glob.glob(os.path.join(ROOT, "teochat_amd", "csrc", "*.hip")):
        for n, line in enumerate(open(f).read().split("\n"), 1):
            if re.match(r"^static (int|bool|float|unsigned) g_\w+", line) or re.match(r"^(thread_local )?(int|bool) g_\w+ =", line):
                bad.append(f"{os.path.basename(f)}:{n}: {line.strip()}")
    assert bad == [], bad


def test_tune_blocks_are_independent_and_validated():
    """teo_tune blocks (no GPU needed: pure host state): defaults, per-block values, validation of keys and values, reset, and a
    second thread's bound block does not leak into this thread's."""
    import ctypes as C
    import threading
    from teochat_amd import _lib as L
    lib = L.load()
    a, b = L.Tune(), L.Tune()
    try:
        assert a.get("gemm_big") == 1 and a.get("gemm_big_cohort") == -1 and a.get("flash_pipe") == -1
        assert a.set("gemm_big", 2) == 0 and a.get("gemm_big") == 2 and b.get("gemm_big") == 1        # b untouched
        assert a.try_set("gemm_big", 7) != 0 and b"not a value" in lib.teo_last_error() and a.get("gemm_big") == 2
        assert a.try_set("no_such_key", 1) != 0 and b"unknown key" in lib.teo_last_error()
        assert a.try_set("gemm_sk_dbg", 1) != 0                   # the wrong-results diagnostic left the product library in round 5
        v = C.c_int(-5)
        assert lib.teo_tune_get(None, b"gemm_big", C.byref(v)) == 0 and v.value == 1                      # NULL block = shipped defaults
        assert a.reset() == 0 and a.get("gemm_big") == 1
        seen = {}

        def other():
            t = L.Tune()
            t.set("attn_chunk", 128)
            t.bind()
            seen["other"] = t.get("attn_chunk")
            lib.teo_tune_bind(None)
            t.close()
        th = threading.Thread(target=other)
        th.start(); th.join()
        assert seen["other"] == 128 and a.get("attn_chunk") == 0
    finally:
        a.close(); b.close()


def test_hand_scheduled_kernels_have_no_spills_and_no_scratch():
    """the K loops of gemm_quad.hip / gemm_narrow.hip count their `s_waitcnt vmcnt(N)` by hand (only the LDS-DMA pieces may be
    in flight; the asm-volatile MFMAs are invisible to the hazard recogniser).  A compiler that spilled a register or used scratch in those
    kernels would put loads / stores into the loop that the counts do not know about -- wrong data with no signal.  The compiler's own
    AMDHSA metadata (hipcc -S with the Makefile's flags, tools/kernel_meta.py) must say 0 spills and 0 scratch bytes for EVERY
    instantiation, on every build; the register budget the layouts were designed for is checked with it."""
    import shutil
    import pytest
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this machine")
    from tools.kernel_meta import kernel_meta
    for src, want in (("gemm_quad.hip", 24), ("gemm_narrow.hip", 12)):
        ks = kernel_meta(os.path.join(ROOT, "teochat_amd", "csrc", src))
        assert len(ks) >= want, (src, len(ks))
        for k in ks:
            assert k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
        if src == "gemm_quad.hip":
            # eight waves (two per SIMD): at most 256 registers per lane incl. the AGPR-pinned accumulators; four waves (one per SIMD): 512
            for k in ks:
                assert k["vgpr_count"] <= (256 if k["max_flat_workgroup_size"] == 512 else 512), k
                big = "<256," in k["name"] or "ILi256E" in k["name"]
                assert k["agpr_count"] >= (100 if big else 16), k     # the accumulators really live in AGPRs (256 x 160: 100 / 160; the small tiles 16 .. 64)
