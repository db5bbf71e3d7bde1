"""Pin the CPU oracle (oracle/teo_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import teo_oracle as O
from teochat_amd.tokenizer_stub import ByteTokenizer
from tests import _tiny as TY

FP32_TOL = 2e-5     # reference (tf-5.15 op order) vs restatement, fp32, tiny configs


def test_prompts_and_token_packing_bit_exact():
    g = TY.load_json("host")
    for c in g["run_inference_single"]:
        T = c["T"]
        prompt = O.build_prompt(c["inp"], T, c["strategy"], c["chrono"])
        ids = O.tokenizer_image_token(prompt, ByteTokenizer(), O.IMAGE_TOKEN_INDEX)
        assert ids == c["input_ids"], c
        assert prompt.split("<image>") == c["prompt_chunks"], c
        assert ids.count(-200) == (T if "<video>" in c["inp"] else 0)
    for c in g["tokenizer_image_token"]:
        ids = O.tokenizer_image_token(c["prompt"], ByteTokenizer(add_bos=c["add_bos"]), -200)
        assert ids == c["ids"], c


def test_stopping_truth_table():
    g = TY.load_json("host")
    tok = ByteTokenizer()
    for c in g["stopping"]:
        crit = O.KeywordsStop(c["keywords"], tok, c["prompt_len"])
        assert crit([c["row"]]) == c["stop"], c
    b = g["stopping_batch"]
    assert O.KeywordsStop(b["keywords"], tok, b["prompt_len"])(b["rows"]) == b["stop"]


def _coded_tables(V, D, n_images, NV):
    emb = torch.arange(V, dtype=torch.float32).view(V, 1).expand(V, D).contiguous()
    feats = []
    for i in range(n_images):
        f = torch.zeros(NV, D)
        for j in range(NV):
            f[j] = -(1000 * i + j + 1)
        feats.append(f)
    return emb, feats


def test_splice_plans_bit_exact():
    g = TY.load_json("splice")
    NV = g["NV"]
    _, lcfg, _ = TY.cfgs("tinyA")
    for name, c in g.items():
        if not isinstance(c, dict) or "plan" not in c:
            continue
        mm = O.MMCfg(tokenizer_model_max_length=c.get("max_len"), tokenizer_padding_side=c.get("padding_side", "right"))
        emb, feats = _coded_tables(lcfg.vocab_size, 8, c["n_images"], NV)
        ids = torch.tensor(c["ids"], dtype=torch.long)
        mask = torch.tensor(c["mask"], dtype=torch.long) if "mask" in c else None
        labels = torch.tensor(c["labels"], dtype=torch.long) if "labels" in c else None
        pos = torch.arange(ids.shape[1]).unsqueeze(0).expand(ids.shape[0], -1).clone() if c.get("pos") else None
        r = O.prepare_inputs_labels_for_multimodal(ids, pos, mask, None, labels, feats, emb, mm)
        assert r[0] is None
        assert r[4][:, :, 0].round().long().tolist() == c["plan"], name
        assert (r[4].abs().sum(-1) == 0).tolist() == c["embeds_is_zero_row"], name
        assert (None if r[1] is None else r[1].tolist()) == c["position_ids"], name
        assert (None if r[2] is None else r[2].long().tolist()) == c["attention_mask"], name
        assert (None if r[5] is None else r[5].tolist()) == c["labels_out"], name
    # error convention
    emb, feats = _coded_tables(lcfg.vocab_size, 8, 1, NV)
    with pytest.raises(IndexError):
        O.prepare_inputs_labels_for_multimodal(torch.tensor([[1, -200, -200]]), None, None, None, None, feats, emb,
                                               O.MMCfg())
    assert g["too_few_images_error"] == "IndexError"
    d = g["decode_branch"]
    r = O.prepare_inputs_labels_for_multimodal(torch.tensor(d["input_ids"]), None, torch.tensor(d["mask_in"]),
                                               ((None, None),), None, feats, emb, O.MMCfg(), past_len=d["past_len"])
    assert r[1].tolist() == d["position_ids"] and r[2].tolist() == d["attention_mask"] and r[4] is None
    p = g["no_images_passthrough"]
    r = O.prepare_inputs_labels_for_multimodal(torch.tensor(p["input_ids"]), None, None, None, None, None, emb, O.MMCfg())
    assert r[0].tolist() == p["input_ids"] and all(x is None for x in r[1:])


@pytest.mark.parametrize("name", ["tinyA", "tinyB", "tinyC"])
def test_numeric_fixtures(name):
    g = TY.load_npz(name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    assert abs(TY.sd_checksum(sd) - float(g["sd_checksum"])) < 1e-6 * float(g["sd_checksum"])
    T = int(g["T"])
    frames = O.synthetic_frames(T, vcfg.image_size, seed=0)
    assert abs(float(sum(f.double().abs().sum() for f in frames)) - float(g["frames_checksum"])) < 1e-3
    ids = torch.from_numpy(g["input_ids"])
    assert torch.equal(ids[0], TY.prompt_ids(name, ids.shape[1], T, lcfg.vocab_size, seed=1))
    if name == "tinyC":
        # the anchored config exists for its token stream: the REFERENCE's greedy stream (the fixture) is varied -- at least 6
        # distinct tokens in 8 steps, with a context-decided jump off the anchors' +1 walk
        gt = g["greedy_tokens"].tolist()
        assert len(set(gt)) >= 6 and any(b != a + 1 for a, b in zip(gt, gt[1:]))
    pix = torch.stack(frames)

    # G4 ViT
    states = O.vit_hidden_states(pix, sd, vcfg)
    assert len(states) == int(g["vit_n_states"])
    np.testing.assert_allclose(states[0][:, :4].numpy(), g["vit_hidden0_row0"], atol=FP32_TOL, rtol=0)
    feats = O.vit_features(pix, sd, vcfg, -2, "patch")
    np.testing.assert_allclose(feats.numpy(), g["vit_features"], atol=FP32_TOL, rtol=1e-5)
    # G5 projector
    proj = O.projector(feats, sd, mm.mm_projector_type)
    np.testing.assert_allclose(proj[:, ::8].numpy(), g["projector_rows"], atol=FP32_TOL, rtol=1e-5)

    # G7 end-to-end prefill
    logits, cache, embeds = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm)
    L = int(g["e2e_L"])
    assert logits.shape[1] == L == ids.shape[1] - T + 256 * T
    sel = torch.from_numpy(g["e2e_sel"])
    np.testing.assert_allclose(logits[0][sel].numpy(), g["e2e_logits_sel"], atol=FP32_TOL, rtol=1e-5)
    assert abs(float(logits.double().abs().sum()) - float(g["e2e_logits_sum_abs"])) < 1e-5 * float(g["e2e_logits_sum_abs"])
    am = logits[0].argmax(-1).numpy()
    assert (am == g["e2e_argmax_all"]).mean() > 0.995   # ties aside

    # G6 greedy decode + KV snapshots
    toks, step_logits, cache = O.greedy_generate(ids, frames, sd, vcfg, lcfg, mm, max_new_tokens=len(g["greedy_tokens"]))
    assert toks == g["greedy_tokens"].tolist()
    np.testing.assert_allclose(step_logits.numpy(), g["greedy_logits"], atol=FP32_TOL, rtol=1e-5)
    ks = torch.from_numpy(g["kv_sel"])
    # after n_new tokens the reference cache holds L + n_new - 1 entries (the last token is never fed back)
    assert cache.length == int(g["kv_len"])
    np.testing.assert_allclose(cache.k[0][0][:, ks].numpy(), g["k_layer0"], atol=FP32_TOL, rtol=1e-5)
    np.testing.assert_allclose(cache.v[0][0][:, ks].numpy(), g["v_layer0"], atol=FP32_TOL, rtol=1e-5)
    np.testing.assert_allclose(cache.k[-1][0][:, ks].numpy(), g["k_last"], atol=FP32_TOL, rtol=1e-5)
    np.testing.assert_allclose(cache.v[-1][0][:, ks].numpy(), g["v_last"], atol=FP32_TOL, rtol=1e-5)

    # text-only forward (no images)
    tids = torch.from_numpy(g["text_only_ids"])
    emb = sd["model.embed_tokens.weight"][tids]
    lg, _ = O.llama_forward(emb, None, None, None, sd, lcfg)
    np.testing.assert_allclose(lg[0].numpy(), g["text_only_logits"], atol=FP32_TOL, rtol=1e-5)


@pytest.mark.parametrize("name", ["tinyB"])
def test_bf16_boundary_mode_is_close_to_reference_bf16(name):
    """Information-level check: the oracle's bf16-boundary mode (what the HIP bf16 path stores) stays as
    close to the fp32 truth as the reference's own bf16 CPU run does (different rounding points)."""
    g = TY.load_npz(name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    sd16 = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    frames = O.synthetic_frames(int(g["T"]), vcfg.image_size, seed=0)
    ids = torch.from_numpy(g["input_ids"])
    sel = torch.from_numpy(g["e2e_sel"])
    lo, _, _ = O.mm_forward(ids, frames, sd16, vcfg, lcfg, mm, rounding="bf16")
    ours = lo[0][sel].numpy()
    truth = g["e2e_logits_sel"]
    ref16 = g["e2e_bf16_logits_sel"]
    scale = np.abs(truth).max()
    err_ours = np.abs(ours - truth).max() / scale
    err_ref = np.abs(ref16 - truth).max() / scale
    assert err_ours < 0.05 and err_ours < 2.0 * err_ref + 1e-2, (err_ours, err_ref)


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_oracle_hidden_states_match_reference(name):
    """Round 5: `output_hidden_states=True` of the kept forward signature (llava_llama.py:56-69): the reference's tuple (fixture
    hidden_*.npz made by make_golden.py `hidden`) = input embeddings, residual stream after each layer but the last, final-normed
    states; the oracle collects the same L + 1 tensors."""
    g = TY.load_npz("hidden_" + name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    assert abs(TY.sd_checksum(sd) - float(g["sd_checksum"])) < 1e-6 * float(g["sd_checksum"])
    frames = O.synthetic_frames(int(g["T"]), vcfg.image_size, seed=0)
    ids = torch.from_numpy(g["input_ids"])
    hs = []
    logits, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, hidden_states=hs)
    assert len(hs) == int(g["n_states"]) == lcfg.num_hidden_layers + 1 and float(g["logits_from_last"]) == 0.0
    got = torch.stack(hs)[:, 0]
    assert got.shape[1] == int(g["S"])
    np.testing.assert_allclose(got[:, torch.from_numpy(g["sel"])].numpy(), g["hidden_sel"], atol=2e-5)
    np.testing.assert_allclose(got.double().abs().sum((1, 2)).numpy(), g["hidden_sum_abs"], rtol=1e-5)
    assert float((O._lin(hs[-1], sd["lm_head.weight"]) - logits).abs().max()) == 0.0


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_oracle_attentions_match_reference(name):
    """Round 6: `output_attentions=True` of the kept forward signature (llava_llama.py:65,95): the reference's per-layer [1, H, S, S] maps
    (fixture attn_*.npz made by make_golden.py `attn`) against the oracle's attention_probs -- 16 query rows of every (layer, head) with
    all their keys, rows that sum to 1, exact zeros above the diagonal, and the per-layer sum of p^2."""
    g = TY.load_npz("attn_" + name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    assert abs(TY.sd_checksum(sd) - float(g["sd_checksum"])) < 1e-6 * float(g["sd_checksum"])
    frames = O.synthetic_frames(int(g["T"]), vcfg.image_size, seed=0)
    ids = torch.from_numpy(g["input_ids"])
    att = []
    O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, attentions=att)
    got = torch.stack(att)[:, 0]
    assert list(got.shape) == g["shape"].tolist() and float(g["upper_triangle_max"]) == 0.0
    np.testing.assert_allclose(got[:, :, torch.from_numpy(g["sel"])].numpy(), g["attn_sel"], atol=2e-6)
    np.testing.assert_allclose(got.double().pow(2).sum((1, 2, 3)).numpy(), g["sum_p2"], rtol=1e-5)
    assert float(got.triu(1).abs().max()) == 0.0 and float((got.sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_oracle_training_shape_loss_matches_reference(name):
    """N4: batch of 2, right padding, labels -> the reference's loss and logits (tests/golden/train_*.npz)."""
    g = TY.load_npz("train_" + name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    ids, mask, labels, frames = TY.train_batch(lcfg.vocab_size, vcfg.image_size)
    loss, logits, lab = O.training_loss(ids, mask, labels, frames, sd, vcfg, lcfg, mm)
    assert list(logits.shape) == g["logits_shape"].tolist()
    assert int((lab[:, 1:] != -100).sum()) == int(g["n_supervised"])
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    assert abs(float(logits[0].double().abs().sum()) - float(g["logits_sum_abs_row0"])) < 1e-4 * float(g["logits_sum_abs_row0"])
    np.testing.assert_allclose(logits[0, -1].numpy(), g["logits_last_valid"][0], atol=2e-5)


@pytest.mark.parametrize("name", ["tinyA", "tinyB", "tinyC"])
def test_kernel_emulating_attention_modes_are_pinned(name, monkeypatch):
    """The bf16 parity chain compares the HIP kernels with the oracle's `flash64` (MFMA flash kernel: 64-key tiles,
    running max, exp2) and `split128` (decode kernel: independent key chunks + combine) attention modes.  Pin those modes
    themselves: with R = identity they must reproduce the REFERENCE's fp32 outputs (tests/golden/tiny*.npz) to fp32
    round-off, exactly like mode "exact" does -- ViT features (flash64, D=64 heads), end-to-end prefill logits (flash64
    causal), greedy tokens / step logits / KV snapshots (split128 decode steps)."""
    g = TY.load_npz(name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    T = int(g["T"])
    frames = O.synthetic_frames(T, vcfg.image_size, seed=0)
    ids = torch.from_numpy(g["input_ids"])
    monkeypatch.setattr(O, "FORCE_KERNEL_MODES", True)
    hd_v = vcfg.hidden_size // vcfg.num_attention_heads
    assert O.kernel_attention_mode(None, hd_v, vcfg.num_positions) == ("flash64" if hd_v in (64, 128) else "exact")
    assert O.kernel_attention_mode(None, lcfg.head_dim, 1, decode_kernel=True) == "split128"
    feats = O.vit_features(torch.stack(frames), sd, vcfg, -2, "patch")
    np.testing.assert_allclose(feats.numpy(), g["vit_features"], atol=FP32_TOL, rtol=1e-5)
    logits, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm)
    sel = torch.from_numpy(g["e2e_sel"])
    np.testing.assert_allclose(logits[0][sel].numpy(), g["e2e_logits_sel"], atol=FP32_TOL, rtol=1e-5)
    toks, step_logits, cache = O.greedy_generate(ids, frames, sd, vcfg, lcfg, mm, max_new_tokens=len(g["greedy_tokens"]))
    assert toks == g["greedy_tokens"].tolist()
    np.testing.assert_allclose(step_logits.numpy(), g["greedy_logits"], atol=FP32_TOL, rtol=1e-5)
    ks = torch.from_numpy(g["kv_sel"])
    np.testing.assert_allclose(cache.k[-1][0][:, ks].numpy(), g["k_last"], atol=FP32_TOL, rtol=1e-5)
    np.testing.assert_allclose(cache.v[-1][0][:, ks].numpy(), g["v_last"], atol=FP32_TOL, rtol=1e-5)


def test_kernel_emulating_modes_agree_with_exact_on_masked_ragged_shapes():
    """flash64 / split128 vs exact on shapes the goldens do not hold: kv_len not a multiple of the tile, q_len < kv_len
    (cached prefix), fully masked leading tiles for late queries are impossible under causal masks but partially masked
    tiles are common; also a peaked score distribution (running-max rescale by many orders of magnitude)."""
    g = torch.Generator().manual_seed(7)
    ident = lambda x: x
    for (Sq, Sk, d, peak) in [(5, 133, 64, 1.0), (70, 70, 128, 1.0), (1, 300, 128, 1.0), (33, 257, 64, 30.0)]:
        q = torch.randn(1, 2, Sq, d, generator=g) * peak
        k = torch.randn(1, 2, Sk, d, generator=g)
        v = torch.randn(1, 2, Sk, d, generator=g)
        qpos = torch.arange(Sk - Sq, Sk).view(Sq, 1)
        vis = (torch.arange(Sk).view(1, Sk) <= qpos).view(1, 1, Sq, Sk)
        ref = O.attention_core(q, k, v, vis, d ** -0.5, ident, "exact")
        for mode in ("flash64", "split128"):
            out = O.attention_core(q, k, v, vis, d ** -0.5, ident, mode)
            assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (mode, Sq, Sk)


def test_preallocated_kv_cache_equals_the_growing_cache():
    """bench.py's cpu_baseline times the oracle with KVCache(capacity=...): same numbers as the torch.cat cache, bit for bit."""
    name = "tinyA"
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(1, 11, lcfg.hidden_size, generator=g)
    a_log, a_cache = O.llama_forward(emb, None, None, None, sd, lcfg)
    b_log, b_cache = O.llama_forward(emb, None, None, O.KVCache(capacity=16), sd, lcfg)
    assert torch.equal(a_log, b_log)
    for _ in range(3):
        e = torch.randn(1, 1, lcfg.hidden_size, generator=g)
        a_log, a_cache = O.llama_forward(e, None, None, a_cache, sd, lcfg)
        b_log, b_cache = O.llama_forward(e, None, None, b_cache, sd, lcfg)
        assert torch.equal(a_log, b_log)
    assert a_cache.length == b_cache.length == 14
    assert all(torch.equal(x, y) for x, y in zip(a_cache.k + a_cache.v, b_cache.k + b_cache.v))
    with pytest.raises(ValueError, match="capacity"):
        O.llama_forward(torch.randn(1, 3, lcfg.hidden_size), None, None, b_cache, sd, lcfg)
    sdt = O.make_state_dict_for_timing(vcfg, lcfg, mm)
    assert set(sdt) == set(sd) and all(sdt[k].shape == sd[k].shape for k in sd)


# ---------------------------------------------------------------------------------------------- H6 / N3: preprocessing
def test_preprocess_analytic_pins():
    """oracle.preprocess_image (processing_image.py:15-25) against facts that hold for ANY correct implementation:
    identity at 224x224 (ToTensor / 255 + Normalize, closed form); constants survive the resize exactly (the anti-aliased
    bicubic weights sum to one); a linear ramp is reproduced away from the borders (the cubic kernel has linear precision);
    the crop window is the centred one; Resize geometry = torchvision's (shorter edge -> 224, int() of the other)."""
    g = torch.Generator().manual_seed(0)
    raw = torch.randint(0, 256, (224, 224, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor(O.OPENAI_DATASET_MEAN).view(3, 1, 1)
    std = torch.tensor(O.OPENAI_DATASET_STD).view(3, 1, 1)
    want = (raw.permute(2, 0, 1).float() / 255.0 - mean) / std
    assert torch.equal(O.preprocess_image(raw), want) and torch.equal(O.preprocess_uint8(raw), want)
    for (h, w) in [(300, 448), (448, 300), (100, 150), (1000, 700)]:
        c = torch.tensor([37, 141, 250], dtype=torch.uint8).view(1, 1, 3).expand(h, w, 3)
        out = O.preprocess_image(c)
        assert out.shape == (3, 224, 224)
        ref = (torch.tensor([37, 141, 250]).float() / 255.0 - mean.view(3)) / std.view(3)
        assert float((out - ref.view(3, 1, 1)).abs().max()) < 2e-6
    assert O.resize_geometry(300, 448) == (224, 334, 0, 55) and O.resize_geometry(448, 300) == (334, 224, 55, 0)
    assert O.resize_geometry(231, 517) == (224, 501, 0, 138)          # int(224 * 517 / 231) = 501; round(138.5) = 138 (half to even)
    assert O.resize_geometry(224, 224) == (224, 224, 0, 0)
    # linear ramp along x, down-scaled 2x: out(x) = a * (x_src) + b with x_src = (x_out + left + 0.5) * scale - 0.5
    h, w = 448, 896
    ramp = (torch.arange(w).float() * (255.0 / (w - 1))).round().to(torch.uint8).view(1, w, 1).expand(h, w, 3)
    out = O.preprocess_image(ramp)
    nh, nw, top, left = O.resize_geometry(h, w)
    xs = (torch.arange(224).float() + left + 0.5) * (w / nw) - 0.5
    lin = ((xs * (255.0 / (w - 1)) / 255.0) - mean[0, 0, 0]) / std[0, 0, 0]
    assert float((out[0, 100, 8:-8] - lin[8:-8]).abs().max()) < 4e-3 / float(std[0, 0, 0])   # uint8 rounding of the ramp itself: 0.5/255


@pytest.mark.parametrize("h,w", [(300, 448), (448, 300), (672, 672), (180, 200)])
def test_preprocess_cross_check_against_pil_bicubic(h, w):
    """Independent implementation of the same resampling: PIL's BICUBIC resize is the anti-aliased a = -0.5 cubic convolution
    ATen's `_upsample_bicubic2d_aa` was written to match (PIL works on uint8 with fixed-point rounding, so the comparison is
    to ~1.5 grey levels on smooth content)."""
    from PIL import Image
    yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    img = torch.stack([(torch.sin(yy / 17.0) * torch.cos(xx / 23.0) * 0.5 + 0.5) * 255, (yy / h) * 255, (xx / w) * 255], dim=-1)
    raw = img.round().clamp(0, 255).to(torch.uint8)
    nh, nw, top, left = O.resize_geometry(h, w)
    pil = Image.fromarray(raw.numpy()).resize((nw, nh), Image.BICUBIC)
    pil_t = torch.from_numpy(np.array(pil)).permute(2, 0, 1).float()[:, top:top + 224, left:left + 224]
    mean = torch.tensor(O.OPENAI_DATASET_MEAN).view(3, 1, 1)
    std = torch.tensor(O.OPENAI_DATASET_STD).view(3, 1, 1)
    ours = (O.preprocess_image(raw) * std + mean) * 255.0
    d = (ours - pil_t).abs()
    assert float(d.max()) < 2.0 and float(d.mean()) < 0.6, (float(d.max()), float(d.mean()))


def test_expand2square_matches_pil_paste():
    """mm_utils.py:14-25 on arrays == the reference's PIL implementation (Image.new + paste), incl. the odd-difference offsets."""
    from PIL import Image
    g = torch.Generator().manual_seed(5)
    fill = O.pad_fill_from_mean(O.OPENAI_DATASET_MEAN)
    assert fill == (122, 116, 104)
    for (h, w) in [(5, 8), (8, 5), (7, 7), (10, 3), (3, 10)]:
        raw = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        pil = Image.fromarray(raw.numpy())
        W, H = pil.size
        if W == H:
            ref = pil
        elif W > H:
            ref = Image.new(pil.mode, (W, W), fill)
            ref.paste(pil, (0, (W - H) // 2))
        else:
            ref = Image.new(pil.mode, (H, H), fill)
            ref.paste(pil, ((H - W) // 2, 0))
        assert torch.equal(O.expand2square_u8(raw, fill), torch.from_numpy(np.array(ref)))


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_teacher_forced_decode_steps_match_reference(name):
    """G6b: 12 decode steps driven with a prescribed, varied token sequence (the greedy streams of the tiny models settle on one
    token after two steps) -- every step's logits against the reference's, so the decode path is pinned beyond step 2."""
    g = TY.load_npz(name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    frames = O.synthetic_frames(int(g["T"]), vcfg.image_size, seed=0)
    ids = torch.from_numpy(g["input_ids"])
    _, cache, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm)
    emb_w = sd["model.embed_tokens.weight"]
    assert len(set(g["forced_logits"].argmax(-1).tolist())) >= 4          # the stream really varies
    for i, t in enumerate(g["forced_tokens"].tolist()):
        logits, cache = O.llama_forward(emb_w[torch.tensor([[t]])], None, None, cache, sd, lcfg, None, decode_kernel=True)
        np.testing.assert_allclose(logits[0, -1].numpy(), g["forced_logits"][i], atol=FP32_TOL, rtol=1e-5, err_msg=f"step {i}")
