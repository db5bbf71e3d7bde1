#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE ITSELF (build container only).

Usage:  python tests/golden/make_golden.py          (writes tests/golden/*.npz / *.json)

The reference (/root/reference, read-only) is imported unmodified through tests/golden/ref_import.py
and executed on CPU with tiny seeded configurations.  Only inputs/outputs (data) are stored; weights
are re-derived from seeds by oracle.teo_oracle.make_state_dict and guarded by a checksum.
The arithmetic underneath the reference here is transformers 5.15.0 (pin is 4.31.0; not installable).

Fixtures (SURVEY.md section 8c list):
  host.json        G1 prompts + G2 tokenizer_image_token ids (through the reference's run_inference_single)
                   G8 KeywordsStoppingCriteria truth table
  splice.json      G3 prepare_inputs_labels_for_multimodal index plans / masks / positions / labels
  tinyA.npz/tinyB.npz  G4 ViT features, G5 projector, G6 LLaMA prefill+decode, G7 end-to-end logits
  hidden_tiny*.npz     `output_hidden_states=True` of the same forward (round 5)
  attn_tiny*.npz       `output_attentions=True` of the same forward (round 6)
"""
import ast
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from oracle import teo_oracle as O  # noqa: E402
from teochat_amd.tokenizer_stub import ByteTokenizer  # noqa: E402

torch.set_grad_enabled(False)

TINY = {
    # generic dims (exercise the shape-agnostic kernels); GQA; quick_gelu
    "tinyA": dict(
        vit=dict(hidden_size=64, num_attention_heads=4, intermediate_size=128, num_hidden_layers=3,
                 hidden_act="quick_gelu"),
        llm=dict(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, intermediate_size=128,
                 num_hidden_layers=2, vocab_size=300),
    ),
    # MFMA-friendly dims (head_dim 64 / 128 like the real model); MHA; gelu
    "tinyB": dict(
        vit=dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=3,
                 hidden_act="gelu"),
        llm=dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=2, intermediate_size=512,
                 num_hidden_layers=2, vocab_size=512),
    ),
}


TINY_STD = 0.08   # large enough that attention is peaked and greedy tokens vary at tiny dims

# round 6: tinyC (anchored successor cycle: a greedy stream with 8 distinct tokens) is defined once, in tests/_tiny.py
from tests import _tiny as TY  # noqa: E402
TINY["tinyC"] = TY.TINY["tinyC"]
assert all(TY.TINY[k] == TINY[k] for k in TINY) and TY.TINY_STD == TINY_STD


def cfgs(name):
    t = TINY[name]
    v = O.VitCfg(**t["vit"])
    l = O.LlamaCfg(**t["llm"])
    mm = O.MMCfg(mm_hidden_size=v.hidden_size)
    return v, l, mm


def sd_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


# ----------------------------------------------------------------------------------------------
# reference model assembly
# ----------------------------------------------------------------------------------------------
def ref_tower_class():
    """Compile the reference's LanguageBindImageTower class body from its own source file without
    running that file's top-level imports (they pull in video/audio deps that are absent here)."""
    path = os.path.join(ref_import.REF_ROOT, "videollava/model/multimodal_encoder/languagebind/__init__.py")
    tree = ast.parse(open(path).read(), filename=path)
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LanguageBindImageTower"][0]
    mod = ast.Module(body=[node], type_ignores=[])
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(mod, path, "exec"), ns)
    return ns["LanguageBindImageTower"]


def build_reference(name):
    ref_import.install()
    LL = ref_import.import_llava_llama()
    from videollava.model.multimodal_encoder.languagebind.image import modeling_image as MI
    from videollava.model.multimodal_encoder.languagebind.image import configuration_image as CI
    import videollava.model.multimodal_encoder.builder as EB

    vcfg, lcfg, mm = cfgs(name)
    Tower = ref_tower_class()

    def factory(cfg):
        t = Tower.__new__(Tower)
        torch.nn.Module.__init__(t)
        t.is_loaded = True
        t.image_tower_name = "stub/LanguageBind_Image"
        t.select_layer = cfg.mm_vision_select_layer
        t.select_feature = getattr(cfg, "mm_vision_select_feature", "patch")
        vc = CI.CLIPVisionConfig(hidden_size=vcfg.hidden_size, intermediate_size=vcfg.intermediate_size,
                                 num_hidden_layers=vcfg.num_hidden_layers,
                                 num_attention_heads=vcfg.num_attention_heads, image_size=vcfg.image_size,
                                 patch_size=vcfg.patch_size, hidden_act=vcfg.hidden_act,
                                 layer_norm_eps=vcfg.layer_norm_eps, add_time_attn=False, lora_r=0)
        vc._attn_implementation = "eager"
        t.image_tower = MI.CLIPVisionTransformer(vc)
        return t

    EB._tower_factory = factory
    cfg = LL.LlavaConfig(hidden_size=lcfg.hidden_size, intermediate_size=lcfg.intermediate_size,
                         num_hidden_layers=lcfg.num_hidden_layers, num_attention_heads=lcfg.num_attention_heads,
                         num_key_value_heads=lcfg.num_key_value_heads, vocab_size=lcfg.vocab_size,
                         rms_norm_eps=lcfg.rms_norm_eps, max_position_embeddings=4096,
                         rope_theta=lcfg.rope_theta, attention_bias=False, tie_word_embeddings=False)
    cfg.pretraining_tp = 1
    cfg.mm_image_tower = "stub/LanguageBind_Image"
    cfg.mm_hidden_size = mm.mm_hidden_size
    cfg.mm_projector_type = mm.mm_projector_type
    cfg.mm_vision_select_layer = mm.mm_vision_select_layer
    cfg.mm_vision_select_feature = mm.mm_vision_select_feature
    cfg._attn_implementation = "eager"
    model = LL.LlavaLlamaForCausalLM(cfg)
    model.eval()
    sd = O.make_state_dict(vcfg, lcfg, mm, seed=2, std=TINY_STD)
    if TINY[name].get("anchors"):
        sd = TY.apply_anchors(sd, TINY[name]["anchors"], TINY_STD)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    bad = [k for k in missing if "post_layernorm" not in k and "inv_freq" not in k and "position_ids" not in k]
    assert not bad, bad
    return model, sd, (vcfg, lcfg, mm)


# ----------------------------------------------------------------------------------------------
# G1/G2/G8: host-side fixtures through the reference's own run_inference_single
# ----------------------------------------------------------------------------------------------
class _RecTok(ByteTokenizer):
    def __init__(self, add_bos=True):
        super().__init__(add_bos)
        self.calls = []

    def __call__(self, text, **kw):
        self.calls.append(text)
        return super().__call__(text, **kw)


class _FakeProc:
    def preprocess(self, path, return_tensors=None):
        return {"pixel_values": [torch.zeros(3, 2, 2)]}


class _FakeModel:
    device = torch.device("cpu")

    def __init__(self, reply_ids):
        self.reply_ids = reply_ids
        self.seen = None

    def generate(self, input_ids=None, images=None, **kw):
        self.seen = dict(input_ids=input_ids.clone(), n_images=len(images), kw={k: v for k, v in kw.items()
                                                                               if k != "stopping_criteria"})
        return torch.cat([input_ids, torch.tensor([self.reply_ids], dtype=torch.long)], dim=1)


def gen_host():
    ref_import.install()
    import videollava.eval.inference as INF
    import videollava.mm_utils as MU

    out = {"run_inference_single": [], "tokenizer_image_token": [], "stopping": []}
    reply = ByteTokenizer(add_bos=False)("Two new buildings.</s>").input_ids
    cases = []
    for T in (1, 2, 8, 16):
        for strat in (None, "interleave"):
            for chrono in (True, False):
                cases.append(dict(inp="<video>\nThese images were taken at times: 2016, 2018. What changed?",
                                  T=T, strategy=strat, chrono=chrono, timestamps=[]))
    cases.append(dict(inp="<video>\nDescribe.", T=3, strategy="interleave", chrono=True,
                      timestamps=["2020-05-01", "2018-01-31", "2019-07-04"]))
    cases.append(dict(inp="No frames token here.", T=0, strategy="interleave", chrono=True, timestamps=[]))
    for c in cases:
        tok = _RecTok()
        fm = _FakeModel(reply)
        paths = ["frame_%d.png" % i for i in range(c["T"])]
        text = INF.run_inference_single(fm, _FakeProc(), tok, c["inp"], paths, conv_mode="v1",
                                        timestamps=list(c["timestamps"]), prompt_strategy=c["strategy"],
                                        chronological_prefix=c["chrono"], temperature=0.2, max_new_tokens=16)
        n_chunks = c["inp"].count("<video>") * max(c["T"], 0) + 1 if "<video>" in c["inp"] else 1
        rec = dict(c)
        rec["prompt_chunks"] = tok.calls[:n_chunks]
        rec["input_ids"] = fm.seen["input_ids"][0].tolist()
        rec["n_images"] = fm.seen["n_images"]
        rec["generate_kwargs"] = {k: (v if not torch.is_tensor(v) else v.tolist()) for k, v in fm.seen["kw"].items()}
        rec["output_text"] = text
        out["run_inference_single"].append(rec)

    # G2 edge cases straight into tokenizer_image_token
    edge = ["<image>", "<image><image>", "a<image>", "<image>a", "a<image><image>b<image>", "", "plain text",
            "x<image>y<image>z", "Image 1: <image>Image 2: <image> end"]
    for add_bos in (True, False):
        for p in edge:
            tok = ByteTokenizer(add_bos=add_bos)
            ids = MU.tokenizer_image_token(p, tok, -200, return_tensors=None)
            out["tokenizer_image_token"].append(dict(prompt=p, add_bos=add_bos, ids=list(map(int, ids))))

    # G8 stopping criterion
    tok = ByteTokenizer()
    base = tok("USER: hi ASSISTANT:").input_ids
    ok = ByteTokenizer(add_bos=False)
    tails = ["", "a", "ab</s>", "</s>", "abc</s>d", "hello", "</s></s>", "x</", "</s"]
    for kw in (["</s>"], ["###"], ["</s>", "END"]):
        for tl in tails + ["##", "###", "a###b", "EN", "END"]:
            ids = torch.tensor([base + ok(tl).input_ids], dtype=torch.long)
            crit = MU.KeywordsStoppingCriteria(kw, tok, torch.tensor([base]))
            res = bool(crit(ids, None))
            out["stopping"].append(dict(keywords=kw, prompt_len=len(base), row=ids[0].tolist(), stop=res))
    # batch semantics: all() over rows
    crit = MU.KeywordsStoppingCriteria(["</s>"], tok, torch.tensor([base]))
    rows = torch.tensor([base + ok("ab</s>").input_ids, base + ok("abc").input_ids])
    out["stopping_batch"] = dict(keywords=["</s>"], prompt_len=len(base), rows=rows.tolist(),
                                 stop=bool(crit(rows, None)))
    json.dump(out, open(os.path.join(HERE, "host.json"), "w"), indent=0)
    print("host.json", len(out["run_inference_single"]), len(out["tokenizer_image_token"]), len(out["stopping"]))


# ----------------------------------------------------------------------------------------------
# G3: splice index plans (integer work -> bit exact)
# ----------------------------------------------------------------------------------------------
def gen_splice():
    model, sd, (vcfg, lcfg, mm) = build_reference("tinyA")
    V, D = lcfg.vocab_size, lcfg.hidden_size
    NV = 5      # visual rows per image for the coded encoder
    # coded tables: embed row v -> value v ; feature (img i, row j) -> -(1000*i + j + 1)
    with torch.no_grad():
        model.model.embed_tokens.weight.copy_(torch.arange(V, dtype=torch.float32).view(V, 1).expand(V, D))

    def coded_encode(images_minibatch):
        n = images_minibatch.shape[0]
        f = torch.zeros(n, NV, D)
        for i in range(n):
            for j in range(NV):
                f[i, j] = -(1000 * i + j + 1)
        return f

    model.encode_images = coded_encode
    IMG = -200
    cases = {
        "single_T2": dict(ids=[[1, 10, IMG, 11, 12, IMG, 13]], n_images=2),
        "lead_trail": dict(ids=[[IMG, 10, 11, IMG]], n_images=2),
        "adjacent": dict(ids=[[1, IMG, IMG, IMG, 7]], n_images=3),
        "batch_unequal": dict(ids=[[1, 10, IMG, 11, 0, 0, 0], [1, IMG, 20, IMG, 21, IMG, 22]], n_images=4,
                              mask=[[1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1]]),
        "batch_zero_image": dict(ids=[[1, 30, 31, 32], [1, IMG, 40, 41]], n_images=2),
        "truncate": dict(ids=[[1, 10, IMG, 11, 12, IMG, 13, 14]], n_images=2, max_len=9),
        "left_pad": dict(ids=[[1, 10, IMG, 11], [1, IMG, 20, IMG]], n_images=3, padding_side="left"),
        "labels": dict(ids=[[1, 10, IMG, 11, 12]], n_images=1, labels=[[-100, -100, -100, 11, 12]]),
        "labels_batch_mask": dict(ids=[[1, IMG, 11, 12, 0], [1, 5, IMG, 6, 7]], n_images=2,
                                  mask=[[1, 1, 1, 1, 0], [1, 1, 1, 1, 1]],
                                  labels=[[-100, -100, 11, 12, -100], [-100, -100, -100, 6, 7]],
                                  pos=True),
    }
    out = {"NV": NV}
    for name, c in cases.items():
        model.config.tokenizer_model_max_length = c.get("max_len")
        model.config.tokenizer_padding_side = c.get("padding_side", "right")
        ids = torch.tensor(c["ids"], dtype=torch.long)
        mask = torch.tensor(c["mask"], dtype=torch.long) if "mask" in c else None
        labels = torch.tensor(c["labels"], dtype=torch.long) if "labels" in c else None
        pos = torch.arange(ids.shape[1]).unsqueeze(0).expand(ids.shape[0], -1).clone() if c.get("pos") else None
        images = [torch.zeros(3, 224, 224) for _ in range(c["n_images"])]
        r = model.prepare_inputs_labels_for_multimodal(ids.clone(), pos, mask, None, labels, images)
        assert r[0] is None
        emb = r[4]
        code = emb[:, :, 0].round().long()
        assert torch.equal(emb, emb[:, :, :1].expand_as(emb))
        rec = dict(c)
        rec["plan"] = code.tolist()           # >=0: vocab id (0 may be a pad row); <0: -(1000*img + row + 1)
        rec["embeds_is_zero_row"] = (emb.abs().sum(-1) == 0).tolist()
        rec["position_ids"] = None if r[1] is None else r[1].tolist()
        rec["attention_mask"] = None if r[2] is None else r[2].long().tolist()
        rec["attention_mask_dtype"] = None if r[2] is None else str(r[2].dtype)
        rec["labels_out"] = None if r[5] is None else r[5].tolist()
        out[name] = rec
    # error convention: more <image> than images -> IndexError (llava_arch.py:284)
    model.config.tokenizer_model_max_length = None
    model.config.tokenizer_padding_side = "right"
    try:
        model.prepare_inputs_labels_for_multimodal(torch.tensor([[1, IMG, IMG]]), None, None, None, None,
                                                   [torch.zeros(3, 224, 224)])
        out["too_few_images_error"] = None
    except Exception as e:  # noqa: BLE001
        out["too_few_images_error"] = type(e).__name__

    # decode branch (llava_arch.py:154-163) with a legacy tuple cache
    past = ((torch.zeros(2, 2, 9, 4), torch.zeros(2, 2, 9, 4)),)
    am = torch.tensor([[1, 1, 1, 1], [0, 1, 1, 1]], dtype=torch.long)
    r = model.prepare_inputs_labels_for_multimodal(torch.tensor([[5], [6]]), None, am, past, None,
                                                   [torch.zeros(3, 224, 224)])
    out["decode_branch"] = dict(past_len=9, mask_in=am.tolist(), input_ids=[[5], [6]],
                                position_ids=r[1].tolist(), attention_mask=r[2].tolist(),
                                embeds_is_none=r[4] is None)
    # images=None passthrough
    r = model.prepare_inputs_labels_for_multimodal(torch.tensor([[1, 2, 3]]), None, None, None, None, None)
    out["no_images_passthrough"] = dict(input_ids=r[0].tolist(), rest_none=all(x is None for x in r[1:]))
    json.dump(out, open(os.path.join(HERE, "splice.json"), "w"), indent=0)
    print("splice.json", list(out.keys()))


# ----------------------------------------------------------------------------------------------
# G4-G7 numeric fixtures
# ----------------------------------------------------------------------------------------------
SEL_ROWS = 16     # logits rows kept (evenly spaced) for prefill


def gen_numeric(name):
    model, sd, (vcfg, lcfg, mm) = build_reference(name)
    T, n_text, n_new = 2, 24, 8
    frames = O.synthetic_frames(T, vcfg.image_size, seed=0)
    ids = TY.prompt_ids(name, n_text, T, lcfg.vocab_size, seed=1).unsqueeze(0)      # == O.synthetic_prompt_ids for tinyA / tinyB
    out = {"sd_checksum": np.float64(sd_checksum(sd)), "input_ids": ids.numpy(), "T": np.int64(T),
           "frames_checksum": np.float64(float(sum(f.double().abs().sum() for f in frames)))}

    tower = model.get_model().get_image_tower()
    pix = torch.stack(frames)
    # G4: tower output = hidden_states[-2][:, 1:]
    feats = tower(pix)
    out["vit_features"] = feats.numpy()
    hs = tower.image_tower(pix, output_hidden_states=True).hidden_states
    out["vit_hidden0_row0"] = hs[0][:, :4].numpy()           # pre_layrnorm(embeddings), first rows incl. CLS
    out["vit_n_states"] = np.int64(len(hs))
    # G5: projector
    proj = model.get_model().mm_projector(feats)
    out["projector_rows"] = proj[:, ::8].numpy()
    enc = model.encode_images(pix)
    assert torch.equal(enc, proj)

    # G7: end-to-end prefill through the reference forward
    res = model(input_ids=ids, images=frames, use_cache=True)
    logits = res.logits[0]
    L = logits.shape[0]
    sel = torch.linspace(0, L - 1, SEL_ROWS).long()
    out["e2e_L"] = np.int64(L)
    out["e2e_sel"] = sel.numpy()
    out["e2e_logits_sel"] = logits[sel].numpy()
    out["e2e_logits_sum_abs"] = np.float64(float(logits.double().abs().sum()))
    out["e2e_argmax_all"] = logits.argmax(-1).numpy()

    # G6: greedy decode, manual loop around the reference forward (generate() is broken under tf-5.15)
    cache = res.past_key_values
    nxt = int(logits[-1].argmax())
    toks, dec_logits = [nxt], [logits[-1].numpy()]
    for _ in range(n_new - 1):
        r2 = model(input_ids=torch.tensor([[nxt]]), past_key_values=cache, use_cache=True)
        cache = r2.past_key_values
        nxt = int(r2.logits[0, -1].argmax())
        toks.append(nxt)
        dec_logits.append(r2.logits[0, -1].numpy())
    out["greedy_tokens"] = np.array(toks, dtype=np.int64)
    out["greedy_logits"] = np.stack(dec_logits)
    # G6b: teacher-forced decode -- the greedy streams of these tiny random models settle on one token after two steps, so the
    # steps are ALSO driven with a prescribed varied token sequence (fresh cache from the same prefill): logits of 12 steps
    res_f = model(input_ids=ids, images=frames, use_cache=True)
    cache_f = res_f.past_key_values
    forced = torch.randint(0, lcfg.vocab_size, (12,), generator=torch.Generator().manual_seed(11)).tolist()
    forced_logits = []
    for t in forced:
        rf = model(input_ids=torch.tensor([[t]]), past_key_values=cache_f, use_cache=True)
        cache_f = rf.past_key_values
        forced_logits.append(rf.logits[0, -1].numpy())
    out["forced_tokens"] = np.array(forced, dtype=np.int64)
    out["forced_logits"] = np.stack(forced_logits)
    # KV snapshot (layer 0 and last), selected positions
    try:
        k0, v0 = cache.layers[0].keys, cache.layers[0].values
        k1, v1 = cache.layers[-1].keys, cache.layers[-1].values
    except AttributeError:
        k0, v0 = cache[0]
        k1, v1 = cache[-1]
    ks = torch.linspace(0, k0.shape[2] - 1, 12).long()
    out["kv_sel"] = ks.numpy()
    out["kv_len"] = np.int64(k0.shape[2])
    out["k_layer0"] = k0[0][:, ks].numpy()
    out["v_layer0"] = v0[0][:, ks].numpy()
    out["k_last"] = k1[0][:, ks].numpy()
    out["v_last"] = v1[0][:, ks].numpy()

    # LLaMA alone from given embeddings (G6 first half): text-only prompt, no images
    tids = O.synthetic_prompt_ids(20, 0, lcfg.vocab_size, seed=5).unsqueeze(0)
    r3 = model(input_ids=tids, images=None)
    out["text_only_ids"] = tids.numpy()
    out["text_only_logits"] = r3.logits[0].numpy()

    # information only: the reference run in bf16 on CPU (its low-precision path), last-row logits
    m16 = model.to(torch.bfloat16)
    r16 = m16(input_ids=ids, images=[f.to(torch.bfloat16) for f in frames])
    out["e2e_bf16_last_logits"] = r16.logits[0, -1].float().numpy()
    out["e2e_bf16_logits_sel"] = r16.logits[0][sel].float().numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "L=%d" % L, "greedy", toks, "size KB",
          os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024)


def gen_hidden(name):
    """Round 5: `output_hidden_states=True` of the kept forward signature (llava_llama.py:56-69) -- what the reference's forward returns in
    `hidden_states` for the G7 inputs and for the batch-of-2 training shape: count, shapes, and 16 evenly spaced rows of every state."""
    model, sd, (vcfg, lcfg, mm) = build_reference(name)
    T, n_text = 2, 24
    frames = O.synthetic_frames(T, vcfg.image_size, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, lcfg.vocab_size, seed=1).unsqueeze(0)
    res = model(input_ids=ids, images=frames, use_cache=True, output_hidden_states=True)
    hs = torch.stack(list(res.hidden_states))[:, 0]              # [L + 1, S, D]
    S = hs.shape[1]
    sel = torch.linspace(0, S - 1, SEL_ROWS).long()
    out = {"sd_checksum": np.float64(sd_checksum(sd)), "input_ids": ids.numpy(), "T": np.int64(T), "n_states": np.int64(hs.shape[0]),
           "S": np.int64(S), "sel": sel.numpy(), "hidden_sel": hs[:, sel].numpy(), "hidden_sum_abs": hs.double().abs().sum((1, 2)).numpy(),
           "logits_from_last": np.float64(float((model.lm_head(hs[-1]) - res.logits[0]).abs().max()))}
    bids, bmask, blabels, bframes = train_batch(lcfg.vocab_size, vcfg.image_size)
    rb = model(input_ids=bids, attention_mask=bmask, labels=blabels, images=bframes, output_hidden_states=True)
    hb = torch.stack(list(rb.hidden_states))                     # [L + 1, 2, W, D]
    out["batch_hidden_shape"] = np.array(hb.shape, dtype=np.int64)
    am = model.prepare_inputs_labels_for_multimodal(bids, None, bmask, None, blabels, bframes)[2]      # mask of the spliced rows
    out["batch_mask"] = am.numpy()
    rows = [torch.nonzero(am[b]).flatten() for b in range(2)]
    out["batch_rows0"], out["batch_rows1"] = rows[0][::37].numpy(), rows[1][::37].numpy()
    out["batch_hidden0"], out["batch_hidden1"] = hb[:, 0, rows[0][::37]].numpy(), hb[:, 1, rows[1][::37]].numpy()
    np.savez_compressed(os.path.join(HERE, "hidden_" + name + ".npz"), **out)
    print("hidden", name, tuple(hs.shape), "lm_head(hidden[-1]) - logits:", out["logits_from_last"], "batch", tuple(hb.shape))


def gen_attn(name):
    """Round 6: `output_attentions=True` of the kept forward signature (llava_llama.py:65,95): what the reference's forward returns in
    `attentions` for the G7 inputs -- one [1, H, S, S] map per layer (eager attention: softmax over the causal keys).  Stored: 16 evenly
    spaced query rows of every (layer, head) with all their keys, and per-layer checksums (row sums, sum of p^2)."""
    model, sd, (vcfg, lcfg, mm) = build_reference(name)
    T, n_text = 2, 24
    frames = O.synthetic_frames(T, vcfg.image_size, seed=0)
    ids = TY.prompt_ids(name, n_text, T, lcfg.vocab_size, seed=1).unsqueeze(0)
    res = model(input_ids=ids, images=frames, use_cache=True, output_attentions=True)
    att = torch.stack(list(res.attentions))[:, 0]                # [L, H, S, S]
    assert att.dim() == 4 and att.shape[0] == lcfg.num_hidden_layers and att.shape[1] == lcfg.num_attention_heads
    S = att.shape[2]
    sel = torch.linspace(0, S - 1, SEL_ROWS).long()
    out = {"sd_checksum": np.float64(sd_checksum(sd)), "input_ids": ids.numpy(), "T": np.int64(T), "shape": np.array(att.shape, dtype=np.int64),
           "sel": sel.numpy(), "attn_sel": att[:, :, sel].numpy(), "row_sum_max_dev": np.float64(float((att.sum(-1) - 1).abs().max())),
           "sum_p2": att.double().pow(2).sum((1, 2, 3)).numpy(), "upper_triangle_max": np.float64(float(att.triu(1).abs().max()))}
    np.savez_compressed(os.path.join(HERE, "attn_" + name + ".npz"), **out)
    print("attn", name, tuple(att.shape), "row sums off by", out["row_sum_max_dev"], "above the diagonal:", out["upper_triangle_max"])


from tests._tiny import train_batch  # noqa: E402  (shared with the tests that replay the fixture)


def gen_train(name):
    """N4: loss + logits of the reference's training-shape forward on a padded batch."""
    model, sd, (vcfg, lcfg, mm) = build_reference(name)
    ids, mask, labels, frames = train_batch(lcfg.vocab_size, vcfg.image_size)
    res = model(input_ids=ids, attention_mask=mask, labels=labels, images=frames)
    out = {"loss": np.float64(float(res.loss)), "logits_shape": np.array(res.logits.shape),
           "logits_sum_abs_row0": np.float64(float(res.logits[0].double().abs().sum())),
           "logits_last_valid": np.stack([res.logits[0, -1].detach().numpy(), res.logits[1, res.logits.shape[1] - 1].detach().numpy()])}
    # per-token losses of the shifted positions that carry a label (for a kernel-level check)
    (_, _, _, _, _, lab) = model.prepare_inputs_labels_for_multimodal(ids.clone(), None, mask, None, labels, frames)
    sl = res.logits[:, :-1].reshape(-1, res.logits.shape[-1])
    tl = lab[:, 1:].reshape(-1)
    per = torch.nn.functional.cross_entropy(sl, tl, ignore_index=-100, reduction="none")
    out["n_supervised"] = np.int64(int((tl != -100).sum()))
    out["per_token_loss_sum"] = np.float64(float(per.double().sum()))
    np.savez_compressed(os.path.join(HERE, "train_" + name + ".npz"), **out)
    print("train", name, "loss", float(res.loss), "supervised", int(out["n_supervised"]))


def gen_metrics():
    """N4: classification_metrics truth table from the reference (videollava/eval/classification.py:15-41)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_classification", os.path.join(ref_import.REF_ROOT, "videollava/eval/classification.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    outputs = [
        {"response": "Yes.", "ground_truth": "yes", "task": "qa"},
        {"response": "No", "ground_truth": "yes", "task": "qa"},
        {"response": "Residential Area!", "ground_truth": "residential area", "task": "cls"},
        {"response": "a port", "ground_truth": "Port", "task": "cls"},
        {"response": "There is major damage here", "ground_truth": "Major Damage", "task": "dmg"},
        {"response": "destroyed", "ground_truth": "no damage", "task": "dmg"},
        {"response": "no-damage", "ground_truth": "No Damage.", "task": "dmg"},
    ]
    cases = {}
    import contextlib
    import io
    for nm, kw in (("default", {}), ("case_sensitive", {"ignore_casing": False}), ("keep_punct", {"ignore_punctuation": False}),
                   ("keywords", {"keywords": ["major damage", "no damage", "yes"]})):
        with contextlib.redirect_stdout(io.StringIO()):
            cases[nm] = {"kwargs": kw, "result": mod.classification_metrics(outputs, **kw)}
    json.dump({"outputs": outputs, "cases": cases}, open(os.path.join(HERE, "metrics.json"), "w"), indent=1)
    print("metrics.json", {k: v["result"] for k, v in cases.items()})


def _shapely_wkt_stub():
    """shapely is absent from this image; the reference uses it ONLY to turn WKT text into coordinate lists (wkt.loads,
    polygon.exterior.coords, iteration over multi-geometries; detection.py:4-5,146-147,171,206,236).  This generator-side
    stand-in does that text parsing (independently of teochat_amd.detection.parse_wkt) so that the reference's own
    rasterisation and metric code can run unmodified.  Geometry objects follow shapely 1.8 iteration semantics (a
    MultiPolygon iterates over its polygons, as the reference's `iter(polygons)` expects)."""
    import re
    import types

    class _Ring:
        def __init__(self, coords):
            self.coords = coords

    class _Polygon:
        def __init__(self, rings):
            self.exterior = _Ring(rings[0])

    def _coords(txt):
        return [tuple(float(v) for v in pt.split()[:2]) for pt in txt.split(",")]

    def _poly(txt):                                     # "(x y, x y, ...), (hole...)" -> exterior only
        rings = re.findall(r"\(([^()]*)\)", txt)
        return _Polygon([_coords(r) for r in rings])

    def loads(w):
        if not isinstance(w, str):
            return [loads(x) for x in w]
        w = w.strip()
        if w.upper().startswith("MULTIPOLYGON"):
            body = w[w.index("(") + 1:w.rindex(")")]
            return [_poly(m) for m in re.findall(r"\((\([^()]*\)(?:\s*,\s*\([^()]*\))*)\)", body)]
        if w.upper().startswith("POLYGON"):
            return _poly(w[w.index("(") + 1:w.rindex(")")])
        raise ValueError(w)

    shapely = types.ModuleType("shapely")
    wkt = types.ModuleType("shapely.wkt")
    wkt.loads = loads
    shapely.wkt = wkt
    return shapely, wkt


def detection_cases():
    """Records in the reference's output format (eval/inference.py:112-135) for every dataset family of detection_metrics."""
    sq = "POLYGON ((40 40, 40 120, 120 120, 120 40, 40 40))"
    tri = "POLYGON ((10.5 200.25, 100 130, 60.75 250, 10.5 200.25))"
    multi = "MULTIPOLYGON (((5 5, 5 30, 30 30, 30 5, 5 5)), ((200 200, 200 240, 250 240, 250 200, 200 200), (210 210, 210 220, 220 220, 220 210, 210 210)))"
    loc = [
        {"response": "[15, 15, 47, 47]", "ground_truth": "[16, 16, 47, 47]", "task": "change_detection_localization", "polygon": sq},
        {"response": "[0, 50, 40, 100], [78, 78, 98, 94] and [bad, box]", "ground_truth": "[4, 51, 39, 98]", "task": "change_detection_localization", "polygon": tri},
        {"response": "No changes.", "ground_truth": "[2, 2, 12, 12], [78, 78, 98, 94]", "task": "change_detection_localization", "polygon": multi},
        {"response": "[10.5, 20.25, 33, 44.75]", "ground_truth": "none", "task": "change_detection_localization", "polygon": sq},
    ]
    dmg = [
        {"response": "No damage.", "ground_truth": "No damage", "task": "change_detection_classification", "polygon": sq},
        {"response": "Destroyed", "ground_truth": "Major damage", "task": "change_detection_classification", "polygon": tri},
        {"response": "major damage", "ground_truth": "Major Damage", "task": "change_detection_classification", "polygon": multi},
        {"response": "it is fine", "ground_truth": "Minor damage", "task": "change_detection_classification", "polygon": sq},
        {"response": "Destroyed", "ground_truth": "unclassified", "task": "change_detection_classification", "polygon": sq},
        {"response": "minor damage", "ground_truth": "minor damage", "task": "change_detection_classification", "polygon": tri},
    ]
    sre = [dict(r, task="spatial_referring_expression") for r in loc[:3]] + [
        {"response": "Yes", "ground_truth": "yes", "task": "question_answering"},
        {"response": "in the Top Left corner", "ground_truth": "top left", "task": "question_answering"},
        {"response": "bottom", "ground_truth": "center", "task": "question_answering"},
        {"response": "Major damage", "ground_truth": "major damage.", "task": "region_based_question_answering"},
        {"response": "none", "ground_truth": "destroyed", "task": "region_based_question_answering"},
    ]
    s2 = [dict(r, task="change_detection_detection") for r in loc]
    qf2 = [
        {"response": "Residential", "ground_truth": "residential", "task": "region_based_question_answering", "polygon": sq},
        {"response": "road", "ground_truth": "Commercial", "task": "region_based_question_answering", "polygon": tri},
        {"response": "mega-projects", "ground_truth": "Mega projects", "task": "region_based_question_answering", "polygon": multi},
        {"response": "a lake", "ground_truth": "industrial", "task": "region_based_question_answering", "polygon": sq},
    ]
    qf5 = [
        {"response": "Land cleared", "ground_truth": "land cleared", "task": "region_based_temporal_question_answering", "polygon": sq},
        {"response": "operational", "ground_truth": "Construction done.", "task": "region_based_temporal_question_answering", "polygon": tri},
        {"response": "land-cleared", "ground_truth": "Greenland", "task": "region_based_temporal_question_answering", "polygon": multi},
        {"response": "demolition", "ground_truth": "demolition", "task": "region_based_question_answering", "polygon": tri},
    ]
    tre = [
        {"response": "image 2", "ground_truth": "Image 2", "task": "temporal_referring_expression"},
        {"response": "image 1", "ground_truth": "image 3", "task": "temporal_referring_expression"},
        {"response": "Excavation.", "ground_truth": "excavation", "task": "region_based_temporal_question_answering"},
    ]
    return {"xbd_loc": loc, "xbd_dmg_cls": dmg, "xbd_sre_qa_rqa": sre, "s2_det": s2,
            "s2_sre_qa": [r for r in sre if r["task"] != "region_based_question_answering"],
            "s2_rqa": [r for r in sre if r["task"] == "region_based_question_answering"],
            "qfabric_rqa2": qf2, "qfabric_rqa5_rtqa5": qf5, "qfabric_tre_rtqa": tre}


def gen_detection():
    """N4: detection_metrics / Evaluator / create_mask / extract_bboxes / run_inference bookkeeping, produced by the
    reference's own code (videollava/eval/detection.py, eval/inference.py:80-137)."""
    import contextlib
    import importlib.util
    import io
    import types
    shapely, wkt = _shapely_wkt_stub()
    sys.modules["shapely"], sys.modules["shapely.wkt"] = shapely, wkt
    ref_import.install()
    if "tqdm" not in sys.modules:
        pass
    det = importlib.import_module("videollava.eval.detection")
    out = {"cases": {}}
    for ds, recs in detection_cases().items():
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            res = det.detection_metrics(recs, ds)
        out["cases"][ds] = {"outputs": recs, "result": {k: float(v) for k, v in res.items()}}
    # all eight pixel metrics of evaluate_masks on the localisation records
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        em = det.evaluate_masks(detection_cases()["xbd_loc"], "xbd_loc")
        cdc = det.change_detection_classification(detection_cases()["xbd_dmg_cls"], ["no damage", "minor damage", "major damage", "destroyed"],
                                                  skip_classes=["unclassified"])
    out["evaluate_masks_xbd_loc"] = {k: float(v) for k, v in em.items()}
    out["change_detection_classification_xbd"] = {k: float(v) for k, v in cdc.items()}
    # masks: pixel counts and a checksum per polygon / box list
    masks = {}
    for nm, w in (("sq", detection_cases()["xbd_loc"][0]["polygon"]), ("tri", detection_cases()["xbd_loc"][1]["polygon"]),
                  ("multi", detection_cases()["xbd_loc"][2]["polygon"])):
        m = det.create_mask(wkt.loads(w), (256, 256))
        ys, xs = np.nonzero(m)
        masks[nm] = {"sum": int(m.sum()), "weighted": int((ys * 257 + xs).sum()), "shape": list(m.shape)}
    out["masks"] = masks
    # Evaluator on a 3-class confusion matrix
    ev = det.Evaluator(3)
    rng = np.random.RandomState(0)
    gt = rng.randint(0, 3, size=(32, 32))
    pr = rng.randint(0, 3, size=(32, 32))
    ev.add_batch(gt, pr)
    out["evaluator3"] = {"cm": ev.confusion_matrix.tolist(), "oa": float(ev.Pixel_Accuracy()), "macc": float(ev.Pixel_Accuracy_Class()[0]),
                         "miou": float(ev.Mean_Intersection_over_Union()), "kappa": float(ev.Kappa_coefficient()),
                         "fwiou": float(ev.Frequency_Weighted_Intersection_over_Union()),
                         "damage_f1": [float(v) for v in ev.Damage_F1_socore()], "cw_f1": float(ev.Class_Weighted_F1_score())}
    out["get_classes"] = {"xbd": det.get_classes("xbd", "classification: Classify the level of damage experienced by the building at location [bbox] in the second image. Choose from: No damage, Minor Damage, Major Damage, Destroyed."),
                          "none": det.get_classes("fmow", "x")}
    # run_inference bookkeeping with the reference loop and a canned run_inference_single
    lm = ref_import.import_llava_llama()  # noqa: F841  (inference.py imports the model package)
    inf = importlib.import_module("videollava.eval.inference")
    examples = [
        {"conversations": [{"value": "<video> Is the building at [12, 30, 45, 60] damaged?"}, {"value": "Yes, see [10, 28, 47, 61] and [1, 2, 3, 4]."}],
         "video": ["a.png", "b.png"], "timestamp": ["2020-01-01", "2019-01-01"], "task": "question_answering", "polygon": "POLYGON ((1 1, 1 2, 2 2, 1 1))"},
        {"conversations": [{"value": "<video> Describe the changes."}, {"value": "Nothing [1,2,3,4] changed [5, 6, 7, 8.5]."}],
         "video": ["c.png"], "timestamp": [], "task": "captioning"},
    ]
    real = inf.run_inference_single
    calls = []

    def fake_single(model, processor, tokenizer, inp, image_paths, **kw):
        calls.append({"inp": inp, "image_paths": list(image_paths), "kw": {k: (v if not isinstance(v, list) else list(v)) for k, v in kw.items()}})
        return f"answer {len(calls)}"

    inf.run_inference_single = fake_single
    try:
        with contextlib.redirect_stderr(io.StringIO()):
            outs = inf.run_inference(examples, "M", "T", "P", "interleave", True, "v1", 0.2, 64)
    finally:
        inf.run_inference_single = real
    out["run_inference"] = {"examples": examples, "outputs": outs, "calls": calls,
                            "extract": {s_: inf.extract_bboxes(s_) for s_ in ["[1, 2, 3, 4]", "[1,2,3,4]", "x [10, 20, 30, 40] y [5, 6, 7, 8]", "[1, 2, 3]", "[-1, 2, 3, 4]"]}}
    json.dump(out, open(os.path.join(HERE, "detection.json"), "w"), indent=1)
    print("detection.json", {k: v["result"] for k, v in out["cases"].items()})


if __name__ == "__main__":
    assert ref_import.available(), "reference tree required"
    torch.manual_seed(0)
    which = sys.argv[1:] or ["host", "splice", "numeric", "train", "hidden", "attn", "metrics", "detection"]
    if "host" in which:
        gen_host()
    if "splice" in which:
        gen_splice()
    for nm in TINY:
        if "numeric" in which:
            gen_numeric(nm)
        if nm == "tinyC":                     # the anchored config exists for its token stream: G4-G7 only
            continue
        if "train" in which:
            gen_train(nm)
        if "hidden" in which:
            gen_hidden(nm)
        if "attn" in which:
            gen_attn(nm)
    if "metrics" in which:
        gen_metrics()
    if "detection" in which:
        gen_detection()
