"""Checkpoint loading (N2), loud failure without a GPU, and edge cases of the drop-in surface."""
import json
import os

import pytest
import torch

from oracle import teo_oracle as O
from tests import _tiny as TY


def _write_checkpoint(tmp_path, name="tinyA"):
    from safetensors.torch import save_file
    from teochat_amd.config import LlavaConfig, VisionConfig
    t = TY.TINY[name]
    cfg = LlavaConfig(**t["llm"], mm_hidden_size=t["vit"]["hidden_size"], max_position_embeddings=1024,
                      vision_config=VisionConfig(**t["vit"]))
    sd = TY.state_dict(name)
    d = tmp_path / "llava-tiny"
    d.mkdir()
    keys = sorted(sd)
    half = len(keys) // 2                      # two shards, like an HF sharded checkpoint
    save_file({k: sd[k].contiguous() for k in keys[:half]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k].contiguous() for k in keys[half:]}, str(d / "model-00002-of-00002.safetensors"))
    json.dump(cfg.to_dict(), open(d / "config.json", "w"))
    return str(d), cfg, sd


def test_config_and_lazy_safetensors_roundtrip(tmp_path):
    from teochat_amd.builder import CheckpointDir
    from teochat_amd.config import LlavaConfig
    path, cfg, sd = _write_checkpoint(tmp_path)
    c2 = LlavaConfig.from_pretrained(path)
    for k in ("hidden_size", "num_key_value_heads", "mm_projector_type", "mm_vision_select_layer", "rms_norm_eps"):
        assert getattr(c2, k) == getattr(cfg, k)
    assert c2.vision_config.hidden_act == cfg.vision_config.hidden_act and c2.head_dim == cfg.head_dim
    assert c2.vision_config_resolved                                       # config.json carried a vision_config block
    lz = CheckpointDir(path, "cpu")
    assert set(lz.keys()) == set(sd)
    for k in ("model.embed_tokens.weight", "model.mm_projector.2.bias", O.VIT_PREFIX + "pre_layrnorm.weight"):
        assert torch.equal(lz[k], sd[k])
    with pytest.raises(KeyError):
        lz["nope"]


def _write_lora(tmp_path, sd, cfg, r=4, alpha=8.0, as_bin=False):
    """A peft adapter directory over the tiny checkpoint: q_proj / v_proj of every LLaMA layer + non_lora_trainables
    (projector), named the way peft / the reference's trainer save them."""
    from safetensors.torch import save_file
    d = tmp_path / "llava-lora-tiny"
    d.mkdir()
    g = torch.Generator().manual_seed(5)
    adapter, merged = {}, {k: v.clone() for k, v in sd.items()}
    for i in range(cfg.num_hidden_layers):
        for n in ("q_proj", "v_proj"):
            k = f"model.layers.{i}.self_attn.{n}.weight"
            out, inp = sd[k].shape
            A = torch.randn(r, inp, generator=g) * 0.1
            B = torch.randn(out, r, generator=g) * 0.1
            adapter[f"base_model.model.model.layers.{i}.self_attn.{n}.lora_A.weight"] = A
            adapter[f"base_model.model.model.layers.{i}.self_attn.{n}.lora_B.weight"] = B
            merged[k] = sd[k] + (alpha / r) * (B @ A)
    if as_bin:
        torch.save(adapter, str(d / "adapter_model.bin"))
    else:
        save_file(adapter, str(d / "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "target_modules": ["q_proj", "v_proj"], "peft_type": "LORA"},
              open(d / "adapter_config.json", "w"))
    nl = {}
    for k in sd:
        if "mm_projector" in k:
            merged[k] = sd[k] * 1.5 + 0.01
            nl["base_model.model." + k] = merged[k].clone()
    torch.save(nl, str(d / "non_lora_trainables.bin"))
    json.dump(cfg.to_dict(), open(d / "config.json", "w"))
    return str(d), merged


@pytest.mark.parametrize("as_bin", [False, True])
def test_lora_and_projector_branches_merge_like_the_reference(tmp_path, as_bin):
    """builder.py:37-72 (LoRA: base + non_lora_trainables + merge_and_unload) and :73-88 (base + mm_projector.bin)."""
    from teochat_amd.builder import CheckpointDir, open_checkpoint, unwrap_peft_tower
    base, cfg, sd = _write_checkpoint(tmp_path)
    lora_dir, merged = _write_lora(tmp_path, sd, cfg, as_bin=as_bin)
    src = open_checkpoint(lora_dir, base, "llava-lora-tiny", "cpu")
    assert set(src.keys()) == set(sd)
    for k in sd:
        torch.testing.assert_close(src[k], merged[k], atol=1e-6, rtol=1e-6)
    # projector-only branch
    pd = tmp_path / "llava-proj"
    pd.mkdir()
    proj = {k: v * 2 for k, v in sd.items() if "mm_projector" in k}
    torch.save(proj, str(pd / "mm_projector.bin"))
    src = open_checkpoint(str(pd), base, "llava-tiny", "cpu")
    for k in sd:
        assert torch.equal(src[k], proj.get(k, sd[k]))
    # pytorch_model-*.bin shards instead of safetensors
    bd = tmp_path / "llava-bin"
    bd.mkdir()
    keys = sorted(sd)
    torch.save({k: sd[k] for k in keys[::2]}, str(bd / "pytorch_model-00001-of-00002.bin"))
    torch.save({k: sd[k] for k in keys[1::2]}, str(bd / "pytorch_model-00002-of-00002.bin"))
    cd = CheckpointDir(str(bd), "cpu")
    assert set(cd.keys()) == set(sd) and all(torch.equal(cd[k], sd[k]) for k in keys)
    # peft-wrapped tower inside a merged checkpoint (modeling_image.py:775-793)
    from safetensors.torch import save_file
    wd = tmp_path / "llava-wrapped"
    wd.mkdir()
    enc = O.VIT_PREFIX + "encoder."
    wrapped, want = {}, {}
    g = torch.Generator().manual_seed(9)
    for k, v in sd.items():
        if k.startswith(enc):
            rest = k[len(enc):]
            if any(rest.endswith(f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj", "out_proj")):
                stem = enc + "base_model.model." + rest[:-len(".weight")]
                A, B = torch.randn(2, v.shape[1], generator=g) * 0.1, torch.randn(v.shape[0], 2, generator=g) * 0.1
                wrapped[stem + ".base_layer.weight"] = v
                wrapped[stem + ".lora_A.default.weight"] = A
                wrapped[stem + ".lora_B.default.weight"] = B
                want[k] = v + (16.0 / 2) * (B @ A)
            else:
                wrapped[enc + "base_model.model." + rest.replace("self_attn.q_proj.bias", "self_attn.q_proj.base_layer.bias")] = v
                want[k] = v
        else:
            wrapped[k] = v
            want[k] = v
    save_file({k: v.contiguous() for k, v in wrapped.items()}, str(wd / "model.safetensors"))
    vcfg = cfg.vision_config
    vcfg.lora_r, vcfg.lora_alpha = 2, 16.0
    src = unwrap_peft_tower(CheckpointDir(str(wd), "cpu"), vcfg)
    assert set(src.keys()) == set(sd)
    for k in sd:
        torch.testing.assert_close(src[k], want[k], atol=1e-6, rtol=1e-6)


@pytest.mark.gpu
def test_load_lora_checkpoint_end_to_end(tmp_path):
    """load_pretrained_model(lora_dir, base, 'llava-lora-...') == an engine built from the manually merged weights."""
    from teochat_amd.builder import load_pretrained_model
    from teochat_amd.engine import TeoEngine
    from teochat_amd.model import LlavaLlamaForCausalLM
    base, cfg, sd = _write_checkpoint(tmp_path)
    lora_dir, merged = _write_lora(tmp_path, sd, cfg)
    # on the LoRA branch the model object comes from from_pretrained(model_base) (builder.py:51), so GenerationMixin's
    # fallback knobs are model_base's generation_config.json (a LoRA directory has none)
    json.dump({"do_sample": True, "temperature": 0.6, "top_p": 0.9, "eos_token_id": 7}, open(os.path.join(base, "generation_config.json"), "w"))
    _, model, _, _ = load_pretrained_model(lora_dir, base, "llava-lora-tiny", device="cuda:0", dtype=torch.float32, max_seq=1024)
    gc = model.generation_config
    assert (gc.do_sample, gc.temperature, gc.top_p, gc.top_k, gc.eos_token_id) == (True, 0.6, 0.9, 50, 7)
    assert model._config_eos() == 7                        # generate(eos_token_id="config") consults generation_config first
    ref = LlavaLlamaForCausalLM(cfg, TeoEngine(merged, cfg, dtype=torch.float32, device="cuda:0", max_seq=1024))
    g = TY.load_npz("tinyA")
    ids = torch.from_numpy(g["input_ids"]).cuda()
    frames = [f.cuda() for f in O.synthetic_frames(int(g["T"]), cfg.vision_config.image_size, seed=0)]
    a = model(input_ids=ids, images=frames).logits
    b = ref(input_ids=ids, images=frames).logits
    assert torch.equal(a, b)
    vcfg, lcfg, mm = TY.cfgs("tinyA")
    want, _, _ = O.mm_forward(ids.cpu(), [f.cpu() for f in frames], merged, vcfg, lcfg, mm)
    assert float((a[0].cpu() - want[0]).abs().max()) < 1e-4


def test_product_fails_loudly_without_gpu_or_library(tmp_path, monkeypatch):
    from teochat_amd import _lib as L
    from teochat_amd.builder import load_pretrained_model
    path, _, _ = _write_checkpoint(tmp_path)
    with pytest.raises(FileNotFoundError):                         # LoRA branch: the base directory must hold weights
        load_pretrained_model(path, str(tmp_path / "no-such-base"), "llava-lora-tiny")
    with pytest.raises(ValueError):                                # not a llava/teochat checkpoint name (builder.py:33)
        load_pretrained_model(path, None, "vicuna-7b")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="MI355X"):          # no CPU fallback
            load_pretrained_model(path, None, "llava-tiny")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(L.TeoLibraryError):
        L.load()


@pytest.mark.gpu
def test_load_model_from_safetensors_equals_direct_engine(tmp_path):
    from teochat_amd.eval import load_model
    from tests.test_model_gpu import build
    path, cfg, sd = _write_checkpoint(tmp_path)
    tok, model, proc = load_model(path, None, device="cuda:0", dtype=torch.float32, max_seq=1024)
    ref_model, _ = build("tinyA", torch.float32)
    frames = O.synthetic_frames(2, 224, seed=0)
    ids = O.synthetic_prompt_ids(24, 2, cfg.vocab_size, seed=1).unsqueeze(0).cuda()
    a = model(input_ids=ids, images=[f.cuda() for f in frames]).logits
    b = ref_model(input_ids=ids, images=[f.cuda() for f in frames]).logits
    assert torch.equal(a, b)
    assert model.model.video_tower is None and proc.crop_size == {"height": 224, "width": 224}


@pytest.mark.gpu
def test_edge_cases_generate_and_forward():
    from teochat_amd.mm_utils import KeywordsStoppingCriteria
    from teochat_amd.tokenizer_stub import ByteTokenizer
    from tests.test_model_gpu import build
    model, sd = build("tinyA", torch.float32)
    dev = model.device
    vcfg, lcfg, mm = TY.cfgs("tinyA")
    frames = [f.to(dev) for f in O.synthetic_frames(1, 224, seed=3)]
    ids = torch.tensor([[1, 20, -200, 21, 22]], device=dev)
    # T = 1, max_new_tokens = 1 and 0
    g1 = model.generate(input_ids=ids, images=frames, max_new_tokens=1, eos_token_id=None)
    assert g1.shape == (1, 6)
    assert model.generate(input_ids=ids, images=frames, max_new_tokens=0).shape == (1, 5)
    ref, _, _ = O.greedy_generate(ids.cpu(), [f.cpu() for f in frames], sd, vcfg, lcfg, mm, max_new_tokens=6)
    g6 = model.generate(input_ids=ids, images=frames, max_new_tokens=6, eos_token_id=None)[0, 5:].tolist()
    assert g6 == ref
    # EOS: stop exactly at the token (inclusive), as HF does
    g_eos = model.generate(input_ids=ids, images=frames, max_new_tokens=6, eos_token_id=ref[2])[0, 5:].tolist()
    assert g_eos == ref[:ref.index(ref[2]) + 1]
    # keyword stopping criterion on ids (device-side suffix test + host check per chunk)
    class Tok(ByteTokenizer):
        def __call__(self, text, **kw):
            r = super().__call__(text, **kw)
            return r
    crit = KeywordsStoppingCriteria(["</s>"], ByteTokenizer(), ids)
    assert crit.keyword_id_lists == [[2]]
    out = model.generate(input_ids=ids, images=frames, max_new_tokens=6, eos_token_id=None, stopping_criteria=[crit], chunk=4)
    assert out[0, 5:].tolist() == ref          # "</s>" (id 2) never generated -> runs to max_new_tokens
    # the reference's default call (eval/inference.py:57-72): keyword "</s>" AND EOS, both the id sequence [2] -> ONE device-side
    # stop candidate (VERDICT r02 missing #4: duplicates used to switch the device stop off)
    model.generate(input_ids=ids, images=frames, max_new_tokens=6, eos_token_id=2, stopping_criteria=[crit], chunk=4)
    assert model.engine.decode_state.n_stop_ids == 1 and int(model.engine.d_stop_ids[0]) == 2
    stop_tok = next((t for t in ref[1:] if t != ref[0]), None)       # a token the DECODE loop (not the prefill) produces first
    if stop_tok is not None:
        crit2 = KeywordsStoppingCriteria(["</s>"], ByteTokenizer(), ids)
        crit2.keyword_id_lists = [[stop_tok]]
        crit2.keyword_ids = [torch.tensor([stop_tok])]
        cut = model.generate(input_ids=ids, images=frames, max_new_tokens=6, eos_token_id=stop_tok, stopping_criteria=[crit2], chunk=6)
        assert cut[0, 5:].tolist() == ref[:ref.index(stop_tok) + 1]
        assert int(model.engine.d_stop.item()) == 1      # the device-side id-suffix test fired inside the chunk
    # text-only prompt (images=None) and a prompt made only of an image
    t_ids = torch.tensor([[1, 5, 6, 7]], device=dev)
    lo, _ = O.llama_forward(sd["model.embed_tokens.weight"][t_ids.cpu()], None, None, None, sd, lcfg)
    torch.testing.assert_close(model(input_ids=t_ids, images=None).logits.cpu(), lo, atol=1e-4, rtol=1e-4)
    only = model(input_ids=torch.tensor([[-200, 5]], device=dev), images=frames).logits
    assert only.shape == (1, 257, lcfg.vocab_size)
    # a 1-token input takes the decode-branch early-out (llava_arch.py:154) even if it is a sentinel: the embedding
    # lookup then rejects -200 exactly like nn.Embedding does in the reference
    with pytest.raises(IndexError):
        model(input_ids=torch.tensor([[-200]], device=dev), images=frames)
    # errors: wrong image size, too many sentinels, sequence beyond max_seq, 4-D (video) item, unknown strategy
    with pytest.raises(ValueError):
        model(input_ids=ids, images=[torch.zeros(3, 112, 112, device=dev)])
    with pytest.raises(IndexError):
        model(input_ids=torch.tensor([[1, -200, -200]], device=dev), images=frames)
    with pytest.raises(ValueError):
        model.generate(input_ids=ids, images=frames, max_new_tokens=5000)
    with pytest.raises(ValueError):
        model(input_ids=ids, images=[torch.zeros(2, 3, 224, 224, device=dev)])
    with pytest.raises(NotImplementedError):        # the attention maps are a single-sequence feature (served at B = 1 since round 6)
        model(input_ids=torch.cat([ids, ids]), images=list(frames) + list(frames), output_attentions=True)


@pytest.mark.gpu
def test_truncation_and_loss_shapes():
    from tests.test_model_gpu import build
    model, sd = build("tinyA", torch.float32, tokenizer_model_max_length=200)
    dev = model.device
    vcfg, lcfg, _ = TY.cfgs("tinyA")
    mm = O.MMCfg(mm_hidden_size=vcfg.hidden_size, tokenizer_model_max_length=200)
    frames = O.synthetic_frames(1, 224, seed=3)
    ids = torch.tensor([[1, 20, -200, 21, 22]])
    labels = torch.tensor([[-100, -100, -100, 21, 22]])
    out = model(input_ids=ids.to(dev), labels=labels.to(dev), images=[f.to(dev) for f in frames])
    lo, _, emb = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm)
    assert out.logits.shape[1] == 200 == lo.shape[1]              # truncated (llava_arch.py:295-299): the text tail is cut
    torch.testing.assert_close(out.logits.cpu(), lo, atol=1e-4, rtol=1e-4)
    assert out.loss is not None and bool(torch.isnan(out.loss))    # every label was truncated away -> nan, as torch gives


@pytest.mark.gpu
def test_real_width_two_layer_model_bf16_vs_oracle():
    """LLaMA-2-7B / ViT-L widths (4096/11008/32 heads, 1024/4096/16 heads, vocab 32000) with 2 layers each: the MFMA,
    flash-attention, GEMV and decode kernels at their production tile shapes against the CPU oracle."""
    from teochat_amd.config import LlavaConfig, VisionConfig
    from teochat_amd.engine import TeoEngine
    from teochat_amd.model import LlavaLlamaForCausalLM
    vit = dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=3, hidden_act="gelu")
    llm = dict(hidden_size=4096, num_attention_heads=32, num_key_value_heads=32, intermediate_size=11008, num_hidden_layers=2,
               vocab_size=32000)
    vcfg, lcfg, mm = O.VitCfg(**vit), O.LlamaCfg(**llm), O.MMCfg()
    sd = O.make_state_dict(vcfg, lcfg, mm, seed=2, std=0.02)
    sd16 = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    cfg = LlavaConfig(**llm, max_position_embeddings=1024, vision_config=VisionConfig(**vit))
    eng = TeoEngine(sd, cfg, dtype=torch.bfloat16, device="cuda:0", max_seq=1024)
    model = LlavaLlamaForCausalLM(cfg, eng)
    frames = O.synthetic_frames(2, 224, seed=0)
    ids = O.synthetic_prompt_ids(32, 2, 32000, seed=1).unsqueeze(0)
    imgs = [f.to("cuda:0", dtype=torch.bfloat16) for f in frames]
    got = model(input_ids=ids.cuda(), images=imgs).logits[0].cpu()
    ref, _, _ = O.mm_forward(ids, frames, sd16, vcfg, lcfg, mm, rounding="bf16")
    rel = float((got - ref[0]).abs().max()) / float(ref.abs().max())
    print(f"real-width 2-layer bf16 logits rel-to-max diff vs boundary oracle: {rel:.2e}")
    assert rel < 2e-2
    assert (got.argmax(-1) == ref[0].argmax(-1)).float().mean() > 0.9
    toks, _, _ = O.greedy_generate(ids, frames, sd16, vcfg, lcfg, mm, max_new_tokens=4, rounding="bf16")
    gen = model.generate(input_ids=ids.cuda(), images=imgs, do_sample=False, max_new_tokens=4, eos_token_id=None)
    assert gen.shape[1] == ids.shape[1] + 4
    print("greedy", gen[0, ids.shape[1]:].tolist(), "oracle", toks)


def _write_tower_dir(root, vit_cfg_dict, tower_sd=None):
    """A LanguageBind_Image-style checkpoint directory: config.json with a `vision_config` block and (optionally) the
    tower weights under `vision_model.*` (the names LanguageBindImage.state_dict() uses, modeling_image.py:763)."""
    from safetensors.torch import save_file
    d = root / "LanguageBind_Image"
    d.mkdir()
    json.dump({"model_type": "LanguageBindImage", "vision_config": vit_cfg_dict, "text_config": {}}, open(d / "config.json", "w"))
    if tower_sd is not None:
        save_file({k: v.contiguous() for k, v in tower_sd.items()}, str(d / "model.safetensors"))
    return str(d)


def test_image_tower_is_resolved_from_mm_image_tower_not_defaulted(tmp_path, monkeypatch):
    """a real llava/teochat config.json has no vision_config and (for a plain-LLaMA model_base) the
    checkpoint has no tower weights; the reference reads both from the `mm_image_tower` repo.  The loader must take
    hidden_act / lora_r / the weights from a local copy of that repo, or fail with an explicit error -- never default."""
    from teochat_amd import builder as B
    from teochat_amd.config import LlavaConfig
    path, cfg, sd = _write_checkpoint(tmp_path)
    # (1) strip the vision_config from config.json and the tower weights from the shards
    cj = json.load(open(os.path.join(path, "config.json")))
    vit = cj.pop("vision_config")
    cj["mm_image_tower"] = "LanguageBind/LanguageBind_Image"
    json.dump(cj, open(os.path.join(path, "config.json"), "w"))
    from safetensors.torch import save_file
    for f in os.listdir(path):
        if f.endswith(".safetensors"):
            os.remove(os.path.join(path, f))
    llm_only = {k: v.contiguous() for k, v in sd.items() if not k.startswith(O.VIT_PREFIX)}
    save_file(llm_only, os.path.join(path, "model.safetensors"))
    c2 = LlavaConfig.from_pretrained(path)
    assert not c2.vision_config_resolved
    src = B.CheckpointDir(path, "cpu")
    monkeypatch.delenv("TEOCHAT_IMAGE_TOWER", raising=False)
    with pytest.raises(FileNotFoundError, match="never guessed"):
        B.resolve_image_tower(c2, src, path, None, "cpu")
    # (2) a local copy of the tower next to the checkpoint: config (gelu, not the quick_gelu default) and weights come from it
    vit["hidden_act"] = "gelu"                                               # differs from VisionConfig's quick_gelu default
    tower_sd = {"vision_model." + k[len(O.VIT_PREFIX):]: v for k, v in sd.items() if k.startswith(O.VIT_PREFIX)}
    tdir = _write_tower_dir(tmp_path, dict(vit, lora_r=0), tower_sd)
    assert B.find_tower_dir(c2, path) == tdir                                # sibling directory named like the repo
    merged = B.resolve_image_tower(c2, src, path, None, "cpu")
    assert c2.vision_config_resolved and c2.vision_config.hidden_act == "gelu"
    assert c2.vision_config.num_hidden_layers == vit["num_hidden_layers"]
    for k in (O.VIT_PREFIX + "embeddings.class_embedding", O.VIT_PREFIX + "encoder.layers.1.mlp.fc2.bias", "model.norm.weight"):
        assert torch.equal(merged[k], sd[k])
    assert set(merged.keys()) == set(sd)
    # (3) $TEOCHAT_IMAGE_TOWER wins; a tower config without hidden_act is rejected
    bad = tmp_path / "bad_tower"
    bad.mkdir()
    json.dump({"vision_config": {"hidden_size": 64}}, open(bad / "config.json", "w"))
    monkeypatch.setenv("TEOCHAT_IMAGE_TOWER", str(bad))
    c3 = LlavaConfig.from_pretrained(path)
    with pytest.raises(ValueError, match="hidden_act"):
        B.resolve_image_tower(c3, src, path, None, "cpu")


def test_generation_config_json_overlays_the_defaults(tmp_path):
    from types import SimpleNamespace
    from teochat_amd.builder import load_generation_config
    m = SimpleNamespace(generation_config=SimpleNamespace(do_sample=False, top_k=50, top_p=1.0, temperature=1.0, eos_token_id=2))
    assert load_generation_config(m, str(tmp_path)) is False                  # no file: HF defaults stay
    assert (m.generation_config.top_p, m.generation_config.top_k) == (1.0, 50)
    json.dump({"do_sample": True, "temperature": 0.6, "top_p": 0.9, "eos_token_id": 2, "transformers_version": "4.31.0"},
              open(tmp_path / "generation_config.json", "w"))                 # what LLaMA-2 ships
    assert load_generation_config(m, str(tmp_path)) is True
    gc = m.generation_config
    assert (gc.do_sample, gc.temperature, gc.top_p, gc.top_k) == (True, 0.6, 0.9, 50)
    assert load_generation_config(m, "synthetic:teochat-7b") is False
