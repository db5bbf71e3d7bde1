"""Batched decode (config C5's variant, teochat_amd/batch.py + teo_llama_decode_batch_*): every conversation of a batch
must produce what it produces alone.

  fp32 : the batched step runs the same kernels row by row -> token streams, KV caches and logits are compared
         bit-for-bit / to FP32_TOL with the single-conversation path, the reference golden tokens and the CPU oracle.
  bf16 : the weights go through the MFMA skinny GEMM (operand-tiled copy) -> logits within BF16_REL of the
         single-conversation path (fp32 accumulation order differs), tokens compared where the top-2 margin allows.
"""
import numpy as np
import pytest
import torch

from oracle import teo_oracle as O
from tests import _tiny as TY
from tests.test_model_gpu import BF16_REL, FP32_TOL, build

pytestmark = pytest.mark.gpu

# bf16, LLaMA-2-7B widths: first decode step through the MFMA skinny GEMMs (batched) vs through the GEMVs (single conversation),
# both after the same prefill.  Every Linear layer sums in a different fp32 order, so its bf16 output flips by one ulp (2^-8
# relative) on some elements; per rounded tensor that is a relative perturbation of at most a half-ulp rms (2^-9) of the
# residual stream.  Error budget: 32 layers x 6 rounded tensors per layer (qkv, attention out, o + residual, normed copy,
# SwiGLU product, down + residual) perturb independently and add in quadrature: sqrt(192) * 2^-9 = 2.7e-2 of the stream's
# scale if every one of them flipped; the measured fraction that does is about half (1.2e-2 with bf16 weights, 1.42e-2 with fp8
# weights, round 2), and the bound takes two thirds of the all-flip budget.
BATCH_VS_SINGLE_REL = (2.0 / 3.0) * (32 * 6) ** 0.5 * 2.0 ** -9          # = 1.80e-2


def conversations(name, n, vocab):
    """n different (ids, frames) pairs: the golden conversation first, then seeded variations of different lengths."""
    g = TY.load_npz(name)
    vcfg, lcfg, mm = TY.cfgs(name)
    T = int(g["T"])
    convs = [(torch.from_numpy(g["input_ids"])[0], O.synthetic_frames(T, vcfg.image_size, seed=0))]
    for i in range(1, n):
        t = 1 + (i % 3)
        ids = O.synthetic_prompt_ids(12 + 5 * i, t, vocab, seed=10 + i)
        convs.append((ids, O.synthetic_frames(t, vcfg.image_size, seed=20 + i)))
    return g, convs


@pytest.mark.parametrize("name", ["tinyA", "tinyB", "tinyC"])
def test_batched_decode_fp32_equals_single_and_reference(name):
    model, sd = build(name, torch.float32)
    dev = model.device
    vcfg, lcfg, mm = TY.cfgs(name)
    g, convs = conversations(name, 5, lcfg.vocab_size)
    n_new = len(g["greedy_tokens"])
    singles = []
    for ids, frames in convs:
        out = model.generate(input_ids=ids.view(1, -1).to(dev), images=[f.to(dev) for f in frames], do_sample=False,
                             max_new_tokens=n_new, eos_token_id=None)
        singles.append(out[0].cpu().tolist())
    outs = model.generate_batch([ids.to(dev) for ids, _ in convs], [[f.to(dev) for f in fr] for _, fr in convs], do_sample=False,
                                max_new_tokens=n_new, eos_token_id=None)
    for b, o in enumerate(outs):
        assert o.cpu().tolist() == singles[b], f"conversation {b} differs from its single-conversation run"
    assert outs[0].cpu().tolist()[-n_new:] == g["greedy_tokens"].tolist()           # the reference's own tokens
    # conversation 0's cache slot holds what the reference's cache held
    dec = model._batch_decoder
    assert not dec.tiled
    ks = torch.from_numpy(g["kv_sel"])
    np.testing.assert_allclose(dec.k_cache[0, 0][:, ks].cpu().numpy(), g["k_layer0"], atol=FP32_TOL)
    np.testing.assert_allclose(dec.v_cache[-1, 0][:, ks].cpu().numpy(), g["v_last"], atol=FP32_TOL)
    np.testing.assert_allclose(dec.vt_cache[-1, 0][:, :, ks].transpose(1, 2).cpu().numpy(), g["v_last"], atol=FP32_TOL)
    assert float((dec.d_logits[0].cpu() - torch.from_numpy(g["greedy_logits"][-1])).abs().max()) < FP32_TOL
    # an independent check of a non-golden conversation against the CPU oracle
    ids, frames = convs[3]
    toks, _, _ = O.greedy_generate(ids.view(1, -1), frames, sd, vcfg, lcfg, mm, max_new_tokens=n_new)
    assert outs[3].cpu().tolist()[-n_new:] == toks
    # eager steps == hipGraph replays
    lg_graph = dec.d_logits.clone()
    dec.reset()
    firsts = []
    for b, (ids, frames) in enumerate(convs):
        (_, _, _, _, emb, _) = model.prepare_inputs_labels_for_multimodal(ids.view(1, -1).to(dev), None, None, None, None,
                                                                          [f.to(dev) for f in frames])
        firsts.append(int(dec.prefill(b, emb[0])[0].argmax()))
    dec.begin(firsts)
    dec.steps(n_new - 1, use_graph=False)
    assert torch.equal(lg_graph, dec.d_logits)
    assert [[f] + r for f, r in zip(firsts, dec.generated().tolist())] == [s[-n_new:] for s in singles]


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_batched_generate_api_padding_eos_and_sampling(name):
    model, _ = build(name, torch.float32)
    dev = model.device
    vcfg, lcfg, mm = TY.cfgs(name)
    g, convs = conversations(name, 3, lcfg.vocab_size)
    imgs = [[f.to(dev) for f in fr] for _, fr in convs]
    ref = model.generate_batch([ids.to(dev) for ids, _ in convs], imgs, max_new_tokens=8, eos_token_id=None)
    # HF-style call: right-padded ids + attention_mask, one image list per row
    width = max(ids.numel() for ids, _ in convs)
    ids_p = torch.zeros(3, width, dtype=torch.long)
    mask = torch.zeros(3, width, dtype=torch.long)
    for b, (ids, _) in enumerate(convs):
        ids_p[b, :ids.numel()] = ids
        mask[b, :ids.numel()] = 1
    out = model.generate(input_ids=ids_p.to(dev), attention_mask=mask.to(dev), images=imgs, max_new_tokens=8, eos_token_id=None)
    assert out.shape == (3, width + 8) and torch.equal(out[:, :width].cpu(), ids_p)      # HF: cat(input_ids, new tokens)
    for b, (ids, _) in enumerate(convs):
        assert out[b, width:].tolist() == ref[b][ids.numel():].tolist()
    # left padding (what HF batched generation expects): same tokens, the pads stay where the caller put them
    ids_l = torch.zeros(3, width, dtype=torch.long)
    mask_l = torch.zeros(3, width, dtype=torch.long)
    for b, (ids, _) in enumerate(convs):
        ids_l[b, width - ids.numel():] = ids
        mask_l[b, width - ids.numel():] = 1
    out_l = model.generate(input_ids=ids_l.to(dev), attention_mask=mask_l.to(dev), images=imgs, max_new_tokens=8, eos_token_id=None)
    assert torch.equal(out_l[:, :width].cpu(), ids_l) and torch.equal(out_l[:, width:], out[:, width:])
    # EOS: conversation 1 stops at its 3rd generated token, the others run on
    eos = int(ref[1][convs[1][0].numel() + 2])
    cut = model.generate_batch([ids.to(dev) for ids, _ in convs], imgs, max_new_tokens=8, eos_token_id=eos)
    for b in range(3):
        gen = ref[b][convs[b][0].numel():].tolist()
        want = gen[:gen.index(eos) + 1] if eos in gen else gen
        assert cut[b][convs[b][0].numel():].tolist() == want
    # sampling: reproducible under a seeded generator, each conversation with its own stream
    gen_a = torch.Generator().manual_seed(7)
    gen_b = torch.Generator().manual_seed(7)
    sa = model.generate_batch([ids.to(dev) for ids, _ in convs], imgs, do_sample=True, temperature=1.5, top_k=20, max_new_tokens=8,
                              eos_token_id=None, generator=gen_a)
    sb = model.generate_batch([ids.to(dev) for ids, _ in convs], imgs, do_sample=True, temperature=1.5, top_k=20, max_new_tokens=8,
                              eos_token_id=None, generator=gen_b)
    assert [x.tolist() for x in sa] == [x.tolist() for x in sb]
    assert [x.tolist() for x in sa] != [x.tolist() for x in ref]


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_batched_decode_bf16_mfma_path_close_to_single(name):
    """bf16: the batched step uses the tiled MFMA skinny GEMM; its logits stay within BF16_REL of the single path."""
    model, _ = build(name, torch.bfloat16)
    dev = model.device
    vcfg, lcfg, mm = TY.cfgs(name)
    g, convs = conversations(name, 4, lcfg.vocab_size)
    eng = model.engine
    single_logits, firsts = [], []
    for ids, frames in convs:
        imgs = [f.to(dev) for f in frames]
        model.generate(input_ids=ids.view(1, -1).to(dev), images=imgs, do_sample=False, max_new_tokens=3, eos_token_id=None)
        single_logits.append(eng.d_logits.clone())
    dec = model.batch_decoder(len(convs), 16)
    assert dec.tiled
    dec.reset()
    for b, (ids, frames) in enumerate(convs):
        (_, _, _, _, emb, _) = model.prepare_inputs_labels_for_multimodal(ids.view(1, -1).to(dev), None, None, None, None,
                                                                          [f.to(dev) for f in frames])
        firsts.append(int(dec.prefill(b, emb[0])[0].argmax()))
    dec.begin(firsts)
    dec.steps(2)
    worst = 0.0
    for b in range(len(convs)):
        rel = float((dec.d_logits[b] - single_logits[b]).abs().max()) / float(single_logits[b].abs().max())
        worst = max(worst, rel)
    print(f"[{name}] batched bf16 logits after 2 steps: worst rel-to-max diff vs single path {worst:.2e}")
    assert worst < BF16_REL


@pytest.mark.parametrize("weights,B", [("native", 8), ("fp8", 8), ("native", 16)])
def test_batched_decode_real_width_vs_single(weights, B):
    """LLaMA-2-7B widths, 2 layers, B = 8 / 16: tiled MFMA GEMMs at production shapes vs the single-conversation GEMV path.
    bf16 weights at B = 8 run the one-tile-per-workgroup kernel, fp8 weights and B = 16 the persistent streaming form."""
    from teochat_amd.config import LlavaConfig, VisionConfig
    from teochat_amd.engine import TeoEngine
    from teochat_amd.model import LlavaLlamaForCausalLM
    vit = dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=3, hidden_act="gelu")
    llm = dict(hidden_size=4096, num_attention_heads=32, num_key_value_heads=32, intermediate_size=11008, num_hidden_layers=2,
               vocab_size=32000)
    vcfg, lcfg, mm = O.VitCfg(**vit), O.LlamaCfg(**llm), O.MMCfg()
    sd = O.make_state_dict(vcfg, lcfg, mm, seed=2, std=0.02)
    cfg = LlavaConfig(**llm, max_position_embeddings=1024, vision_config=VisionConfig(**vit))
    eng = TeoEngine(sd, cfg, dtype=torch.bfloat16, device="cuda:0", max_seq=512, weight_format=weights)
    model = LlavaLlamaForCausalLM(cfg, eng)
    convs = [(O.synthetic_prompt_ids(20 + 3 * b, 0, 32000, seed=30 + b), []) for b in range(B)]     # text-only prompts
    single_logits, single_tokens = [], []
    for ids, _ in convs:
        out = model.generate(input_ids=ids.view(1, -1).cuda(), images=None, do_sample=False, max_new_tokens=2, eos_token_id=None)
        single_logits.append(eng.d_logits.clone())          # logits of the first decode step (input: the prefill's argmax)
        out = model.generate(input_ids=ids.view(1, -1).cuda(), images=None, do_sample=False, max_new_tokens=6, eos_token_id=None)
        single_tokens.append(out[0, -6:].tolist())
    # one batched step from the same prefill state: directly comparable logits (no token history to diverge)
    model.generate_batch([ids.cuda() for ids, _ in convs], None, do_sample=False, max_new_tokens=2, eos_token_id=None)
    dec = model._batch_decoder
    assert dec.tiled
    want_kernel = b"skinny_stream" if (weights == "fp8" or B > 8) else b"skinny_gemm"      # the lm_head GEMM ran last
    assert eng.lib.teo_last_kernel().startswith(want_kernel), eng.lib.teo_last_kernel()
    worst = 0.0
    for b in range(B):
        rel = float((dec.d_logits[b] - single_logits[b]).abs().max()) / float(single_logits[b].abs().max())
        worst = max(worst, rel)
    outs = model.generate_batch([ids.cuda() for ids, _ in convs], None, do_sample=False, max_new_tokens=6, eos_token_id=None)
    same = sum(int(outs[b][-6:].tolist() == single_tokens[b]) for b in range(B))
    print(f"real-width batched ({weights}) first-step logits worst rel diff vs single path {worst:.2e}; "
          f"identical 6-token streams {same}/{B}")
    assert worst < BATCH_VS_SINGLE_REL
    assert same >= B - 3 * (B // 8)          # random-weight logits are nearly flat: a 1-ulp difference may flip a near-tie


@pytest.mark.parametrize("name,dtype", [("tinyA", torch.float32), ("tinyB", torch.float32), ("tinyB", torch.bfloat16)])
def test_prefill_all_equals_per_slot_prefill(name, dtype):
    """teo_llama_prefill_batch (rows of all conversations concatenated) == teo_llama_prefill slot by slot: same logits,
    same K / V / V^T caches (a row's GEMM result does not depend on how many other rows share the launch)."""
    model, _ = build(name, dtype)
    dev = model.device
    vcfg, lcfg, mm = TY.cfgs(name)
    g, convs = conversations(name, 4, lcfg.vocab_size)
    dec = model.batch_decoder(len(convs), 16)
    embs = []
    for ids, frames in convs:
        (_, _, _, _, emb, _) = model.prepare_inputs_labels_for_multimodal(ids.view(1, -1).to(dev), None, None, None, None,
                                                                          [f.to(dev) for f in frames])
        embs.append(emb[0])
    dec.reset()
    one = torch.cat([dec.prefill(b, e) for b, e in enumerate(embs)])
    k1, v1, vt1 = dec.k_cache.clone(), dec.v_cache.clone(), dec.vt_cache.clone()
    for c in (dec.k_cache, dec.v_cache, dec.vt_cache):
        c.zero_()
    dec.reset()
    allb = dec.prefill_all(embs)
    assert dec.cache_len == [int(e.shape[0]) for e in embs]
    assert torch.equal(allb, one)
    for b, e in enumerate(embs):
        n = int(e.shape[0])
        assert torch.equal(dec.k_cache[:, b, :, :n], k1[:, b, :, :n]) and torch.equal(dec.v_cache[:, b, :, :n], v1[:, b, :, :n])
        assert torch.equal(dec.vt_cache[:, b, :, :, :n], vt1[:, b, :, :, :n])


@pytest.mark.parametrize("name,dtype", [("tinyA", torch.float32), ("tinyB", torch.float32), ("tinyB", torch.bfloat16)])
def test_forward_api_batched_continuation_equals_generate_batch(name, dtype):
    """An HF-style batched generate loop written against the reference API -- forward(input_ids [B, W], attention_mask, images,
    use_cache=True), then forward(input_ids [B, 1], past_key_values=<returned cache>) per step (llava_arch.py:154-163,
    llava_llama.py:101-108) -- produces generate_batch()'s greedy streams token for token, for conversations of different lengths
    (right-padded rows); the step logits equal the batched device loop's bit for bit (same kernels, same caches)."""
    model, sd = build(name, dtype)
    dev = model.device
    vcfg, lcfg, mm = TY.cfgs(name)
    g, convs = conversations(name, 4, lcfg.vocab_size)
    n_new = 6
    ids_list = [ids.to(dev) for ids, _ in convs]
    frames_list = [[f.to(dev, dtype=dtype) for f in fr] for _, fr in convs]
    want = model.generate_batch(ids_list, frames_list, do_sample=False, max_new_tokens=n_new, eos_token_id=None)
    want_new = [w[ids_list[b].numel():].tolist() for b, w in enumerate(want)]
    dec = model._batch_decoder
    want_last_logits = dec.d_logits.clone()

    B = len(convs)
    W = max(int(i.numel()) for i in ids_list)
    ids_p = torch.zeros(B, W, dtype=torch.long, device=dev)
    mask = torch.zeros(B, W, dtype=torch.long, device=dev)
    for b, i in enumerate(ids_list):
        ids_p[b, :i.numel()] = i
        mask[b, :i.numel()] = 1
    flat = [f for fr in frames_list for f in fr]
    out = model(input_ids=ids_p, attention_mask=mask, images=flat, use_cache=True)
    from teochat_amd.model import TeoBatchKVCache
    pkv = out.past_key_values
    assert isinstance(pkv, TeoBatchKVCache) and out.logits.shape[0] == B
    # last REAL position of every (right-padded) row
    (_, _, new_mask, _, _, _) = model.prepare_inputs_labels_for_multimodal(ids_p, None, mask, None, None, flat)
    last = [int(torch.nonzero(new_mask[b].to(torch.bool)).flatten()[-1]) for b in range(B)]
    assert list(pkv.decoder.cache_len) == [l + 1 for l in last] and pkv[-1][-1].shape[-2] == max(last) + 1
    # fp32: the loop picks its own tokens (what HF's greedy loop does).  bf16: the prompt logits come from the all-position lm_head
    # GEMM here and from the last-row form in generate_batch (another fp32 summation order), so a near-tie may pick another token;
    # the loop is teacher-forced with generate_batch's stream there and the statement is the bit-equality of the step logits below
    own = dtype == torch.float32
    toks = [[int(out.logits[b, last[b]].argmax()) if own else want_new[b][0]] for b in range(B)]
    for step in range(n_new - 1):
        nxt = torch.tensor([[t[-1]] for t in toks], dtype=torch.long, device=dev)
        mask = torch.cat([mask, torch.ones(B, 1, dtype=mask.dtype, device=dev)], dim=1)
        _in = model.prepare_inputs_for_generation(nxt, past_key_values=pkv, images=flat, attention_mask=mask, use_cache=True)
        out = model(**_in)
        assert out.logits.shape == (B, 1, lcfg.vocab_size) and out.past_key_values is pkv
        for b in range(B):
            toks[b].append(int(out.logits[b, 0].argmax()) if own else want_new[b][step + 1])
    assert toks == want_new
    # the loop's last step consumed token n_new - 2 of every stream, as generate_batch's last device step did: same logits, bit for bit
    assert torch.equal(out.logits[:, 0], want_last_logits)
    # errors: a one-sequence cache cannot continue a batch, and more than one new token per row is not a decode step
    one = model(input_ids=ids_list[0].view(1, -1), images=frames_list[0], use_cache=True).past_key_values
    with pytest.raises(ValueError):
        model(input_ids=nxt, past_key_values=one, attention_mask=mask)
    out2 = model(input_ids=ids_p, attention_mask=mask[:, :W], images=flat, use_cache=True)
    with pytest.raises(ValueError):
        model(input_ids=torch.cat([nxt, nxt], dim=1), past_key_values=out2.past_key_values, attention_mask=mask)
    # the training-shape forward (use_cache left None) keeps nothing
    assert model(input_ids=ids_p, attention_mask=mask[:, :W], images=flat).past_key_values is None
