"""BASELINE.json's three big configurations, end to end through `generate()` at the full model size (LLaMA-2-7B + CLIP-ViT-L/14
shapes, 32 + 23 layers, synthetic weights), each under size-independent checks (the CPU oracle cannot run these sizes in
seconds; VERDICT r01 "configs_untested"):

  C3  T=8 frames, 128-token prompt -> L=2168, 256 new tokens, bf16, one GPU
  C4  T=16 frames -> L=4208 (full length, beyond LLaMA-2's 4096 positions) and the reference's truncation mode
      (`tokenizer_model_max_length=3072`, llava_arch.py:295-299); the frame-sharded tower of C4 is tests/test_shard_frames_gpu.py
  C5  B=8 conversations x T=8 frames, fp8-e4m3 decode weights, 32 layers, batched decode

Checks (bit-exact unless a tolerance is stated):
  determinism            the same conversation twice -> identical token streams
  teacher-forced decode  every token the device-resident decode loop produced is re-derived by ONE prefill over
                         prompt + generated tokens (different kernels: MFMA GEMM + flash attention instead of GEMV + split-KV
                         decode attention): where the prefill's top-2 margin exceeds the bf16 noise bound the tokens must agree,
                         and the last decode step's logits must match the prefill's within PREFILL_DECODE_REL of max|logit|
  frame locality         changing the LAST frame leaves every logit before its splice position bit-identical
  truncation == prefix   logits of the truncated run equal the first rows of the untruncated run bit for bit (causality)
  batched == single      C5: first tokens of the batch equal the single-conversation ones; the batched streams pass the
                         teacher-forced check against single-conversation prefills
"""
import pytest
import torch

from oracle import teo_oracle as O

pytestmark = pytest.mark.gpu

# bf16 prefill (MFMA GEMM + flash attention) vs bf16 decode (GEMV + split-KV attention) on a 32-layer stack of random
# weights: measured 2.6e-2 (C3, ctx 2423) / 2.8e-2 (C4, ctx 4255) of max|logit| on the last step's logits (round 2,
# gpurun_out/r02/a1.log) -- random-walk accumulation of ~220 bf16 roundings; bound = measured + 40 %.  The per-kernel
# statement (<= 1 bf16 ulp on every element) is tests/test_bf16_walk_gpu.py.
PREFILL_DECODE_REL = 4e-2
VOCAB = 32000


def _load(max_seq, weight_format=None):
    from teochat_amd.builder import load_pretrained_model
    _, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device="cuda:0",
                                           dtype=torch.bfloat16, max_seq=max_seq, weight_format=weight_format)
    return model


@pytest.fixture(scope="module")
def model_long():
    m = _load(4608)
    yield m
    del m
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def model_fp8():
    m = _load(2560, "fp8")
    yield m
    del m
    torch.cuda.empty_cache()


def conversation(T, n_text, seed):
    frames = [f.to("cuda:0", dtype=torch.bfloat16) for f in O.synthetic_frames(T, 224, seed=seed)]
    ids = O.synthetic_prompt_ids(n_text, T, VOCAB, seed=seed + 1).view(1, -1).cuda()
    return frames, ids


def teacher_forced_check(m, ids, frames, stream, last_step_logits=None, tag=""):
    """One prefill over prompt + stream[:-1]; row (L-1+i) must predict stream[i] wherever its top-2 margin is above the
    noise bound.  Returns (agreeing tokens, tokens with a decisive margin, rel diff of the last step's logits)."""
    n = len(stream)
    full_ids = torch.cat([ids, torch.tensor([stream[:-1]], dtype=ids.dtype, device=ids.device)], dim=1) if n > 1 else ids
    logits = m(input_ids=full_ids, images=frames).logits[0]
    L = logits.shape[0] - (n - 1)
    rows = logits[L - 1:]                                        # [n, V]
    top2 = torch.topk(rows, 2, dim=-1)
    margin = (top2.values[:, 0] - top2.values[:, 1])
    scale = rows.abs().amax(dim=-1)
    decisive = margin > 2.0 * PREFILL_DECODE_REL * scale         # both paths may be off by REL each
    st = torch.tensor(stream, device=rows.device)
    agree = top2.indices[:, 0] == st
    n_dec = int(decisive.sum())
    assert bool(agree[decisive].all()), f"{tag}: decode tokens differ from the prefill argmax at decisive positions " \
                                        f"{(decisive & ~agree).nonzero().flatten().tolist()}"
    # the chosen token must at least be a near-top candidate of the prefill row everywhere
    chosen = rows.gather(1, st.view(-1, 1)).flatten()
    assert bool(((top2.values[:, 0] - chosen) <= 2.0 * PREFILL_DECODE_REL * scale).all()), f"{tag}: a decode token is far from the prefill top"
    rel = None
    if last_step_logits is not None:
        rel = float((rows[-1] - last_step_logits).abs().max()) / float(rows[-1].abs().max())
        assert rel < PREFILL_DECODE_REL, (tag, rel)
    print(f"[{tag}] L={L} new={n}: decode==prefill argmax at {int(agree.sum())}/{n} positions "
          f"({n_dec} decisive, all agree); last-step logits rel diff {rel}")
    return int(agree.sum()), n_dec, rel


# ------------------------------------------------------------------------------------------------------------ C3
def test_c3_generate_256_deterministic_and_consistent_with_prefill(model_long):
    m = model_long
    frames, ids = conversation(8, 128, seed=0)
    out = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=256, eos_token_id=None)
    last_logits = m.engine.d_logits.clone()                   # logits of the final decode step (chose token 256)
    assert out.shape == (1, 128 + 256)
    assert torch.equal(out[:, :128], ids)
    stream = out[0, 128:].tolist()
    assert all(0 <= t < VOCAB for t in stream)
    assert m.engine.cache_len == 2168 + 255                   # L = 128 - 8 + 8*256, the last token is never fed back
    again = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=256, eos_token_id=None)
    assert torch.equal(out, again)
    chunked = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=256, eos_token_id=None, chunk=256)
    assert torch.equal(out, chunked)                          # host look-ahead granularity does not change the stream
    agree, n_dec, rel = teacher_forced_check(m, ids, frames, stream, last_logits, tag="C3")
    assert agree >= 0.5 * len(stream)


def test_c3_frame_locality_at_full_length(model_long):
    m = model_long
    frames, ids = conversation(8, 128, seed=2)
    a = m(input_ids=ids, images=frames).logits[0]
    assert a.shape[0] == 2168 and bool(torch.isfinite(a).all())
    pos = (ids[0] == -200).nonzero().flatten().tolist()
    start_last = pos[7] + 7 * 255                             # row where frame 7's 256 tokens begin
    frames2 = frames[:7] + [(frames[7] * 0.5).contiguous()]
    b = m(input_ids=ids, images=frames2).logits[0]
    assert torch.equal(a[:start_last], b[:start_last])
    assert not torch.equal(a[start_last:], b[start_last:])
    # changing frame 0 instead changes (almost) everything after its first token but nothing before it
    frames3 = [(frames[0] * 0.5).contiguous()] + frames[1:]
    c = m(input_ids=ids, images=frames3).logits[0]
    assert torch.equal(a[:pos[0]], c[:pos[0]]) and not torch.equal(a[pos[0]:], c[pos[0]:])


# ------------------------------------------------------------------------------------------------------------ C4
def test_c4_full_length_generate_beyond_4096_positions(model_long):
    m = model_long
    frames, ids = conversation(16, 128, seed=4)
    out = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=48, eos_token_id=None)
    last_logits = m.engine.d_logits.clone()
    assert m.engine.cache_len == 4208 + 47                    # L = 128 - 16 + 16*256 = 4208 > LLaMA-2's 4096 positions
    again = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=48, eos_token_id=None)
    assert torch.equal(out, again)
    teacher_forced_check(m, ids, frames, out[0, 128:].tolist(), last_logits, tag="C4 full length")


def test_c4_truncation_mode_is_a_prefix_of_the_full_run(model_long):
    """llava_arch.py:295-299: with tokenizer_model_max_length set, the spliced sequence is cut to that length."""
    m = model_long
    frames, ids = conversation(16, 128, seed=6)
    full = m(input_ids=ids, images=frames).logits[0]
    assert full.shape[0] == 4208
    m.config.tokenizer_model_max_length = 3072
    try:
        (_, pos, mask, _, emb, _) = m.prepare_inputs_labels_for_multimodal(ids, None, torch.ones_like(ids), None, None, frames)
        assert emb.shape[1] == 3072 and mask.shape[1] == 3072 and bool(mask.all())
        cut = m(input_ids=ids, images=frames).logits[0]
        assert cut.shape[0] == 3072
        assert torch.equal(cut, full[:3072])
        out = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=8, eos_token_id=None)
        assert m.engine.cache_len == 3072 + 7
        assert int(out[0, 128]) == int(cut[-1].argmax())      # first generated token comes from the truncated prompt
    finally:
        m.config.tokenizer_model_max_length = None


# ------------------------------------------------------------------------------------------------------------ C5
def test_c5_batch8_fp8_32_layers_against_single_conversations(model_fp8):
    m = model_fp8
    B, n_new = 8, 24
    convs = [conversation(8, 128, seed=10 + 2 * b) for b in range(B)]
    outs = m.generate_batch([ids[0] for _, ids in convs], [fr for fr, _ in convs], do_sample=False, max_new_tokens=n_new,
                            eos_token_id=None)
    assert len(outs) == B and all(o.numel() == 128 + n_new for o in outs)
    again = m.generate_batch([ids[0] for _, ids in convs], [fr for fr, _ in convs], do_sample=False, max_new_tokens=n_new,
                             eos_token_id=None)
    assert all(torch.equal(a, b) for a, b in zip(outs, again))
    same_stream = 0
    for b, (frames, ids) in enumerate(convs):
        single = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=n_new, eos_token_id=None)
        assert m.engine.cache_len == 2168 + n_new - 1
        # the first token comes from prefill kernels that are bitwise equal in both paths (teo_llama_prefill_batch)
        assert int(outs[b][128]) == int(single[0, 128]), b
        same_stream += int(torch.equal(outs[b], single[0]))
        teacher_forced_check(m, ids, frames, outs[b][128:].tolist(), tag=f"C5 conversation {b} (batched, fp8)")
    print(f"C5: {same_stream}/{B} batched streams identical to the single-conversation streams")
    assert same_stream >= 2          # measured 4/8: near-tie flips between the GEMV and the skinny-GEMM summation orders; the
                                     # teacher-forced check above is what guards every token


def test_c5_w8a8_prefill_on_the_fp8_mfma(model_fp8):
    """Config C5's "fp8 weight path on CDNA4 MFMA" for the PREFILL phase (tune prefill_fp8 = 1): every Linear layer of the 32
    decoder layers runs as a w8a8 GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 (activations quantised per token to e4m3, the
    decode path's per-row e4m3 weights).  Against the exact path (bf16 MFMA on the dequantised weights = the same weights):
      * per-token activation quantisation is lossy (3 mantissa bits): the logits move by a few percent of max|logit| on this
        random-weight model -- bound below = measured + margin, reported;
      * determinism, and the KV cache it leaves behind lets the decode loop continue (tokens are produced, finite logits);
      * switching it off again restores the exact path bit for bit."""
    from teochat_amd import _lib as L
    m = model_fp8
    lib = m.engine.lib
    frames, ids = conversation(8, 128, seed=30)
    exact = m(input_ids=ids, images=frames).logits[0]
    assert lib.teo_tune_set(b"prefill_fp8", 1) == 0
    try:
        q1 = m(input_ids=ids, images=frames).logits[0]
        q2 = m(input_ids=ids, images=frames).logits[0]
        assert torch.equal(q1, q2) and bool(torch.isfinite(q1).all())
        assert not torch.equal(q1, exact)                      # the fp8 path really ran
        rel = float((q1 - exact).abs().max()) / float(exact.abs().max())
        rms = float((q1 - exact).pow(2).mean().sqrt()) / float(exact.pow(2).mean().sqrt())
        agree = float((q1.argmax(-1) == exact.argmax(-1)).float().mean())
        print(f"w8a8 prefill vs exact: max rel-to-max {rel:.3f}, rms ratio {rms:.3f}, argmax agreement {agree:.3f}")
        assert rms < 0.35 and rel < 0.6
        out = m.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=8, eos_token_id=None)
        assert out.shape[1] == 128 + 8 and m.engine.cache_len == 2168 + 7
        assert int(out[0, 128]) == int(q1[-1].argmax())
    finally:
        lib.teo_tune_set(b"prefill_fp8", 0)
    assert torch.equal(m(input_ids=ids, images=frames).logits[0], exact)
