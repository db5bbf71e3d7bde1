"""The overlapped decode step (launch chain: csrc/common.h, runtime.hip llama_decode_chain_steps) against the round-2 step.

The chain launches the step's kernels with the AQL barrier bit cleared and hands over through a device-side progress word, with
write-through stores / coherent loads for everything that crosses a kernel; its kernels (ck_gemv, fat-split attention) are new,
so the arithmetic is checked against the old graph-replay step AND against the oracle:
  * mid-size model (D=2048, 16 heads x 128, F=5632, 3 layers; the chain needs K >= 2048) on the anchored synthetic checkpoint:
    token streams of generate() identical with the chain on and off, last-step logits within the bf16 bound, repeated calls,
    chunked calls, sampling with a fixed seed, fp8 weights;
  * the oracle's greedy tokens (bf16 boundaries) at every decisive position;
  * the chain's error word stays 0 (a wait that gives up raises in TeoEngine.decode_steps).
The chain is OFF by default (tune "decode_chain"): measured slower than the graph-replay step; these tests keep it
correct."""
import pytest
import torch

from oracle import teo_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(weight_format=None, seed=3):
    from teochat_amd.config import LlavaConfig, VisionConfig
    from teochat_amd.engine import TeoEngine
    from teochat_amd.model import LlavaLlamaForCausalLM
    from teochat_amd.synthetic import synthetic_state_dict
    cfg = LlavaConfig(hidden_size=2048, num_attention_heads=16, num_key_value_heads=16, intermediate_size=5632, num_hidden_layers=3,
                      vocab_size=4096, mm_hidden_size=128, max_position_embeddings=2048,
                      vision_config=VisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                                                 hidden_act="gelu"))
    sd = synthetic_state_dict(cfg, seed=seed, std=0.02, dtype=torch.bfloat16, device=DEV, anchored=True)
    eng = TeoEngine(sd, cfg, dtype=torch.bfloat16, device=DEV, max_seq=1024, weight_format=weight_format)
    return LlavaLlamaForCausalLM(cfg, eng), sd, cfg


def _ids(n, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 4096, (1, n), generator=g)
    ids[0, 0] = 1
    return ids.to(DEV)


@pytest.mark.parametrize("weight_format", [None, "fp8"])
def test_chain_streams_equal_the_round2_step(weight_format):
    m, sd, cfg = _model(weight_format)
    lib = m.engine.lib
    assert lib.teo_tune_set(b"decode_chain", 1) == 0 and lib.teo_tune_set(b"attn_fat", 1) == 0       # the chain uses the fat-split attention
    if weight_format == "fp8" and lib.teo_llama_decode_chain_supported(m.engine.llama_desc) != 1:
        pytest.skip("fp8 rows of K = 2048 are shorter than one chain prefetch block (the 7B shapes are covered)")
    assert lib.teo_llama_decode_chain_supported(m.engine.llama_desc) == 1
    ids = _ids(300, 7)
    outs, logits = {}, {}
    for chain in (1, 0, 1):
        assert lib.teo_tune_set(b"decode_chain", chain) == 0 and lib.teo_tune_set(b"attn_fat", chain) == 0     # 0: the round-2 step as shipped
        try:
            o = m.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=64, eos_token_id=None)
            lg = m.engine.d_logits.clone()
            o2 = m.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=64, eos_token_id=None, chunk=7)    # ragged chunks
        finally:
            lib.teo_tune_set(b"decode_chain", 0)
            lib.teo_tune_set(b"attn_fat", 0)
        assert torch.equal(o, o2), f"chain={chain}: chunking changed the stream"
        if chain in outs:
            assert torch.equal(outs[chain], o) and torch.equal(logits[chain], lg)                # deterministic across calls
        outs[chain], logits[chain] = o, lg
    same = (outs[1][0, 300:] == outs[0][0, 300:]).tolist()
    first_diff = same.index(False) if False in same else len(same)
    rel = float((logits[1] - logits[0]).abs().max()) / float(logits[0].abs().max())
    print(f"chain vs round-2 step ({weight_format or 'bf16'} weights): streams equal for {first_diff}/64 tokens, "
          f"{len(set(outs[1][0, 300:].tolist()))} distinct tokens; last-step logits rel diff {rel:.2e}")
    assert first_diff == 64, "the overlapped step produced a different token stream"
    assert rel < 2e-2


@pytest.fixture
def chain_on():
    from teochat_amd import _lib as L
    lib = L.load()
    assert lib.teo_tune_set(b"decode_chain", 1) == 0 and lib.teo_tune_set(b"attn_fat", 1) == 0
    yield lib
    lib.teo_tune_set(b"decode_chain", 0)
    lib.teo_tune_set(b"attn_fat", 0)


def test_chain_sampling_and_stop_ids(chain_on):
    m, sd, cfg = _model()
    assert chain_on.teo_llama_decode_chain_supported(m.engine.llama_desc) == 1
    ids = _ids(64, 11)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    a = m.generate(input_ids=ids, images=None, do_sample=True, temperature=0.7, top_k=20, max_new_tokens=24, eos_token_id=None, generator=g1)
    b = m.generate(input_ids=ids, images=None, do_sample=True, temperature=0.7, top_k=20, max_new_tokens=24, eos_token_id=None, generator=g2)
    assert torch.equal(a, b)
    greedy = m.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=24, eos_token_id=None)[0, 64:].tolist()
    stop = next((t for t in greedy[1:] if t != greedy[0]), None)
    if stop is not None:
        cut = m.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=24, eos_token_id=stop)[0, 64:].tolist()
        assert cut == greedy[:greedy.index(stop) + 1]


def test_chain_tokens_against_the_oracle(chain_on):
    """Greedy tokens of the chained device loop vs the oracle (bf16 boundaries, same bf16 weights) at decisive positions."""
    from teochat_amd.synthetic import anchor_gains
    m, sd_dev, cfg = _model()
    lcfg = O.LlamaCfg(hidden_size=2048, num_attention_heads=16, num_key_value_heads=16, intermediate_size=5632, num_hidden_layers=3,
                      vocab_size=4096)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    ids = _ids(200, 13)
    n_new = 24
    got = m.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=n_new, eos_token_id=None)[0, 200:].tolist()
    emb_w = sd["model.embed_tokens.weight"].float()
    logits, cache = O.llama_forward(emb_w[ids.cpu()], None, None, None, sd, lcfg, "bf16", last_only=True)
    gains = anchor_gains(4096)
    n_dec = 0
    for i, t in enumerate(got):                                   # teacher-forced on the device's stream
        row = logits[0, -1]
        top2 = torch.topk(row, 2)
        sigma = float(row[gains == 1.0].std())
        decisive = float(top2.values[0] - top2.values[1]) > 0.12 * sigma * float(gains[top2.indices[0]] + gains[top2.indices[1]])
        if decisive:
            n_dec += 1
            assert int(top2.indices[0]) == t, f"step {i}: device token {t}, oracle {int(top2.indices[0])}"
        logits, cache = O.llama_forward(emb_w[torch.tensor([[t]])], None, None, cache, sd, lcfg, "bf16", decode_kernel=True)
    print(f"chain vs oracle: {n_dec}/{n_new} positions decisive, all equal")
    assert n_dec >= n_new // 2
