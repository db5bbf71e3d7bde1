"""A synthetic checkpoint with the statistics of a TRAINED LLaMA-2-class model (teochat_amd/synthetic.py realistic=True: heavy-tailed
weights, massive-activation channels 10^2-10^3 x the median of the residual stream, non-unit norm gains) through the same parity
and quantisation checks as the N(0, 0.02^2) one.  No real TEOChat weights exist in this environment (VERDICT r03 "What's missing" #3;
what a real checkpoint goes through: videollava/model/builder.py:94-112).

  (a) C2 (T = 2, L = 638), ViT-L/14 + projector + 3 LLaMA layers at 7B width, bf16: HIP path vs the oracle at the same rounding points,
      next to the oracle's own self-difference (K order changed) -- the parity statement of tests/test_true_shapes_gpu.py on weights
      that exercise range;
  (b) the fp8 weight path on it: what the power-of-two e4m3 ROW scales cost when one 10-sigma outlier sets a row's scale (fp8-weight
      logits vs bf16-weight logits), and what per-token e4m3 ACTIVATION quantisation (w8a8 prefill) costs when a token carries a
      600 x outlier channel -- both next to the same numbers on the Gaussian checkpoint."""
import time

import pytest
import torch

from oracle import teo_oracle as O
from teochat_amd.synthetic import MASSIVE_CHANNELS, synthetic_state_dict
from tests.test_true_shapes_gpu import DEV, N_LAYERS_DEEP, _oracle_cfgs, _stats, _threads

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _model(sd, cfg, weight_format=None, max_seq=1024, dtype=bf):
    from teochat_amd.engine import TeoEngine
    from teochat_amd.model import LlavaLlamaForCausalLM
    eng = TeoEngine(sd, cfg, dtype=dtype, device=DEV, max_seq=max_seq, weight_format=weight_format)
    return LlavaLlamaForCausalLM(cfg, eng)


def _cfg():
    from teochat_amd.config import teochat_7b_config
    cfg = teochat_7b_config()
    cfg.num_hidden_layers = N_LAYERS_DEEP
    return cfg


def test_realistic_checkpoint_has_the_statistics_it_claims():
    cfg = _cfg()
    sd = synthetic_state_dict(cfg, seed=2, dtype=bf, device=DEV, realistic=True)
    w = sd["model.layers.0.mlp.gate_proj.weight"].float()
    kurt = float(((w / w.std()) ** 4).mean())
    row_peak = float((w.abs().amax(1) / w.std()).median())
    assert 0.018 < float(w.std()) < 0.022 and kurt > 6.0 and row_peak > 7.0, (float(w.std()), kurt, row_peak)
    # residual stream after the massive layer, through the oracle on 64 tokens
    lcfg = O.LlamaCfg(num_hidden_layers=N_LAYERS_DEEP)
    sdc = {k: v.cpu() for k, v in sd.items() if k.startswith("model.layers.") or k.startswith("model.embed")}
    S = 64
    ids = torch.randint(3, 32000, (1, S), generator=torch.Generator().manual_seed(0))
    R = O._rounder("bf16")
    h = R(sdc["model.embed_tokens.weight"].float()[ids])
    cos, sin = O.rope_cos_sin(torch.arange(S).unsqueeze(0), lcfg.head_dim, lcfg.rope_theta, torch.float32)
    vis = (torch.arange(S).view(1, S) <= torch.arange(S).view(S, 1)).view(1, 1, S, S)
    cache = O.KVCache()
    for i in range(2):
        h = O.llama_layer(h, i, sdc, lcfg, cos, sin, vis, cache, R, "exact")
    a = h.abs()
    ratio = float(a[..., list(MASSIVE_CHANNELS)].mean() / a.median())
    print(f"\n[realistic checkpoint] weights: std {float(w.std()):.4f}, kurtosis {kurt:.1f}, median row absmax {row_peak:.1f} sigma;  residual stream after "
          f"layer 1: median |h| {float(a.median()):.3f}, massive channels {float(a[..., list(MASSIVE_CHANNELS)].mean()):.0f} ({ratio:.0f} x the median)")
    assert 100.0 < ratio < 5000.0


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_realistic_checkpoint_c2_against_the_oracle_and_its_noise_floor(fmt):
    """bf16 and -- round 5 -- fp16, the reference's own inference type (model/builder.py:104-105, eval/inference.py:53), whose largest
    finite value is 65 504: the massive-activation channels (190-600 x the median of the residual stream) must stay finite through every
    snapshot of the residual stream (forward(output_hidden_states=True)); where the format itself overflows, the oracle -- the reference's
    semantics at the same rounding points -- must overflow at exactly the same places (no kernel-made inf / NaN)."""
    _threads()
    t0 = time.perf_counter()
    h16 = torch.float16 if fmt == "fp16" else bf
    T, n_text = 2, 128
    cfg = _cfg()
    vcfg, lcfg, mm = _oracle_cfgs(N_LAYERS_DEEP)
    sd_dev = synthetic_state_dict(cfg, seed=2, dtype=h16, device=DEV, realistic=True)
    m = _model(sd_dev, cfg, dtype=h16)
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, 32000, seed=1).unsqueeze(0)
    out = m(input_ids=ids.to(DEV), images=[f.to(DEV, dtype=h16) for f in frames], output_hidden_states=True)
    got = out.logits[0].float().cpu()
    got_hs = torch.stack(out.hidden_states)[:, 0].float().cpu()
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del m, sd_dev, out
    torch.cuda.empty_cache()
    want_hs = []
    want, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, None, fmt, torch.float32, hidden_states=want_hs)
    want_hs = torch.stack(want_hs)[:, 0]
    try:
        O.K_ORDER = (8, True)
        self_, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, None, fmt, torch.float32)
    finally:
        O.K_ORDER = None
    mx, p99, med, sc = _stats(got, want[0])
    smx, sp99, smed, _ = _stats(self_[0], want[0])
    agree = float((got.argmax(-1) == want[0].argmax(-1)).float().mean())
    peak = float(want_hs[1:-1].abs().max())
    print(f"\n[realistic checkpoint, C2, {fmt}, {N_LAYERS_DEEP} LLaMA layers at 7B width] HIP vs oracle: max {mx:.2e}  p99 {p99:.2e}  median {med:.2e} of "
          f"max|logit| {sc:.2f} (argmax agreement {agree * 100:.1f} %);  oracle vs itself (K order): max {smx:.2e}  p99 {sp99:.2e}  median {smed:.2e};  "
          f"HIP / self: {mx / smx:.2f}x / {p99 / sp99:.2f}x / {med / smed:.2f}x;  largest |residual stream| {peak:.0f} "
          f"({'%.1f %% of the fp16 range' % (peak / 65504 * 100) if fmt == 'fp16' else 'bf16 range is 3e38'});  wall {time.perf_counter() - t0:.1f} s")
    # no inf / NaN anywhere in the residual stream or the logits -- and wherever the ORACLE is non-finite (the format's own overflow), so are we
    assert torch.equal(torch.isfinite(got_hs), torch.isfinite(want_hs)), "non-finite residual-stream entries differ from the oracle's"
    assert bool(torch.isfinite(want_hs).all()) and bool(torch.isfinite(got_hs).all()) and bool(torch.isfinite(got).all())
    # the snapshots themselves are the oracle's (16-bit noise apart): the massive channels are where range would go wrong first
    rel = float((got_hs - want_hs).abs().max()) / float(want_hs.abs().max())
    assert rel < (3e-2 if fmt == "bf16" else 4e-3), rel
    assert mx <= 1.5 * smx and p99 <= 1.5 * sp99 and med <= 1.5 * smed, (mx, smx, p99, sp99, med, smed)


def test_fp8_weight_and_w8a8_activation_quantisation_on_realistic_vs_gaussian_statistics():
    """What quantisation costs on each checkpoint, at C2, logits of every position, as fractions of max|logit|:
         weights  : fp8-e4m3 rows with power-of-two scales (decode weight stream; prefill on the exactly dequantised bf16 copies) vs bf16 weights
         + w8a8   : the same fp8 weights with per-token e4m3 activations in the prefill GEMMs vs the exact prefill on those weights."""
    T, n_text = 2, 128
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, 32000, seed=1).unsqueeze(0)
    rows = {}
    for name, realistic in (("gaussian", False), ("realistic", True)):
        cfg = _cfg()
        sd = synthetic_state_dict(cfg, seed=2, dtype=bf, device=DEV, realistic=realistic)
        imgs = [f.to(DEV, dtype=bf) for f in frames]
        m = _model(sd, cfg)
        base = m(input_ids=ids.to(DEV), images=imgs).logits[0].float().cpu()
        del m
        torch.cuda.empty_cache()
        m8 = _model(sd, cfg, weight_format="fp8")
        q_exact = m8(input_ids=ids.to(DEV), images=imgs).logits[0].float().cpu()
        m8.engine.set_options(prefill_fp8=True)
        q_w8a8 = m8(input_ids=ids.to(DEV), images=imgs).logits[0].float().cpu()
        m8.engine.set_options(prefill_fp8=False)
        del m8, sd
        torch.cuda.empty_cache()
        assert bool(torch.isfinite(q_exact).all()) and bool(torch.isfinite(q_w8a8).all())
        wq = _stats(q_exact, base)
        aq = _stats(q_w8a8, q_exact)
        rows[name] = (wq, aq, float((q_exact.argmax(-1) == base.argmax(-1)).float().mean()), float((q_w8a8.argmax(-1) == q_exact.argmax(-1)).float().mean()))
    print("\n[quantisation cost at C2, 3 LLaMA layers at 7B width; max / p99 / median of |d| / max|logit|]")
    for name, (wq, aq, ag_w, ag_a) in rows.items():
        print(f"  {name:9s}: fp8 weights vs bf16 weights: {wq[0]:.2e} / {wq[1]:.2e} / {wq[2]:.2e} (argmax agreement {ag_w * 100:.1f} %);   "
              f"+ w8a8 prefill vs exact prefill: {aq[0]:.2e} / {aq[1]:.2e} / {aq[2]:.2e} (argmax agreement {ag_a * 100:.1f} %)")
    # sanity bounds (measured values are printed and copied into BASELINE.md): quantisation noise stays a perturbation, not a failure
    for name, (wq, aq, ag_w, ag_a) in rows.items():
        assert wq[2] < 0.05 and aq[2] < 0.05, (name, wq, aq)
