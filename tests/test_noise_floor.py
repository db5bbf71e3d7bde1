"""The bf16 noise floor, DEMONSTRATED (VERDICT r03 "What's weak" #1): the oracle against ITSELF.

north_star asks for bf16 logits within 1e-3 of the reference CPU path.  tests/test_true_shapes_gpu.py measures 2e-2 (max) /
8e-3 (p99) / 2e-3 (median) of max|logit| between the HIP path and the oracle at C2 / C3 width, and DESIGN.md section 2 argues that
this is rounding noise: every kernel is within one bf16 ulp of the oracle on every element, and the handful of 1-ulp flips that a
different fp32 summation order produces random-walk through ~26 layers.  This file is the control for that argument: the SAME
oracle, same weights, same bf16 rounding points, run twice with only the fp32 summation order of its Linear layers changed
(oracle.K_ORDER: K accumulated in 8 chunks, ascending vs descending) disagrees with itself by the same amount.  Nothing on a GPU
is involved.  The GPU test asserts its own statistics against this self-difference (<= 1.5x).

Reference path restated by the oracle: videollava/model/language_model/llava_llama.py:56-99 via llava_arch.py:148-346."""
import time

import torch

from oracle import teo_oracle as O


def logit_stats(a, b):
    """max / p99 / median of |a - b| as fractions of max|b| (the statistics tests/test_true_shapes_gpu.py reports)."""
    d = (a.double() - b.double()).abs().flatten()
    scale = float(b.abs().max())
    k99 = max(1, int(0.99 * d.numel()))
    return float(d.max()) / scale, float(d.kthvalue(k99).values) / scale, float(d.median()) / scale, scale


def oracle_self_difference(T, n_text, n_layers, sd=None, base=None):
    """(stats, logits_a): the oracle's bf16 prefill logits with K_ORDER = (8, ascending) vs (8, descending).  `base`: reuse an
    already computed run (any K order) as one side."""
    vcfg, lcfg, mm = O.VitCfg(hidden_act="gelu", num_hidden_layers=24), O.LlamaCfg(num_hidden_layers=n_layers), O.MMCfg()
    if sd is None:
        sd = O.make_state_dict(vcfg, lcfg, mm, seed=2, std=0.02, dtype=torch.bfloat16)
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, 32000, seed=1).unsqueeze(0)
    runs = []
    try:
        for order in (((8, False), (8, True)) if base is None else ((8, True),)):
            O.K_ORDER = order
            lg, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, None, "bf16", torch.float32)
            runs.append(lg[0])
    finally:
        O.K_ORDER = None
    a = runs[0] if base is None else base
    return logit_stats(runs[-1], a), a


def test_oracle_disagrees_with_itself_by_the_bf16_noise_floor_at_c2_width():
    """C2 (T = 2, L = 638), ViT-L/14 (23 layers) + projector + 3 LLaMA layers at 7B width: max / p99 / median of the self-difference.
    Measured (this container, torch 2.10 CPU): max 1.95e-2, p99 7.9e-3, median 1.9e-3 of max|logit| 7.6 -- the HIP path sits at
    2.26e-2 / 8.9e-3 / 2.1e-3 against the same oracle (round 3, MI355X): 1.1-1.15x the oracle's own disagreement."""
    torch.set_num_threads(min(16, torch.get_num_threads()))
    t0 = time.perf_counter()
    (mx, p99, med, scale), _ = oracle_self_difference(2, 128, 3)
    print(f"\n[bf16 noise floor, oracle vs itself, C2 width, 3 LLaMA layers] K order ascending vs descending (8 chunks): "
          f"max {mx:.2e}  p99 {p99:.2e}  median {med:.2e} of max|logit| {scale:.2f};  wall {time.perf_counter() - t0:.1f} s")
    # the demonstration: a 1e-3 max-norm bound on end-to-end bf16 logits cannot hold for ANY implementation that does not reproduce the
    # reference's summation order bit for bit -- the reference restatement misses it against itself by more than 10x
    assert mx > 1e-2 and p99 > 4e-3 and med > 1e-3
    # and the floor is what DESIGN.md says it is (a regression guard on the oracle's rounding points: more noise than this would mean
    # a rounding point was added)
    assert mx < 4e-2 and p99 < 1.2e-2 and med < 3e-3
