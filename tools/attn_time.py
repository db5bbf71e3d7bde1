#!/usr/bin/env python3
"""Prefill / tower attention alone (teo_attention on the flash kernel): microseconds per launch at the model's shapes, 20 launches back to back.
usage (GPU box): python tools/attn_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpu as G  # noqa: E402


def us(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, B, H, S, D, causal in (("llama C2 L=638", 1, 32, 638, 128, True), ("llama C3 L=2168", 1, 32, 2168, 128, True), ("llama C4 L=4208", 1, 32, 4208, 128, True),
                                ("tower T=2", 2, 16, 257, 64, False), ("tower T=8", 8, 16, 257, 64, False), ("tower T=16", 16, 16, 257, 64, False)):
    g = torch.Generator().manual_seed(S)
    q, k, v = (torch.randn(B, H, S, D, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    vt = G.make_vt(v)
    t = min(us(lambda: G.attention(q, k, v, causal, D ** -0.5, vt=vt)) for _ in range(3))
    flops = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    print(f"{name:18s} [{G.lib().teo_last_kernel().decode()}] {t:7.1f} us  {flops / t / 1e6:6.0f} TFLOP/s")
