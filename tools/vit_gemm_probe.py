#!/usr/bin/env python3
"""The four GEMMs of a ViT-L/14 encoder layer at T = 8 (M = 2056) and the projector (M = 2048): every tile family, weights of 8 layers in
rotation.  (Round 4 also ran it with an n-fastest tile order and an in-workgroup split-K kernel; both lost -- profiles/r04_vit_gemm_probe.txt.)  usage: python tools/vit_gemm_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

lib = L.load()
bf = torch.bfloat16
FAM = (("auto", {}), ("plain", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0}),
       ("plain 128 rows", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0, "gemm_bm": 128}),
       ("128 sk", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 2}), ("128x256", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 0}),
       ("128x256 sk", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 2}), ("256x256", {"gemm_big": 2, "gemm_big_hybrid": 0}),
       ("256x256 hybrid", {"gemm_big": 2, "gemm_big_hybrid": 2}),
       ("narrow 64", {"gemm_narrow": 2, "gemm_narrow_bm": 64}), ("narrow 128", {"gemm_narrow": 2, "gemm_narrow_bm": 128}), ("narrow 128 w8", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_waves": 8}),
       ("no narrow", {"gemm_narrow": 0}),
       ("quad 256x160", {"gemm_quad": 2}), ("quad 256x160 four waves", {"gemm_quad": 2, "gemm_quad_waves": 4}), ("no quad", {"gemm_quad": 0}))
ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws")


def timeit(fn, iters=24, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, M, N, K, act, with_res in (("vit qkv", 2056, 3072, 1024, 0, False), ("vit out", 2056, 1024, 1024, 0, True),
                                     ("vit fc1", 2056, 4096, 1024, L.ACT_GELU_ERF, False), ("vit fc2", 2056, 1024, 4096, 0, True),
                                     ("proj 1", 2048, 4096, 1024, L.ACT_GELU_ERF, False), ("proj 2", 2048, 4096, 4096, 0, False),
                                     ("vit qkv T2", 514, 3072, 1024, 0, False), ("vit out T2", 514, 1024, 1024, 0, True), ("vit fc1 T2", 514, 4096, 1024, L.ACT_GELU_ERF, False),
                                     ("vit fc2 T2", 514, 1024, 4096, 0, True), ("vit fc2 T16", 4112, 1024, 4096, 0, True), ("vit out T16", 4112, 1024, 1024, 0, True),
                                     ("vit fc1 T16", 4112, 4096, 1024, L.ACT_GELU_ERF, False), ("vit qkv T16", 4112, 3072, 1024, 0, False),
                                     ("llama o", 2168, 4096, 4096, 0, True), ("llama down", 2168, 4096, 11008, 0, True), ("llama o T2", 638, 4096, 4096, 0, True), ("llama down T2", 638, 4096, 11008, 0, True),
                                     ("llama qkv T2", 638, 12288, 4096, 0, False), ("lm_head-like", 638, 32000, 4096, 0, False),
                                     ("llama o T3", 893, 4096, 4096, 0, True), ("llama down T3", 893, 4096, 11008, 0, True), ("llama o T4-", 1022, 4096, 4096, 0, True),
                                     ("llama down T4-", 1022, 4096, 11008, 0, True), ("llama o T5", 1403, 4096, 4096, 0, True), ("llama down T5", 1403, 4096, 11008, 0, True),
                                     ("llama o T1", 383, 4096, 4096, 0, True), ("llama down T1", 383, 4096, 11008, 0, True)):
    A = torch.randn(M, K, device="cuda").to(bf)
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
    bias = torch.randn(N, device="cuda").to(bf)
    Cc = torch.empty(M, N, dtype=bf, device="cuda")
    res = torch.randn(M, N, device="cuda").to(bf) if with_res else None
    cnt = [0]

    def run():
        W = Ws[cnt[0] % 8]
        cnt[0] += 1
        L.check(lib.teo_gemm_ws(G.p(A), G.p(W), G.p(bias), G.p(res), G.p(Cc), M, N, K, K, N, act, 0, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm")
    line = f"{name:8s} M={M} N={N} K={K} ({2.0 * M * N * K / 1e9:5.1f} GF)"
    ref = None
    for fam, knobs in FAM:
        L.tune_reset()
        for k, v in knobs.items():
            L.tune_set(k.encode(), v)
        try:
            cnt[0] = 0
            run()
            out = Cc.clone()
            ref = out if ref is None else ref
            kern = lib.teo_last_kernel().decode().replace("gemm_", "")
            t = min(timeit(run) for _ in range(2))
            line += f" | {fam} [{kern}] {t:5.1f}{'' if torch.equal(out, ref) else ' DIFF'}"
        except Exception as e:  # noqa: BLE001
            line += f" | {fam} err"
    L.tune_reset()
    print(line, flush=True)
