#!/usr/bin/env python3
"""Every tile family on every prefill GEMM shape, cold weights (8 matrices in rotation): is the dispatch's choice the fastest?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

lib = L.load()
bf = torch.bfloat16
FAM = (("auto", {}), ("128x128", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0}), ("128x128 sk", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 2}),
       ("128x256", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 0}), ("128x256 sk", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 2}),
       ("256x256", {"gemm_big": 2, "gemm_big_hybrid": 0, "gemm_big_ragged": 2}), ("256x256 hybrid", {"gemm_big": 2, "gemm_big_hybrid": 2, "gemm_big_ragged": 2}),
       ("256x256 no-ragged", {"gemm_big": 2, "gemm_big_hybrid": 0, "gemm_big_ragged": 0}), ("256x256 hybrid no-ragged", {"gemm_big": 2, "gemm_big_hybrid": 2, "gemm_big_ragged": 0}),
       ("narrow64", {"gemm_narrow": 2, "gemm_narrow_bm": 64}), ("narrow128", {"gemm_narrow": 2, "gemm_narrow_bm": 128}), ("quad160", {"gemm_quad": 2}))
DEF = {"gemm_wide": 1, "gemm_big": 1, "gemm_sk": 1, "gemm_big_hybrid": 1, "gemm_narrow": 1, "gemm_narrow_bm": 0, "gemm_quad": 1, "gemm_big_ragged": 1}
ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws")


def timeit(fn, iters=16, warm=3):
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


Ms = [int(a) for a in sys.argv[1:]] or [2168, 638, 4208]
for M in Ms:
    for name, N, K, flags, with_res in (("qkv", 12288, 4096, 0, False), ("o", 4096, 4096, 0, True), ("gateup", 22016, 4096, L.GEMM_SWIGLU16, False),
                                        ("down", 4096, 11008, 0, True)):
        A = torch.randn(M, K, device="cuda").to(bf)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        res = torch.randn(M, Nc, device="cuda").to(bf) if with_res else None
        cnt = [0]

        def run():
            W = Ws[cnt[0] % 8]
            cnt[0] += 1
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, G.p(res), G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm")
        line = f"M={M:5d} {name:7s}"
        best = None
        for fam, knobs in FAM:
            for k, v in DEF.items():
                L.tune_set(k.encode(), v)
            for k, v in knobs.items():
                L.tune_set(k.encode(), v)
            try:
                run()
                kern = lib.teo_last_kernel().decode()
                t = min(timeit(run) for _ in range(2))
            except Exception as e:  # noqa: BLE001
                kern, t = "err", float("nan")
            line += f" | {fam} [{kern.replace('gemm_', '')}] {t:6.1f}"
            if fam != "auto" and (best is None or t < best[1]):
                best = (fam, t)
            if fam == "auto":
                auto_t = t
        for k, v in DEF.items():
            L.tune_set(k.encode(), v)
        print(line + f"  => best {best[0]} {best[1]:.1f} (auto {auto_t:.1f}, {100 * (auto_t / best[1] - 1):+.1f} %)", flush=True)
