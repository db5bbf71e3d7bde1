#!/usr/bin/env python3
"""Where does the 256 x 160 four-wave tile (gemm_quad.hip) beat the dispatch's other choices?  Every LLaMA-2-7B linear layer at the
prefill lengths of configs C2 / C3 / C4 and a few between, weights in rotation (cold), residual epilogue where the model has one.
usage: python tools/quad_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

lib = L.load()
bf = torch.bfloat16
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


for M in (638, 1100, 1400, 1800, 2168, 2304, 2600, 3100, 4208):
    for name, N, K, with_res in (("qkv", 12288, 4096, False), ("o", 4096, 4096, True), ("down", 4096, 11008, True), ("lm_head", 32000, 4096, False)):
        if name == "lm_head" and M > 700:
            continue
        n = max(2, int(600e6 // (N * K * 2)))
        A = torch.randn(M, K, device="cuda").to(bf)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(n)]
        Cc = torch.empty(M, N, dtype=bf, device="cuda")
        res = torch.randn(M, N, device="cuda").to(bf) if with_res else None
        cnt = [0]

        def run():
            W = Ws[cnt[0] % n]
            cnt[0] += 1
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, G.p(res), G.p(Cc), M, N, K, K, N, 0, 0, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm")
        line = f"M={M:5d} {name:7s} N={N:5d} K={K:5d}: 256x160 tiles {-(-M // 256) * -(-N // 160):5d} = {-(-M // 256) * -(-N // 160) / 256:4.2f} rounds"
        ref = None
        for fam, knobs in (("no quad", {"gemm_quad": 0}), ("quad", {"gemm_quad": 2}), ("auto", {})):
            L.tune_reset()
            for k, v in knobs.items():
                L.tune_set(k.encode(), v)
            cnt[0] = 0
            run()
            out = Cc.clone()
            ref = out if ref is None else ref
            kern = lib.teo_last_kernel().decode().replace("gemm_", "")
            t = min(timeit(run) for _ in range(3))
            line += f" | {fam} [{kern}] {t:6.1f}{'' if torch.equal(out, ref) else ' DIFF'}"
        L.tune_reset()
        print(line, flush=True)
        del Ws
