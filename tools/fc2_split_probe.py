#!/usr/bin/env python3
"""Would a two-launch split-K (two K halves side by side in one launch + an elementwise reduce) beat the plain tiles for the tower's fc2
(M = 2056, N = 1024, K = 4096)?  Emulated with the shipped kernels: ONE GEMM over 4112 rows (two stacked copies of A: the tile count and
co-residency of a 2-way split) at K = 2048, plus a torch add as the stand-in for the reduce.  usage: python tools/fc2_split_probe.py"""
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


def timeit(fn, iters=30, warm=5):
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


def gemm(A, W, C, out_dtype):
    M, K = A.shape
    N = W.shape[0]
    L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, None, G.p(C), M, N, K, K, N, 0, 0, L.TEO_BF16, out_dtype, G.p(ws), G.stream()), "gemm")


Ws = [(torch.randn(1024, 4096, device="cuda") * 0.02).to(bf) for _ in range(8)]
Wh = [(torch.randn(1024, 2048, device="cuda") * 0.02).to(bf) for _ in range(8)]
Wq = [(torch.randn(1024, 1024, device="cuda") * 0.02).to(bf) for _ in range(8)]
cnt = [0]
for name, M, K, Wl, splits in (("plain fc2", 2056, 4096, Ws, 1), ("2-way split (4112 rows, K = 2048)", 4112, 2048, Wh, 2), ("4-way split (8224 rows, K = 1024)", 8224, 1024, Wq, 4)):
    A = torch.randn(M, K, device="cuda").to(bf)
    for od, tag in ((L.TEO_BF16, "bf16 out"), (L.TEO_F32, "fp32 out")):
        C = torch.empty(M, 1024, dtype=bf if od == L.TEO_BF16 else torch.float32, device="cuda")

        def run():
            cnt[0] += 1
            gemm(A, Wl[cnt[0] % 8], C, od)
        t = min(timeit(run) for _ in range(2))
        print(f"{name:38s} {tag}: {t:6.1f} us  [{lib.teo_last_kernel().decode()}]", flush=True)
    if splits > 1:
        P = torch.randn(splits, 2056, 1024, device="cuda")
        res = torch.randn(2056, 1024, device="cuda").to(bf)
        t = min(timeit(lambda: (P.sum(0) + res.float()).to(bf)) for _ in range(2))
        print(f"   torch stand-in for the {splits}-partial reduce + residual + round: {t:6.1f} us", flush=True)
