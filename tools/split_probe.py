#!/usr/bin/env python3
"""gate/up prefill GEMM at M = 2168: 128x256 kernel vs whole rounds of 256x256 tiles + small remainder beside it (gemm_split),
warm (one W) and cold (8 matrices in rotation, as in the layer loop)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

lib = L.load()
bf = torch.bfloat16
M, N, K = 2168, 22016, 4096
A = torch.randn(M, K, device="cuda").to(bf)
Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
Cc = torch.empty(M, N // 2, dtype=bf, device="cuda")
ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws")
cnt = [0]


def run(rot):
    W = Ws[cnt[0] % 8 if rot else 0]
    cnt[0] += 1
    L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, None, G.p(Cc), M, N, K, K, N // 2, 0, L.GEMM_SWIGLU16, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm")


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


for name, knobs in (("wide (round 2)", {"gemm_split": 0}), ("split: 85 col tiles big + 256 cols small, any-order", {"gemm_split": 1}),
                    ("big plain forced (4 rounds)", {"gemm_split": 0, "gemm_big": 2, "gemm_big_hybrid": 0}),
                    ("big hybrid forced", {"gemm_split": 0, "gemm_big": 2, "gemm_big_hybrid": 2})):
    for k, v in {"gemm_split": 1, "gemm_big": 1, "gemm_big_hybrid": 1}.items():
        lib.teo_tune_set(k.encode(), v)
    for k, v in knobs.items():
        lib.teo_tune_set(k.encode(), v)
    run(False)
    kern = lib.teo_last_kernel().decode()
    tw = min(timeit(lambda: run(False)) for _ in range(3))
    tc = min(timeit(lambda: run(True)) for _ in range(3))
    fl = 2.0 * M * N * K
    print(f"{name:55s} [{kern:16s}] warm {tw:7.1f} us {fl / tw / 1e6:7.1f} TF/s | cold {tc:7.1f} us {fl / tc / 1e6:7.1f} TF/s", flush=True)
# the pieces of the split alone
A2 = A
for label, n in (("big kernel on 21760 columns (765 tiles)", 21760), ("128x128 kernel on the last 256 columns", 256)):
    lib.teo_tune_set(b"gemm_split", 0)
    if n == 21760:
        lib.teo_tune_set(b"gemm_big", 2); lib.teo_tune_set(b"gemm_big_hybrid", 0)
    else:
        lib.teo_tune_set(b"gemm_big", 0); lib.teo_tune_set(b"gemm_wide", 0)
    C2 = torch.empty(M, n // 2, dtype=bf, device="cuda")
    f = lambda: L.check(lib.teo_gemm(G.p(A), G.p(Ws[cnt[0] % 8]), None, None, G.p(C2), M, n, K, K, n // 2, 0, L.GEMM_SWIGLU16, L.TEO_BF16, L.TEO_BF16, G.stream()), "g")  # noqa: E731
    t = min(timeit(f) for _ in range(3))
    print(f"{label:55s} [{lib.teo_last_kernel().decode():16s}] {t:7.1f} us", flush=True)
    lib.teo_tune_set(b"gemm_big", 1); lib.teo_tune_set(b"gemm_wide", 1); lib.teo_tune_set(b"gemm_big_hybrid", 1)
