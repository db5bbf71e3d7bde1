#!/usr/bin/env python3
"""Kernel microbenchmarks at LLaMA-2-7B / ViT-L shapes (GPU box).  Prints one line per kernel: time, GB/s or TFLOP/s.
Usage: python tools/bench_kernels.py [gemv] [attn_decode] [gemm] [attn_prefill] [norm]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

from tools import bench_shim  # noqa: E402

bf = torch.bfloat16
lib = L.load()
SHIM = bench_shim.load()           # event-timed launch chains over the public C ABI (tools/bench_shim.hip)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def bench_gemv():
    shapes = [("qkv", 12288, 4096, True, 0), ("o", 4096, 4096, False, 0), ("gateup", 22016, 4096, True, L.GEMM_SWIGLU16),
              ("down", 4096, 11008, False, 0), ("lm_head", 32000, 4096, True, 0)]
    if os.environ.get("GV_SHAPES") == "splitk":
        shapes = [("o", 4096, 4096, False, 0), ("down", 4096, 11008, False, 0)]
    for name, N, K, norm, flags in shapes:
        n = max(2, int(600e6 // (N * K * 2)))          # > 256 MB L3 in rotation
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(n)]
        x = torch.randn(K, device="cuda").to(bf)
        nw = torch.ones(K, device="cuda").to(bf) if norm else None
        y = torch.empty(N, dtype=bf, device="cuda")
        arr, pp = L.ptr_array([w.data_ptr() for w in Ws])
        avg = C.c_float(0)
        L.check(SHIM.teo_bench_gemv_chain(x.data_ptr(), pp, None, n, nw.data_ptr() if norm else None, y.data_ptr(), N, K, 1e-5,
                                        flags, L.TEO_BF16, 10, C.byref(avg), G.stream()), "chain")
        us = avg.value * 1e3
        print(f"gemv {name:8s} N={N:6d} K={K:6d}: {us:7.2f} us  {N * K * 2 / us / 1e3:7.1f} GB/s", flush=True)
        del Ws


def bench_gemv_mall():
    """Infinity-Cache probe: the same weight matrix re-read every launch (resident if it fits 256 MiB) vs a >600 MB rotation."""
    shapes = [("o", 4096, 4096, False, 0), ("qkv", 12288, 4096, True, 0), ("gateup", 22016, 4096, True, L.GEMM_SWIGLU16)]
    for nt in (1, 0):
        L.tune_set(b"gemv_nt", nt)
        for name, N, K, norm, flags in shapes:
            nrot = max(2, int(600e6 // (N * K * 2)))
            Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(nrot)]
            x = torch.randn(K, device="cuda").to(bf)
            nw = torch.ones(K, device="cuda").to(bf) if norm else None
            y = torch.empty(N, dtype=bf, device="cuda")
            for n in (1, nrot):
                arr, pp = L.ptr_array([w.data_ptr() for w in Ws[:n]])
                avg = C.c_float(0)
                L.check(SHIM.teo_bench_gemv_chain(x.data_ptr(), pp, None, n, nw.data_ptr() if norm else None, y.data_ptr(), N, K,
                                                1e-5, flags, L.TEO_BF16, 20, C.byref(avg), G.stream()), "chain")
                us = avg.value * 1e3
                print(f"mall nt={nt} {name:7s} rot={n:2d} ({N * K * 2 / 1e6:6.1f} MB): {us:7.2f} us {N * K * 2 / us / 1e3:7.1f} GB/s", flush=True)
            del Ws
    L.tune_set(b"gemv_nt", 1)


def bench_skinny():
    """Batched-decode GEMM: weights streamed once for MB conversations; GB/s counts the weight bytes."""
    from teochat_amd.engine import quantize_fp8_rows
    shapes = [("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gateup", 22016, 4096, L.GEMM_SWIGLU16), ("down", 4096, 11008, 0),
              ("lm_head", 32000, 4096, 0)]
    if os.environ.get("SK_SHAPES") == "decode":       # the shapes and flags of the batched decode step
        shapes = [("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gateup8", 22016, 4096, L.GEMM_SWIGLU8), ("down", 4096, 11008, 0)]
    if os.environ.get("SK_SHAPES") == "gu":
        shapes = [("gateup", 22016, 4096, L.GEMM_SWIGLU16), ("gu_plain", 22016, 4096, 0)]
    tiled = L.GEMM_WTILED if int(os.environ.get("SK_TILED", "1")) else 0      # tiled and row-major cost the same to set up here
    L.tune_set(b"skinny_nt", int(os.environ.get("SK_NT", "1")))
    L.tune_set(b"skinny_stream", int(os.environ.get("SK_STREAM", "1")))
    L.tune_set(b"skinny_ring", int(os.environ.get("SK_RING", "0")))
    L.tune_set(b"skinny_unr", int(os.environ.get("SK_UNR", "0")))
    L.tune_set(b"skinny_waves", int(os.environ.get("SK_WAVES", "0")))
    L.tune_set(b"skinny_grid", int(os.environ.get("SK_GRID", "0")))
    for fp8 in (False, True):
        for name, N, K, flags in shapes:
            wb = 1 if fp8 else 2
            n = max(2, int(600e6 // (N * K * wb)))
            if fp8:
                qs = [quantize_fp8_rows((torch.randn(N, K, device="cuda") * 0.02).to(bf))[:2] for _ in range(n)]
                Ws = [q for q, _ in qs]
                Ss = [s_ for _, s_ in qs]
                arr2, pp2 = L.ptr_array([s_.data_ptr() for s_ in Ss])
            else:
                Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(n)]
                pp2 = None
            arr, pp = L.ptr_array([w.data_ptr() for w in Ws])
            norm = name in ("qkv", "gateup", "lm_head") and int(os.environ.get("SK_NORM", "1"))
            nw = torch.ones(K, device="cuda").to(bf)
            for MB in (8, 16):
                x = torch.randn(MB, K, device="cuda").to(bf)
                y = torch.empty(MB, N, dtype=bf, device="cuda")
                line = f"skinny {'fp8 ' if fp8 else 'bf16'} {name:8s} MB={MB:2d}:"
                for tiles in [int(t) for t in os.environ.get("SK_TILES", "1,2,4,8").split(",")]:
                    if tiles == 1 and (flags & L.GEMM_SWIGLU16):
                        continue
                    L.tune_set(b"skinny_tiles", tiles)
                    avg = C.c_float(0)
                    L.check(SHIM.teo_bench_skinny_chain(x.data_ptr(), pp, pp2, n, nw.data_ptr() if norm else None, y.data_ptr(), MB, N, K, flags | tiled, 10, C.byref(avg),
                                                      G.stream()), "chain")
                    us = avg.value * 1e3
                    line += f"  T{tiles} {us:6.1f}us {N * K * wb / us / 1e3:6.0f}GB/s"
                print(line, flush=True)
            L.tune_set(b"skinny_tiles", 0)
            if not fp8:                                   # the prefill GEMM at the same M for comparison
                x = torch.randn(16, K, device="cuda").to(bf)
                us = timeit(lambda: [G.gemm(x, w, flags=flags) for w in Ws], iters=5) / len(Ws)
                print(f"   (prefill GEMM kernel at M=16: {us:6.1f}us {N * K * 2 / us / 1e3:6.0f}GB/s)", flush=True)
            del Ws


def bench_gemv_fp8_sweep():
    for v in (10, 11, 12, 13):
        L.tune_set(b"gemv_variant", v)
        print("fp8 variant", v, flush=True)
        bench_gemv_fp8()
    L.tune_set(b"gemv_variant", -1)


def bench_gemv_splitk_sweep():
    """o / down projections of the single-conversation step: rows per workgroup x chunks per thread of the split-K GEMV."""
    os.environ["GV_SHAPES"] = "splitk"
    for r in (2, 4):
        for u in (1, 2, 3, 4, 6):
            L.tune_set(b"gemv_splitk_r", r)
            L.tune_set(b"gemv_splitk_u", u)
            print(f"split-K rows {r} chunks {u}", flush=True)
            bench_gemv_fp8()
            bench_gemv()
    L.tune_set(b"gemv_splitk_r", 0)
    L.tune_set(b"gemv_splitk_u", 0)


def bench_gemv_fp8():
    shapes = [("qkv", 12288, 4096, True, 0), ("o", 4096, 4096, False, 0), ("gateup", 22016, 4096, True, L.GEMM_SWIGLU16),
              ("down", 4096, 11008, False, 0), ("lm_head", 32000, 4096, True, 0)]
    if os.environ.get("GV_SHAPES") == "splitk":
        shapes = [("o", 4096, 4096, False, 0), ("down", 4096, 11008, False, 0)]
    for name, N, K, norm, flags in shapes:
        n = max(2, int(600e6 // (N * K)))
        Ws = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device="cuda") for _ in range(n)]
        Ss = [torch.full((N,), 2.0 ** -9, device="cuda") for _ in range(n)]
        x = torch.randn(K, device="cuda").to(bf)
        nw = torch.ones(K, device="cuda").to(bf) if norm else None
        y = torch.empty(N, dtype=bf, device="cuda")
        arr, pp = L.ptr_array([w.data_ptr() for w in Ws])
        arr2, pp2 = L.ptr_array([s_.data_ptr() for s_ in Ss])
        avg = C.c_float(0)
        L.check(SHIM.teo_bench_gemv_chain(x.data_ptr(), pp, pp2, n, nw.data_ptr() if norm else None, y.data_ptr(), N, K, 1e-5,
                                        flags, L.TEO_BF16, 10, C.byref(avg), G.stream()), "chain")
        us = avg.value * 1e3
        print(f"gemv fp8 {name:8s} N={N:6d} K={K:6d}: {us:7.2f} us  {N * K / us / 1e3:7.1f} GB/s", flush=True)
        del Ws


def bench_gemv_sweep():
    shapes = [("qkv", 12288, 4096, True, 0), ("o", 4096, 4096, False, 0), ("gateup", 22016, 4096, True, L.GEMM_SWIGLU16),
              ("down", 4096, 11008, False, 0)]
    bufs = {}
    for name, N, K, norm, flags in shapes:
        n = max(2, int(600e6 // (N * K * 2)))
        bufs[name] = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(n)]
    for maxb in [int(v_) for v_ in os.environ.get("GV_MAXB", "512,768,1024,1376,1536,2048,2752,3072").split(",")]:
        for nt in (1,):
            for v in [int(v_) for v_ in os.environ.get("GV_VARIANTS", "-1").split(",")]:
                L.tune_set(b"gemv_variant", v)
                L.tune_set(b"gemv_nt", nt)
                L.tune_set(b"gemv_max_blocks", maxb)
                line = f"variant {v:2d} nt={nt} maxb={maxb}:"
                tot = 0.0
                for name, N, K, norm, flags in shapes:
                    Ws = bufs[name]
                    x = torch.randn(K, device="cuda").to(bf)
                    nw = torch.ones(K, device="cuda").to(bf) if norm else None
                    y = torch.empty(N, dtype=bf, device="cuda")
                    arr, pp = L.ptr_array([w.data_ptr() for w in Ws])
                    avg = C.c_float(0)
                    L.check(SHIM.teo_bench_gemv_chain(x.data_ptr(), pp, None, len(Ws), nw.data_ptr() if norm else None, y.data_ptr(),
                                                    N, K, 1e-5, flags, L.TEO_BF16, 5, C.byref(avg), G.stream()), "chain")
                    us = avg.value * 1e3
                    tot += us
                    line += f"  {name} {us:6.1f}us {N * K * 2 / us / 1e3:6.0f}GB/s"
                print(line + f"  | layer {tot:6.1f}us", flush=True)
    L.tune_set(b"gemv_variant", -1)


def bench_gemm():
    """MFMA GEMM at the prefill shapes as the library dispatches it: without a workspace (wide / plain tiling) vs with one
    (stream-K forms where the rounds model picks them), interleaved."""
    shapes = [("qkv", 2168, 12288, 4096, 0), ("o", 2168, 4096, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16),
              ("down", 2168, 4096, 11008, 0), ("vit_qkv", 2056, 3072, 1024, 0), ("vit_fc1", 2056, 4096, 1024, 0),
              ("vit_fc2", 2056, 1024, 4096, 0), ("sq4096", 4096, 4096, 4096, 0), ("qkv_T16", 4208, 12288, 4096, 0),
              ("down_B8", 17344, 4096, 11008, 0)]
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    if os.environ.get("TEO_SK_FORCE"):
        L.tune_set(b"gemm_sk", 2)
        shapes = [("sq4096", 4096, 4096, 4096, 0), ("sq8192x4096", 8192, 4096, 4096, 0), ("3072x4096", 3072, 4096, 4096, 0)]
    for name, M, N, K, flags in shapes:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")

        def run(w):
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, None, G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16,
                                    G.p(w) if w is not None else None, G.stream()), "gemm")
        res = {0: [], 1: []}
        for _ in range(3):
            res[0].append(timeit(lambda: run(None)))
            res[1].append(timeit(lambda: run(ws)))
        a, b = min(res[0]), min(res[1])
        print(f"gemm {name:8s} M={M} N={N} K={K}: plain {a:8.1f} us {2.0 * M * N * K / a / 1e6:7.1f} TFLOP/s | stream-K {b:8.1f} us "
              f"{2.0 * M * N * K / b / 1e6:7.1f} TFLOP/s", flush=True)


def bench_gemm_wide():
    """wide-tile LDS-DMA kernel (gemm_wide = 2: forced) vs the 128 x 128 kernel (gemm_wide = 0), interleaved."""
    shapes = [("qkv", 2168, 12288, 4096, 0), ("o", 2168, 4096, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16),
              ("down", 2168, 4096, 11008, 0), ("vit_fc1", 2056, 4096, 1024, 0), ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0),
              ("qkv_T16", 4208, 12288, 4096, 0), ("qkv_T2", 638, 12288, 4096, 0), ("gateup_B8", 17344, 22016, 4096, L.GEMM_SWIGLU16)]
    for name, M, N, K, flags in shapes:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        res = {0: [], 2: []}
        outs = {}
        for _ in range(3):
            for mode in (2, 0):
                L.tune_set(b"gemm_wide", mode)
                res[mode].append(timeit(lambda: G.gemm(A, W, flags=flags)))
                outs[mode] = G.gemm(A, W, flags=flags)
        L.tune_set(b"gemm_wide", 1)
        a, b_ = min(res[2]), min(res[0])
        fl = 2.0 * M * N * K
        print(f"gemm {name:9s} M={M} N={N} K={K}: wide {a:8.1f} us {fl / a / 1e6:7.1f} TFLOP/s | 128x128 {b_:8.1f} us {fl / b_ / 1e6:7.1f} TFLOP/s"
              f" | bit-identical: {bool(torch.equal(outs[0], outs[2]))}", flush=True)


def bench_gemm_big():
    """256 x 256 kernel (gemm_big = 2: forced) vs the default dispatch."""
    shapes = [("qkv", 2168, 12288, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16), ("o", 2168, 4096, 4096, 0),
              ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0), ("qkv_T16", 4208, 12288, 4096, 0),
              ("gateup_T16", 4208, 22016, 4096, L.GEMM_SWIGLU16), ("qkv_2048", 2048, 12288, 4096, 0), ("gateup_B8", 17344, 22016, 4096, L.GEMM_SWIGLU16)]
    for name, M, N, K, flags in shapes:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        res = {0: [], 2: [], 3: []}
        outs = {}
        for _ in range(3):
            for mode in (2, 3, 0):
                L.tune_set(b"gemm_big", 2 if mode == 2 else (1 if mode == 3 else 0))
                res[mode].append(timeit(lambda: G.gemm(A, W, flags=flags)))
                outs[mode] = G.gemm(A, W, flags=flags)
        L.tune_set(b"gemm_big", 2)
        ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
        L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        run_ws = lambda: L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, None, G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm_ws")
        t_h = min(timeit(run_ws) for _ in range(3))
        same_h = bool(torch.equal(Cc, outs[0]))
        L.tune_set(b"gemm_big", 1)
        t_auto_ws = min(timeit(run_ws) for _ in range(3))
        a, b_ = min(res[2]), min(res[0])
        fl = 2.0 * M * N * K
        tiles = -(-M // 256) * -(-N // 256)
        print(f"gemm {name:10s} M={M} N={N} K={K}: big {a:8.1f} us {fl / a / 1e6:7.1f} TFLOP/s ({tiles} tiles = {tiles / 256:.2f} rounds, {a / -(-tiles // 256) / (K // 64) * 1e3:6.0f} ns per K tile) | "
              f"hybrid {t_h:8.1f} us {fl / t_h / 1e6:7.1f} ({'=' if same_h else 'DIFF'}) | auto+ws {t_auto_ws:8.1f} | auto {min(res[3]):8.1f} us | no big {b_:8.1f} us {fl / b_ / 1e6:7.1f} TFLOP/s | bit-identical: {bool(torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3]))}", flush=True)


def bench_gemm_cohort():
    """round 5: stream-K ranges of the 256 x 256 hybrid kernel as XCD-local cohorts (gemm_big_cohort = 8 / 16 / 32) vs the linear ranges (0), on cold
    weights (8 matrices in rotation, as in the layer loop) and interleaved; every form must be bit-identical to the linear one."""
    shapes = [("qkv", 2168, 12288, 4096, 0, "big"), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16, "big"), ("o", 2168, 4096, 4096, 0, "wide"),
              ("down", 2168, 4096, 11008, 0, "wide"), ("qkv_T2", 638, 12288, 4096, 0, "big"), ("gateup_T2", 638, 22016, 4096, L.GEMM_SWIGLU16, "big"),
              ("qkv_T16", 4208, 12288, 4096, 0, "big"), ("gateup_T16", 4208, 22016, 4096, L.GEMM_SWIGLU16, "big"), ("down_T16", 4208, 4096, 11008, 0, "big"),
              ("o_T16", 4208, 4096, 4096, 0, "big"), ("vit_fc1", 2056, 4096, 1024, 0, "wide")]
    only = os.environ.get("COHORT_SHAPES")
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    for name, M, N, K, flags, fam in shapes:
        if only and name not in only.split(","):
            continue
        A = torch.randn(M, K, device="cuda").to(bf)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        cnt = [0]

        def run():
            W = Ws[cnt[0] % 8]
            cnt[0] += 1
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, None, G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm_ws")
        variants = [("auto", {}), ("wide_sk", {"gemm_big": 0, "gemm_sk": 2})] + [(f"big{c}", {"gemm_big_cohort": c, "gemm_big": 2, "gemm_big_hybrid": 2}) for c in (0, 8, 16, 32)]
        res = {v: [] for v, _ in variants}
        outs, kern = {}, {}
        for rep in range(3):
            for v, knobs in variants:
                L.tune_reset()
                for k_, val in knobs.items():
                    assert L.tune_set(k_.encode(), val) == 0, k_
                res[v].append(timeit(run, iters=16, warm=2))
                if rep == 0:
                    cnt[0] = 0
                    run()
                    torch.cuda.synchronize()
                    outs[v] = Cc.clone()
                    kern[v] = lib.teo_last_kernel().decode()
        L.tune_reset()
        fl = 2.0 * M * N * K
        ref = outs["big0"]
        line = " | ".join(f"{v} [{kern[v].replace('gemm_', '')}] {min(res[v]):7.1f}{'' if torch.equal(outs[v], ref) else ' DIFF'}" for v, _ in variants)
        best = min(min(r) for r in res.values())
        print(f"cohort {name:10s} M={M} N={N} K={K} ({fl / 1e9:6.1f} GF; best {fl / best / 1e6:6.0f} TF/s): {line}", flush=True)
        st = C.c_int(0)
        L.check(lib.teo_gemm_workspace_status(G.p(ws), C.byref(st), G.stream()), "ws status")
        assert st.value == 0, "stream-K hand-off timed out"


def bench_gemm_cold():
    """the prefill GEMMs with the weight matrix resident in the Infinity Cache (one W repeated) vs cold (8 matrices in rotation, as in
    the real layer loop where every layer's weights come from HBM once)."""
    for name, M, N, K, flags in [("qkv", 2168, 12288, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16), ("o", 2168, 4096, 4096, 0),
                                 ("down", 2168, 4096, 11008, 0)]:
        A = torch.randn(M, K, device="cuda").to(bf)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
        L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
        cnt = [0]
        def run(rot):
            W = Ws[cnt[0] % 8 if rot else 0]
            cnt[0] += 1
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, None, G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm_ws")
        warm = min(timeit(lambda: run(False)) for _ in range(3))
        cold = min(timeit(lambda: run(True)) for _ in range(3))
        fl = 2.0 * M * N * K
        print(f"gemm {name:7s}: W resident {warm:7.1f} us {fl / warm / 1e6:7.1f} TFLOP/s | W cold (8 in rotation) {cold:7.1f} us {fl / cold / 1e6:7.1f} TFLOP/s", flush=True)


def bench_gemm_prefetch():
    """cold weights (8 matrices in rotation) with the NEXT matrix touched from a side stream while the current GEMM runs: does
    pulling W into the 256 MB Infinity Cache ahead of the GEMM recover the resident-W time?"""
    side = torch.cuda.Stream()
    for name, M, N, K, flags in [("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16), ("down", 2168, 4096, 11008, 0), ("o", 2168, 4096, 4096, 0),
                                 ("qkv", 2168, 12288, 4096, 0)]:
        A = torch.randn(M, K, device="cuda").to(bf)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
        L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
        sink = torch.zeros(1, device="cuda")
        cnt = [0]
        def run(prefetch):
            i = cnt[0] % 8
            cnt[0] += 1
            if prefetch:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    sink.add_(Ws[(i + 1) % 8].view(torch.int16)[:, ::64].sum())      # one 2-byte read per 128-byte line
            L.check(lib.teo_gemm_ws(G.p(A), G.p(Ws[i]), None, None, G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm_ws")
        cold = min(timeit(lambda: run(False)) for _ in range(3))
        pf = min(timeit(lambda: run(True)) for _ in range(3))
        torch.cuda.synchronize()
        fl = 2.0 * M * N * K
        print(f"gemm {name:7s}: cold {cold:7.1f} us {fl / cold / 1e6:7.1f} TFLOP/s | cold + next W touched from a side stream {pf:7.1f} us {fl / pf / 1e6:7.1f} TFLOP/s", flush=True)


def bench_gemm_wide_sched():
    """instruction-order variants of the wide kernel's K loop (gemm_wide_sched), forced wide, plain (non-stream-K) launch."""
    shapes = [("qkv", 2168, 12288, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16), ("sq8192", 8192, 8192, 8192, 0),
              ("qkv_T16", 4208, 12288, 4096, 0)]
    L.tune_set(b"gemm_wide", 2)
    for name, M, N, K, flags in shapes:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        fl = 2.0 * M * N * K
        ref = None
        line = f"gemm {name:9s}:"
        for sched, grp in ((0, 1), (1, 1), (1, 4)):
            L.tune_set(b"gemm_wide_sched", sched)
            L.tune_set(b"gemm_wide_group", grp)
            t = min(timeit(lambda: G.gemm(A, W, flags=flags)) for _ in range(3))
            out = G.gemm(A, W, flags=flags)
            ref = out if ref is None else ref
            line += f" s{sched}g{grp} {t:7.1f} us {fl / t / 1e6:7.1f} TF ({'=' if torch.equal(out, ref) else 'DIFF'}) |"
        print(line, flush=True)
    L.tune_set(b"gemm_wide_sched", 0)
    L.tune_set(b"gemm_wide_group", 1)
    L.tune_set(b"gemm_wide", 1)


def bench_gemm_fp8():
    """w8a8 GEMM on the scaled fp8 MFMA vs the bf16 MFMA kernel at the prefill shapes (incl. the activation quantiser)."""
    for name, M, N, K, flags in [("qkv", 2168, 12288, 4096, 0), ("o", 2168, 4096, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16),
                                 ("down", 2168, 4096, 11008, 0), ("qkv_B8", 17344, 12288, 4096, 0), ("gateup_B8", 17344, 22016, 4096, L.GEMM_SWIGLU16),
                                 ("down_B8", 17344, 4096, 11008, 0), ("o_B8", 17344, 4096, 4096, 0)]:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        A8 = torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8)
        W8 = torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8)
        sa, sw = torch.ones(M, device="cuda"), torch.ones(N, device="cuda")
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        q8 = torch.empty(M, K, dtype=torch.uint8, device="cuda")
        t16 = timeit(lambda: G.gemm(A, W, flags=flags))
        run8 = lambda: L.check(lib.teo_gemm_fp8(G.p(A8), G.p(sa), G.p(W8), G.p(sw), None, G.p(Cc), M, N, K, K, Nc, flags, L.TEO_BF16,
                                                 G.stream()), "gemm_fp8")
        L.tune_set(b"gemm_fp8_wide", 0)
        t8n = timeit(run8)
        c0 = Cc.clone()
        L.tune_set(b"gemm_fp8_wide", 2)
        t8 = timeit(run8)
        same = bool(torch.equal(c0, Cc))
        L.tune_set(b"gemm_fp8_wide", 1)
        L.tune_set(b"gemm_fp8_big", 2)
        t8b = timeit(run8)
        same = same and bool(torch.equal(c0, Cc))
        L.tune_set(b"gemm_fp8_big", 1)
        ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
        L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
        tsk = timeit(lambda: L.check(lib.teo_gemm_fp8_ws(G.p(A8), G.p(sa), G.p(W8), G.p(sw), None, G.p(Cc), M, N, K, K, Nc, flags, L.TEO_BF16,
                                                         G.p(ws), G.stream()), "gemm_fp8_ws"))
        same = same and bool(torch.equal(c0, Cc))
        tq = timeit(lambda: L.check(lib.teo_quant_rows_fp8(G.p(A), None, G.p(q8), G.p(sa), M, K, K, 1e-5, G.stream()), "quant"))
        fl = 2.0 * M * N * K
        print(f"gemm_fp8 {name:8s} M={M} N={N} K={K}: bf16 {t16:8.1f} us {fl / t16 / 1e6:7.1f} TFLOP/s | fp8 128x128 {t8n:8.1f} us {fl / t8n / 1e6:7.1f} | "
              f"fp8 wide {t8:8.1f} us {fl / t8 / 1e6:7.1f} | fp8 256x256 {t8b:8.1f} us {fl / t8b / 1e6:7.1f} | dispatch+ws {tsk:8.1f} us {fl / tsk / 1e6:7.1f} TFLOP/s (bit-identical {same}) | quantiser {tq:6.1f} us", flush=True)


def bench_attn_prefill():
    """flash32 (flash.hip): causal workgroup order heavy-first with the second pass mirrored (flash_order = 1) vs plain heavy-first (0)."""
    for (B, H, S, d, causal) in [(1, 32, 2168, 128, True), (1, 32, 4208, 128, True), (8, 16, 257, 64, False), (1, 32, 638, 128, True)]:
        q = torch.randn(B, H, S, d, device="cuda").to(bf)
        k = torch.randn(B, H, S, d, device="cuda").to(bf)
        v = torch.randn(B, H, S, d, device="cuda").to(bf)
        vt = G.make_vt(v)
        fl = 4.0 * B * H * S * S * d * (0.5 if causal else 1.0)
        res = {}
        for rnd_ in range(3):
            for flash in (1, 0, 2):
                L.tune_set(b"flash_order", 1)
                L.tune_set(b"flash_pipe", 0 if flash == 2 else -1)
                L.tune_set(b"flash_order", 0 if flash == 0 else 1)
                us = timeit(lambda: G.attention(q, k, v, causal, d ** -0.5, vt=vt))
                res.setdefault(flash, []).append(us)
        L.tune_reset()
        a, b_ = min(res[1]), min(res[0])
        print(f"attn prefill B={B} H={H} S={S} d={d} causal={causal}: mirrored {a:8.1f} us {fl / a / 1e6:7.1f} TFLOP/s | "
              f"heavy-first {b_:8.1f} us {fl / b_ / 1e6:7.1f} TFLOP/s | one tile at a time {min(res[2]):8.1f} us", flush=True)


def bench_norm():
    for rows, dim in [(2168, 4096), (2056, 1024)]:
        x = torch.randn(rows, dim, device="cuda").to(bf)
        w = torch.ones(dim, device="cuda").to(bf)
        us = timeit(lambda: G.rmsnorm(x, w, 1e-5))
        print(f"rmsnorm {rows}x{dim}: {us:7.2f} us  {rows * dim * 4 / us / 1e3:7.1f} GB/s", flush=True)
        us = timeit(lambda: G.layernorm(x, w, w, 1e-5))
        print(f"layernorm {rows}x{dim}: {us:7.2f} us  {rows * dim * 4 / us / 1e3:7.1f} GB/s", flush=True)




def bench_gemm_depth():
    for depth in (1, 2):
        L.tune_set(b"gemm_depth", depth)
        print("gemm_depth", depth, flush=True)
        bench_gemm()
    L.tune_set(b"gemm_depth", 0)


def bench_gemm_stride():
    """Does the power-of-two row stride (K=4096 -> 8 KB) hurt?  Same M,N with K = 4096 vs 4160 (= 65 * 64)."""
    for (M, N, K) in [(2168, 4096, 4096), (2168, 4096, 4160), (2168, 12288, 4096), (2168, 12288, 4160), (4096, 4096, 4096),
                      (4096, 4096, 4160), (2168, 4096, 11008), (2168, 4096, 11072)]:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        us = timeit(lambda: G.gemm(A, W))
        print(f"gemm M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)

def bench_yardstick():
    """NOT a product path: what the vendor libraries reach on this box at the same shapes (torch.matmul -> hipBLASLt / rocBLAS,
    scaled_dot_product_attention -> the bundled flash kernels), next to the library's own kernels in the same process.  A yardstick for
    the MFMA phases: the product cannot use them (every tile family here is bit-identical to every other, the library kernels are not, and
    the fused epilogues / S^T layouts are the library's own), but they say what a tuned kernel of that shape gets out of the chip."""
    import torch.nn.functional as F
    shapes = [("qkv", 2168, 12288, 4096, 0), ("o", 2168, 4096, 4096, 0), ("gateup", 2168, 22016, 4096, L.GEMM_SWIGLU16),
              ("down", 2168, 4096, 11008, 0), ("vit_qkv", 2056, 3072, 1024, 0), ("vit_fc1", 2056, 4096, 1024, 0),
              ("vit_fc2", 2056, 1024, 4096, 0), ("qkv_C2", 638, 12288, 4096, 0), ("gateup_C2", 638, 22016, 4096, L.GEMM_SWIGLU16),
              ("qkv_T16", 4208, 12288, 4096, 0), ("sq8192", 8192, 8192, 8192, 0)]
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    for name, M, N, K, flags in shapes:
        n = max(2, int(600e6 // (N * K * 2)))          # weights in rotation: > the 256 MB Infinity Cache, as in a real prefill
        A = torch.randn(M, K, device="cuda").to(bf)
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(n)]
        Nc = N // 2 if flags else N
        Cc = torch.empty(M, Nc, dtype=bf, device="cuda")
        Cl = torch.empty(M, N, dtype=bf, device="cuda")
        it = [0]

        def ours():
            w = Ws[it[0] % n]; it[0] += 1
            L.check(lib.teo_gemm_ws(G.p(A), G.p(w), None, None, G.p(Cc), M, N, K, K, Nc, 0, flags, L.TEO_BF16, L.TEO_BF16, G.p(ws), G.stream()), "gemm")

        def theirs():
            w = Ws[it[0] % n]; it[0] += 1
            torch.matmul(A, w.t(), out=Cl)
        a = min(timeit(ours, iters=4 * n) for _ in range(3))
        b = min(timeit(theirs, iters=4 * n) for _ in range(3))
        fl = 2.0 * M * N * K
        print(f"yardstick gemm {name:9s} M={M} N={N} K={K}: this library{' (+SwiGLU)' if flags else ''} {a:8.1f} us {fl / a / 1e6:7.1f} TFLOP/s | "
              f"torch.matmul {b:8.1f} us {fl / b / 1e6:7.1f} TFLOP/s | ratio {a / b:5.2f}", flush=True)
        del Ws
    for (B, H, S, d, causal) in [(1, 32, 2168, 128, True), (1, 32, 4208, 128, True), (8, 16, 257, 64, False), (1, 32, 638, 128, True)]:
        q = torch.randn(B, H, S, d, device="cuda").to(bf)
        k = torch.randn(B, H, S, d, device="cuda").to(bf)
        v = torch.randn(B, H, S, d, device="cuda").to(bf)
        vt = G.make_vt(v)
        fl = 4.0 * B * H * S * S * d * (0.5 if causal else 1.0)
        a = min(timeit(lambda: G.attention(q, k, v, causal, d ** -0.5, vt=vt)) for _ in range(3))
        try:
            b = min(timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal)) for _ in range(3))
            theirs = f"{b:8.1f} us {fl / b / 1e6:7.1f} TFLOP/s | ratio {a / b:5.2f}"
        except Exception as e:  # noqa: BLE001
            theirs = f"unavailable ({type(e).__name__})"
        print(f"yardstick attention B={B} H={H} S={S} d={d} causal={causal}: this library {a:8.1f} us {fl / a / 1e6:7.1f} TFLOP/s | torch SDPA {theirs}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemv", "gemm", "attn_prefill", "norm"]
    for w in which:
        {"gemv": bench_gemv, "gemv_mall": bench_gemv_mall, "skinny": bench_skinny, "gemv_fp8": bench_gemv_fp8, "gemv_fp8_sweep": bench_gemv_fp8_sweep, "gemv_splitk_sweep": bench_gemv_splitk_sweep, "gemv_sweep": bench_gemv_sweep, "gemm_stride": bench_gemm_stride, "gemm_depth": bench_gemm_depth, "gemm": bench_gemm, "gemm_fp8": bench_gemm_fp8, "gemm_wide": bench_gemm_wide, "gemm_big": bench_gemm_big, "gemm_prefetch": bench_gemm_prefetch, "gemm_cold": bench_gemm_cold, "gemm_cohort": bench_gemm_cohort, "gemm_wide_sched": bench_gemm_wide_sched, "attn_prefill": bench_attn_prefill, "norm": bench_norm, "yardstick": bench_yardstick}[w]()

