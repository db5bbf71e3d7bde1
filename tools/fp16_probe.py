#!/usr/bin/env python3
"""Why is the fp16 line behind the bf16 one (VERDICT r05 "Next round" #4)?  Per-kernel A/B on the C3 shapes with the data as the variable:

  the SAME kernel instantiation, the SAME shapes, cold weights (rotating copies), three kinds of operand data
    random  : N(0, s^2) rounded to the 16-bit format (fp16: 10 fraction bits populated; bf16: 7)
    trunc7  : fp16 only -- the same values with the low 3 fraction bits cleared (what a bf16 checkpoint cast to fp16 holds: bf16's 7 bits)
    zeros   : all-zero operands (the floor of the data-dependent power draw)

If fp16 `trunc7` runs at bf16 `random`'s time, the gap is the multiplier array's switching activity (power -> sustained clock), a
hardware effect of the operand data, not an instruction-level difference between the two instantiations.
usage (GPU box): python tools/fp16_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402

lib = L.load()
dev = "cuda:0"
cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
L.check(lib.teo_gemm_workspace_init(ws.data_ptr(), cur), "ws")


def event_us(fn, iters=24, warm=4):
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


def make(shape, std, dtype, kind):
    if kind == "zeros":
        return torch.zeros(*shape, dtype=dtype, device=dev)
    t = (torch.randn(*shape, device=dev) * std).to(dtype)
    if kind == "trunc7":
        assert dtype == torch.float16
        t = (t.view(torch.int16) & ~7).view(torch.float16)           # clear the low 3 of the 10 fraction bits
    return t


def gemm_case(name, M, N, K, flags, with_res):
    out = {}
    for dtype, dt, kinds in ((torch.bfloat16, L.TEO_BF16, ("random", "zeros")), (torch.float16, L.TEO_F16, ("random", "trunc7", "zeros"))):
        for kind in kinds:
            A = make((M, K), 1.0, dtype, kind)
            Ws = [make((N, K), 0.02, dtype, kind) for _ in range(6)]
            Nc = N // 2 if flags else N
            Cc = torch.empty(M, Nc, dtype=dtype, device=dev)
            res = make((M, Nc), 1.0, dtype, kind) if with_res else None
            cnt = [0]

            def run():
                W = Ws[cnt[0] % len(Ws)]
                cnt[0] += 1
                L.check(lib.teo_gemm_ws(A.data_ptr(), W.data_ptr(), None, res.data_ptr() if res is not None else None, Cc.data_ptr(), M, N, K, K, Nc, 0,
                                        flags, dt, dt, ws.data_ptr(), cur), "gemm")
            run()
            kern = lib.teo_last_kernel().decode()
            out[(str(dtype).split(".")[-1], kind)] = (min(event_us(run) for _ in range(3)), kern)
            del A, Ws, Cc, res
    return out


def gemv_case(name, N, K, flags, with_norm):
    out = {}
    for dtype, dt, kinds in ((torch.bfloat16, L.TEO_BF16, ("random", "zeros")), (torch.float16, L.TEO_F16, ("random", "trunc7", "zeros"))):
        for kind in kinds:
            x = make((K,), 1.0, dtype, kind if kind != "zeros" else "random")
            Ws = [make((N, K), 0.02, dtype, kind) for _ in range(8)]
            g = torch.ones(K, dtype=dtype, device=dev) if with_norm else None
            y = torch.empty(N // 2 if flags else N, dtype=dtype, device=dev)
            cnt = [0]

            def run():
                W = Ws[cnt[0] % len(Ws)]
                cnt[0] += 1
                L.check(lib.teo_gemv(x.data_ptr(), W.data_ptr(), g.data_ptr() if g is not None else None, None, y.data_ptr(), N, K, 1e-5, flags, dt, dt, cur), "gemv")
            out[(str(dtype).split(".")[-1], kind)] = (min(event_us(run, iters=64) for _ in range(3)), "gemv")
            del Ws
    return out


def show(name, r):
    b = r[("bfloat16", "random")][0]
    line = f"{name:28s} [{r[('bfloat16', 'random')][1]}]  bf16 random {b:7.1f} us | bf16 zeros {r[('bfloat16', 'zeros')][0]:7.1f}"
    for kind in ("random", "trunc7", "zeros"):
        t = r[("float16", kind)][0]
        line += f" | fp16 {kind} {t:7.1f} ({100 * (t / b - 1):+.1f} %)"
    print(line, flush=True)


M = 2168
print(f"# fp16 vs bf16, operand data as the variable (M = {M}; us, min of 3 x 24 launches over 6 rotating weight copies)")
show("prefill qkv", gemm_case("qkv", M, 12288, 4096, 0, False))
show("prefill o", gemm_case("o", M, 4096, 4096, 0, True))
show("prefill gate/up + SwiGLU", gemm_case("gateup", M, 22016, 4096, L.GEMM_SWIGLU16, False))
show("prefill down", gemm_case("down", M, 4096, 11008, 0, True))
show("decode gate/up GEMV", gemv_case("gateup", 22016, 4096, L.GEMM_SWIGLU16, True))
show("decode lm_head-like GEMV", gemv_case("lmh", 32000, 4096, 0, True))
show("decode down GEMV", gemv_case("down", 4096, 11008, 0, False))
