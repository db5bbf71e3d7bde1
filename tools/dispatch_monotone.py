#!/usr/bin/env python3
"""A GEMM with more rows cannot be faster: walk M in steps of 64 through the production dispatch for the four LLaMA prefill GEMMs (and the
tower's four) and flag every row count whose time exceeds that of a LARGER row count by more than 8 % -- each such inversion is a dispatch
rule picking a worse tile family than its neighbour's (the complement of tools/shape_sweep.py, which walks the shapes a caller can produce
through the whole phases; this walks the rule boundaries themselves).  Weights rotate over enough copies (>= 600 MB) that they come from HBM, not from the Infinity Cache.

--fp8 walks the four LLaMA GEMMs of the w8a8 prefill (teo_gemm_fp8_ws) instead.

usage (GPU box): python tools/dispatch_monotone.py [--fp8] [--lo 128] [--hi 4736] [--step 64] [--out gpurun_out/dispatch_monotone.txt]"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402

lib = L.load()
dev = "cuda:0"
bf = torch.bfloat16
cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
L.check(lib.teo_gemm_workspace_init(ws.data_ptr(), cur), "ws")


def event_us(fn, iters=12, warm=2):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=128)
    ap.add_argument("--hi", type=int, default=4736)
    ap.add_argument("--step", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dispatch_monotone.txt"))
    ap.add_argument("--tol", type=float, default=0.08)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--tune", action="append", default=[], help="knob=value in the calling thread's tune block")
    ap.add_argument("--only", default="", help="comma-separated shape names (qkv,o,gateup,down,v.qkv,v.out,v.fc1,v.fc2)")
    args = ap.parse_args()
    shapes = (("qkv", 12288, 4096, 0, L.ACT_NONE, False, False), ("o", 4096, 4096, 0, L.ACT_NONE, False, True),
              ("gateup", 22016, 4096, L.GEMM_SWIGLU16, L.ACT_NONE, False, False), ("down", 4096, 11008, 0, L.ACT_NONE, False, True),
              ("v.qkv", 3072, 1024, 0, L.ACT_NONE, True, False), ("v.out", 1024, 1024, 0, L.ACT_NONE, True, True),
              ("v.fc1", 4096, 1024, 0, L.ACT_GELU_ERF, True, False), ("v.fc2", 1024, 4096, 0, L.ACT_NONE, True, True))
    if args.fp8:
        shapes = tuple(s for s in shapes if not s[0].startswith("v."))
    if args.only:
        shapes = tuple(s for s in shapes if s[0] in args.only.split(","))
    for kv in args.tune:
        k, v = kv.split("=")
        L.tune_set(k.encode(), int(v))
    Ms = list(range(args.lo, args.hi + 1, args.step))
    lines, flagged = [], []
    for name, N, K, flags, act, with_bias, with_res in shapes:
        Ws = [(torch.randn(N, K, device=dev) * 0.02).to(bf) for _ in range(max(8, -(-600_000_000 // (N * K * 2))))]       # > 2 x the 256 MB Infinity Cache: every launch streams its weights from HBM
        if args.fp8:
            Ws = [(W.float() * 100).to(torch.float8_e4m3fn).view(torch.uint8) for W in Ws]
            w_scale = torch.rand(N, device=dev) * 0.01 + 0.005
            a_scale = torch.rand(Ms[-1], device=dev) + 0.5
        Nc = N // 2 if flags else N
        bias = torch.randn(N, device=dev).to(bf) if with_bias else None
        Amax = torch.randn(Ms[-1], K, device=dev).to(bf)
        if args.fp8:
            Amax = (Amax.float() * 50).to(torch.float8_e4m3fn).view(torch.uint8)
        Cmax = torch.empty(Ms[-1], Nc, dtype=bf, device=dev)
        Rmax = torch.randn(Ms[-1], Nc, device=dev).to(bf) if with_res else None

        def time_M(M, trials=2):
            cnt = [0]

            def run():
                W = Ws[cnt[0] % len(Ws)]
                cnt[0] += 1
                if args.fp8:
                    L.check(lib.teo_gemm_fp8_ws(Amax.data_ptr(), a_scale.data_ptr(), W.data_ptr(), w_scale.data_ptr(),
                                                Rmax.data_ptr() if Rmax is not None else None, Cmax.data_ptr(), M, N, K, K, Nc, flags, L.TEO_BF16,
                                                ws.data_ptr(), cur), "gemm_fp8")
                    return
                L.check(lib.teo_gemm_ws(Amax.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        Rmax.data_ptr() if Rmax is not None else None, Cmax.data_ptr(), M, N, K, K, Nc, act, flags, L.TEO_BF16, L.TEO_BF16,
                                        ws.data_ptr(), cur), "gemm")
            run()
            kern = lib.teo_last_kernel().decode().replace("gemm_", "")
            return min(event_us(run) for _ in range(trials)), kern

        rows = []
        for M in Ms:
            t, kern = time_M(M)
            rows.append((M, t, kern))
        # a point that looks inverted is timed again, back to back with the larger row count it lost to (clock ramps between kernel variants
        # show up as one-off 10 % outliers): it stays flagged only if the repeat agrees
        for _ in range(2):
            best = (float("inf"), None)
            redo = []
            for i in range(len(rows) - 1, -1, -1):
                if best[1] is not None and rows[i][1] > (1.0 + args.tol) * best[0]:
                    redo.append((i, best[1]))
                if rows[i][1] < best[0]:
                    best = (rows[i][1], i)
            for i, j in redo:
                ti, _ = time_M(rows[i][0], 3)
                tj, _ = time_M(rows[j][0], 3)
                rows[i] = (rows[i][0], min(rows[i][1], ti), rows[i][2])
                rows[j] = (rows[j][0], min(rows[j][1], tj), rows[j][2])
        # inversions: t(M) against the minimum over all larger M
        suffix_min = [0.0] * len(rows)
        best = (float("inf"), None)
        for i in range(len(rows) - 1, -1, -1):
            suffix_min[i] = best
            if rows[i][1] < best[0]:
                best = (rows[i][1], rows[i][0])
        for i, (M, t, kern) in enumerate(rows):
            later, at = suffix_min[i]
            flag = ""
            if at is not None and t > (1.0 + args.tol) * later:
                flag = f"   <-- slower than M = {at} ({later:.1f} us) by {100 * (t / later - 1):.0f} %"
                flagged.append((name, M, t, kern, at, later))
            lines.append(f"{name:7s} M={M:5d} [{kern:18s}] {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TF/s{flag}")
        del Ws, Amax, Cmax, Rmax
    lines.append("")
    lines.append(f"inversions beyond {100 * args.tol:.0f} %: " + ("none" if not flagged else ""))
    for name, M, t, kern, at, later in flagged:
        lines.append(f"  {name} M={M} [{kern}] {t:.1f} us > M={at} {later:.1f} us")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
