#!/usr/bin/env python3
"""Throughput sweep of the MFMA phases over the shapes a caller can produce (VERDICT r05 "Next round" #2): the dispatch was tuned on
four row counts (M = 638 / 2056 / 2168 / 4208); this walks T x prompt and looks for cliffs between them.

    T in {1,2,3,4,5,6,8,12,16} frames  x  prompt in {32,128,512} tokens   (L = prompt - T + 256 T rows in the LLaMA prefill)

Per point: tower / projector / prefill milliseconds through the engine (the product path: teo_vit_encode, teo_projector, teo_llama_prefill,
median of 5 after a warm-up), prefill TFLOP/s and its fraction of the 2.5 PFLOP/s dense bf16 peak (SURVEY.md section 8d FLOP counts), and -- through
the public GEMM entry point with the same shapes, flags and a workspace, weights rotated so that they come from HBM -- the tile family the
dispatch picks for each LLaMA / tower GEMM (teo_last_kernel) with its time.  A point whose prefill_frac is below 0.85 x the linear interpolation
(in L) of its two neighbours in the same prompt column is flagged: a dispatch bug to fix.

usage (GPU box):  python tools/shape_sweep.py [--dtype bf16|fp16] [--out gpurun_out/shape_sweep.json] [--quick]
Reference path timed: videollava/model/multimodal_encoder/languagebind/image/modeling_image.py:136-151 (ViT layer),
videollava/model/language_model/llava_llama.py:88-99 (LLaMA forward), called from videollava/eval/inference.py:64-72."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from teochat_amd.builder import load_pretrained_model  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0


def wall_ms(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shape_sweep.json"))
    ap.add_argument("--quick", action="store_true", help="T in {2, 8, 16} only")
    ap.add_argument("--no-gemms", action="store_true")
    args = ap.parse_args()
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dt = L.TEO_F16 if args.dtype == "fp16" else L.TEO_BF16
    dev = "cuda:0"
    Ts = (2, 8, 16) if args.quick else (1, 2, 3, 4, 5, 6, 8, 12, 16)
    prompts = (32, 128, 512)
    max_seq = (max(prompts) + 255 * max(Ts) + 64 + 255) // 256 * 256
    tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=dev, dtype=dtype, max_seq=max_seq)
    eng, cfg, lib = model.engine, model.config, L.load()
    D, F, H, hd = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim
    QKV = (H + 2 * cfg.num_key_value_heads) * hd
    import ctypes as C
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
    cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.teo_gemm_workspace_init(ws.data_ptr(), cur), "ws init")

    # one set of rotating weights per GEMM shape (8 copies > the 256 MB Infinity Cache for every LLaMA shape)
    def weights(N, K, n=8):
        return [(torch.randn(N, K, device=dev) * 0.02).to(dtype) for _ in range(n)]
    gemm_w = {}
    llama_shapes = (("qkv", QKV, D, 0, False), ("o", D, H * hd, 0, True), ("gateup", 2 * F, D, L.GEMM_SWIGLU16, False), ("down", D, F, 0, True))
    Dv, Fv = 1024, 4096
    tower_shapes = (("v.qkv", 3 * Dv, Dv, L.ACT_NONE, True, False), ("v.out", Dv, Dv, L.ACT_NONE, True, True),
                    ("v.fc1", Fv, Dv, L.ACT_GELU_ERF, True, False), ("v.fc2", Dv, Fv, L.ACT_NONE, True, True))

    def time_gemm(name, M, N, K, flags, act, with_bias, with_res):
        key = (N, K)
        if key not in gemm_w:
            gemm_w[key] = weights(N, K, 8 if N * K * 2 > 16e6 else 16)
        Ws = gemm_w[key]
        A = torch.randn(M, K, device=dev).to(dtype)
        Nc = N // 2 if (flags & L.GEMM_SWIGLU16) else N
        Cc = torch.empty(M, Nc, dtype=dtype, device=dev)
        res = torch.randn(M, Nc, device=dev).to(dtype) if with_res else None
        bias = torch.randn(N, device=dev).to(dtype) if with_bias else None
        cnt = [0]

        def run():
            W = Ws[cnt[0] % len(Ws)]
            cnt[0] += 1
            L.check(lib.teo_gemm_ws(A.data_ptr(), W.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    res.data_ptr() if res is not None else None, Cc.data_ptr(), M, N, K, K, Nc, act, flags, dt, dt,
                                    ws.data_ptr(), cur), "teo_gemm_ws")
        run()
        kern = lib.teo_last_kernel().decode()
        us = min(event_us(run) for _ in range(2))
        flops = 2.0 * M * N * K
        return {"kernel": kern.replace("gemm_", ""), "us": round(us, 1), "tflops": round(flops / us / 1e6, 0)}

    rows = []
    for n_text in prompts:
        for T in Ts:
            if n_text <= T + 9:
                continue
            Lseq = n_text - T + 256 * T
            px = torch.randn(T, 3, 224, 224, device=dev).to(dtype)
            emb = (torch.randn(Lseq, D, device=dev) * 0.02).to(dtype)
            feats = eng.vit_features(px)
            r = {"T": T, "prompt": n_text, "L": Lseq}
            r["tower_ms"] = round(wall_ms(lambda: eng.vit_features(px)), 3)
            r["projector_ms"] = round(wall_ms(lambda: eng.project(feats)), 3)

            def prefill():
                eng.reset_cache()
                eng.prefill(emb, last_only=True)
            r["prefill_ms"] = round(wall_ms(prefill), 3)
            tf = (2.0 * Lseq * 6.476e9 + 2.0 * 4096 * 32000 + float(Lseq) ** 2 * 262144.0) / 1e12
            r["prefill_tflops"] = round(tf / (r["prefill_ms"] * 1e-3), 1)
            r["prefill_frac"] = round(r["prefill_tflops"] / MFMA_PEAK_TFLOPS, 4)
            r["tower_tflops"] = round(T * 155.3e9 / 1e12 / (r["tower_ms"] * 1e-3), 1)
            r["ttft_ms"] = round(r["tower_ms"] + r["projector_ms"] + r["prefill_ms"], 3)
            if not args.no_gemms:
                r["gemms"] = {nm: time_gemm(nm, Lseq, N, K, fl, L.ACT_NONE, False, wr) for nm, N, K, fl, wr in llama_shapes}
                r["tower_gemms"] = {nm: time_gemm(nm, T * 257, N, K, 0, act, wb, wr) for nm, N, K, act, wb, wr in tower_shapes}
            rows.append(r)
            g = "  ".join(f"{k}[{v['kernel']}] {v['us']:.0f}us" for k, v in r.get("gemms", {}).items())
            print(f"T={T:2d} prompt={n_text:3d} L={Lseq:4d}: tower {r['tower_ms']:.2f} ms, prefill {r['prefill_ms']:.2f} ms "
                  f"({r['prefill_tflops']:.0f} TF/s, frac {r['prefill_frac']:.3f})  {g}", flush=True)
    # cliffs: a point under 0.85 x the linear interpolation (in L) of its neighbours in the same prompt column
    cliffs = []
    for n_text in prompts:
        col = sorted((r for r in rows if r["prompt"] == n_text), key=lambda r: r["L"])
        for i in range(1, len(col) - 1):
            a, b, c = col[i - 1], col[i], col[i + 1]
            w = (b["L"] - a["L"]) / float(c["L"] - a["L"])
            for key in ("prefill_frac", "tower_tflops"):
                interp = a[key] * (1 - w) + c[key] * w
                if b[key] < 0.85 * interp:
                    cliffs.append({"T": b["T"], "prompt": n_text, "L": b["L"], "metric": key, "value": b[key], "interpolated": round(interp, 4),
                                   "ratio": round(b[key] / interp, 3)})
    out = {"dtype": args.dtype, "rows": rows, "cliffs": cliffs}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    md = args.out.replace(".json", ".md")
    with open(md, "w") as f:
        f.write(f"# shape sweep ({args.dtype}): T x prompt -> tower / prefill, tile family per GEMM (tools/shape_sweep.py)\n\n")
        f.write("| T | prompt | L | tower ms | tower TF/s | proj ms | prefill ms | prefill TF/s | prefill_frac | TTFT ms | qkv | o | gate/up | down | v.qkv | v.out | v.fc1 | v.fc2 |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            cell = lambda d: f"{d['kernel']} {d['us']:.0f} us ({d['tflops']:.0f})" if d else ""
            gs = [cell(r.get("gemms", {}).get(k)) for k in ("qkv", "o", "gateup", "down")]
            ts = [cell(r.get("tower_gemms", {}).get(k)) for k in ("v.qkv", "v.out", "v.fc1", "v.fc2")]
            f.write(f"| {r['T']} | {r['prompt']} | {r['L']} | {r['tower_ms']:.2f} | {r['tower_tflops']:.0f} | {r['projector_ms']:.2f} | {r['prefill_ms']:.2f} | "
                    f"{r['prefill_tflops']:.0f} | {r['prefill_frac']:.3f} | {r['ttft_ms']:.2f} | " + " | ".join(gs + ts) + " |\n")
        f.write("\n## cliffs (value < 0.85 x the linear interpolation of the neighbours in the same prompt column)\n\n")
        if cliffs:
            for c in cliffs:
                f.write(f"* T={c['T']} prompt={c['prompt']} (L={c['L']}): {c['metric']} {c['value']} vs interpolated {c['interpolated']} (x{c['ratio']})\n")
        else:
            f.write("none\n")
    print("cliffs:", cliffs or "none")
    print("wrote", args.out, md)


if __name__ == "__main__":
    main()
