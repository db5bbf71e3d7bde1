#!/usr/bin/env python3
"""The decode step over the context lengths a conversation can reach: ms per token for one conversation (the graph-replayed step of
config C3) and ms per step for B = 8 conversations, at ctx in {64 ... max_seq - 64}; against the line  t(ctx) = t_weights + ctx x kv_bytes /
rate  fitted through the two end points, a point more than 4 % above the line is a cliff (the attention split count, the chunk size or the
whole-context form changing at that length).  max_seq is the engine's (KV cache rows); the model is built once with max_seq = 4608.

usage (GPU box): python tools/ctx_sweep.py [--weights bf16|fp8] [--batch 8] [--out gpurun_out/ctx_sweep.json]
Reference path: the per-token forward inside model.generate (videollava/eval/inference.py:64-72 -> llava_llama.py:88-99 with past_key_values)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd.batch import BatchDecoder  # noqa: E402
from teochat_amd.builder import load_pretrained_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--max-seq", type=int, default=4608)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ctx_sweep.json"))
    args = ap.parse_args()
    dev, dtype = "cuda:0", torch.bfloat16
    tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=dev, dtype=dtype, max_seq=args.max_seq,
                                             weight_format=("fp8" if args.weights == "fp8" else None))
    eng, cfg = model.engine, model.config
    D = cfg.hidden_size
    ctxs = [c for c in (64, 128, 256, 384, 512, 640, 768, 1024, 1280, 1536, 1792, 2048, 2178, 2304, 2560, 3072, 3584, 4096, 4352, 4480)
            if c + args.steps + 8 <= args.max_seq]
    g = torch.Generator(device=dev).manual_seed(11)
    emb = torch.randn(max(ctxs), D, device=dev, generator=g).mul_(0.02).to(dtype)

    def one(ctx):
        best = None
        for rep in range(3):
            eng.reset_cache()
            lg = eng.prefill(emb[:ctx], last_only=True)
            eng.decode_begin(int(lg[0].argmax()))
            torch.cuda.synchronize()
            t = time.perf_counter()
            eng.decode_steps(args.steps)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) * 1e3 / args.steps
            if rep > 0:
                best = ms if best is None else min(best, ms)
        return best

    B = args.batch
    dec = BatchDecoder(eng, B, max_new=args.steps + 8)

    def many(ctx):
        best = None
        dec.reset()
        lg = dec.prefill_all([emb[:ctx]] * B)
        for rep in range(3):
            dec.cache_len = [ctx] * B
            dec.begin([int(lg[b].argmax()) for b in range(B)])
            torch.cuda.synchronize()
            t = time.perf_counter()
            dec.steps(args.steps)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) * 1e3 / args.steps
            if rep > 0:
                best = ms if best is None else min(best, ms)
        return best

    rows = []
    for ctx in ctxs:
        rows.append({"ctx": ctx, "ms_per_token": round(one(ctx), 4), f"ms_per_step_b{B}": round(many(ctx), 4)})
        print(rows[-1], flush=True)
    flags = []
    for key in ("ms_per_token", f"ms_per_step_b{B}"):
        (c0, t0), (c1, t1) = (rows[0]["ctx"], rows[0][key]), (rows[-1]["ctx"], rows[-1][key])
        for r in rows:
            line = t0 + (t1 - t0) * (r["ctx"] - c0) / (c1 - c0)
            r[key + "_over_line"] = round(r[key] / line, 4)
            if r[key] > 1.04 * line:
                flags.append(f"{key} at ctx {r['ctx']}: {r[key]} ms is {100 * (r[key] / line - 1):.1f} % above the end-point line ({line:.3f} ms)")
    out = {"weights": args.weights, "batch": B, "steps": args.steps, "max_seq": args.max_seq, "rows": rows, "flags": flags}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    md = [f"decode step vs context, {args.weights} weights, max_seq {args.max_seq}, {args.steps} graph-replayed steps from each context (best of 2)", "",
          f"| ctx | ms / token (1 conversation) | / line | ms / step (B = {B}) | / line |", "|---|---|---|---|---|"]
    md += [f"| {r['ctx']} | {r['ms_per_token']:.3f} | {r['ms_per_token_over_line']:.3f} | {r[f'ms_per_step_b{B}']:.3f} | {r[f'ms_per_step_b{B}_over_line']:.3f} |"
           for r in rows]
    md += ["", "flags: " + ("none" if not flags else "")] + [f"- {f}" for f in flags]
    open(os.path.splitext(args.out)[0] + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
