#!/usr/bin/env python3
"""The batched decode step over the batch sizes a caller can ask for (config C5: B conversations per GPU share one weight stream):
ms per step, conversations x tokens per second and the step's fraction of the HBM peak (weights once + B x KV per step) for
B in {1,2,3,4,6,8,9,12,16} at the headline context (2178 rows per conversation), bf16 or fp8 weights.  Two properties are checked
and flagged: a step with MORE conversations cannot be faster (beyond 3 %), and tokens/s cannot fall as B grows.  The skinny-GEMM
dispatch (tile kernel vs persistent stream kernel, unroll per row count) was tuned on B = 8 and 16; this walks what lies between.

usage (GPU box): python tools/batch_sweep.py [--weights bf16|fp8] [--ctx 2178] [--out gpurun_out/batch_sweep.json]
Reference path: one model.generate call per conversation (videollava/eval/inference.py:64-72); the reference has no batched decode of
several conversations with different images -- this is the serving shape of BASELINE.json configs[4]."""
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

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--ctx", type=int, default=2178)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--batches", default="1,2,3,4,6,8,9,12,16")
    ap.add_argument("--tune", action="append", default=[], help="knob=value (teo_tune), e.g. attn_whole=0")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "batch_sweep.json"))
    args = ap.parse_args()
    dev, dtype = "cuda:0", torch.bfloat16
    max_seq = (args.ctx + args.steps + 64 + 255) // 256 * 256
    tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=dev, dtype=dtype, max_seq=max_seq,
                                             weight_format=("fp8" if args.weights == "fp8" else None))
    eng, cfg = model.engine, model.config
    for kv in args.tune:
        k, v = kv.split("=")
        eng.tune_set(k, int(v))
    D, F, V, Lr = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.num_hidden_layers
    qkv = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
    w_bytes = (Lr * (qkv * D + D * D + 3 * F * D) + V * D) * (1 if args.weights == "fp8" else 2)
    kv_bytes = 2 * Lr * cfg.num_key_value_heads * cfg.head_dim * 2          # per cached position, K and V, 16-bit
    g = torch.Generator(device=dev).manual_seed(7)
    rows = []
    for B in [int(b) for b in args.batches.split(",")]:
        dec = BatchDecoder(eng, B, max_new=args.steps + 8)
        seqs = [torch.randn(args.ctx, D, device=dev, generator=g).mul_(0.02).to(dtype) for _ in range(B)]
        lg = dec.prefill_all(seqs)
        best = None
        for rep in range(3):
            dec.cache_len = [args.ctx] * B
            dec.begin([int(lg[b].argmax()) for b in range(B)])
            torch.cuda.synchronize()
            t = time.perf_counter()
            dec.steps(args.steps)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) * 1e3 / args.steps
            best = ms if best is None or (rep > 0 and ms < best) else best          # rep 0 includes the graph capture
            if rep == 0:
                best = None
        step_bytes = w_bytes + B * kv_bytes * (args.ctx + args.steps / 2)
        rows.append({"batch": B, "ms_per_step": round(best, 4), "tokens_per_s": round(B / best * 1e3, 1),
                     "frac_of_hbm_peak": round(step_bytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        print(rows[-1], flush=True)
        del dec, seqs
        torch.cuda.empty_cache()
    flags = []
    for i, r in enumerate(rows):
        for q in rows[i + 1:]:
            if r["ms_per_step"] > 1.03 * q["ms_per_step"]:
                flags.append(f"B={r['batch']} step {r['ms_per_step']} ms > B={q['batch']} step {q['ms_per_step']} ms")
            if r["tokens_per_s"] > 1.0 * q["tokens_per_s"]:
                flags.append(f"B={r['batch']} {r['tokens_per_s']} tok/s > B={q['batch']} {q['tokens_per_s']} tok/s")
    out = {"weights": args.weights, "tune": args.tune, "ctx": args.ctx, "steps": args.steps, "rows": rows, "flags": flags}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    md = [f"batched decode step, {args.weights} weights{(' ' + ' '.join(args.tune)) if args.tune else ''}, {args.ctx} cached rows per conversation, {args.steps} graph-replayed steps (best of 2)", "",
          "| B | ms / step | conversations x tok/s | step bytes / time as a fraction of 8 TB/s |", "|---|---|---|---|"]
    md += [f"| {r['batch']} | {r['ms_per_step']:.3f} | {r['tokens_per_s']:.0f} | {r['frac_of_hbm_peak']:.3f} |" for r in rows]
    md += ["", "flags: " + ("none" if not flags else "")] + [f"- {f}" for f in flags]
    open(os.path.splitext(args.out)[0] + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
