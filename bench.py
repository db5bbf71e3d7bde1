#!/usr/bin/env python3
"""Headline benchmark: end-to-end tokens/sec (prefill+decode), T=8 frames, LLaMA-2-7B (BASELINE.json metric).

One "step" = one conversation of config C3 through the whole hot path on one GPU:
    8 synthetic 224x224 frames -> CLIP-ViT-L/14 (23 layers) -> mlp2x_gelu projector -> image-token splice of a
    128-token prompt (L = 2168) -> LLaMA-2-7B prefill -> 256 forced greedy tokens (device-resident loop, hipGraph).
value = generated tokens of all ranks / wall time (inputs and weights resident in HBM before the timed region).
N > 1: one process per GPU, conversation-level data parallel replicas (weak scaling, no collective on the data
path; --shard-frames exercises the frame-sharded ViT + RCCL all-gather of config C4 instead).

Launch:  python bench.py --gpus 1 --steps 3 --warmup 1
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMAGE_TOKEN_INDEX = -200
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling


def synthetic_inputs(T, n_text, vocab, seed, device, dtype):
    """U{0..255} frames through the H6 normalisation; [BOS] + uniform ids with T sentinels."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.randint(0, 256, (T, 224, 224, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(1, 3, 1, 1)
    std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(1, 3, 1, 1)
    px = (raw.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    frames = [px[t].to(device=device, dtype=dtype) for t in range(T)]
    g2 = torch.Generator().manual_seed(seed + 1)
    ids = torch.randint(3, vocab, (n_text,), generator=g2, dtype=torch.long)
    ids[0] = 1
    lo, span = 8, n_text - 9
    for j in range(T):
        ids[lo + (j * span) // T] = IMAGE_TOKEN_INDEX
    assert int((ids == IMAGE_TOKEN_INDEX).sum()) == T
    return frames, ids.unsqueeze(0).to(device)


def cpu_baseline(T, n_text, n_out):
    """The oracle (CPU restatement of the reference path, torch-CPU fp32, all host cores) on a BOUNDED sample of the
    same workload: full-width model with 2 of 32 LLaMA layers and 2 of 23 ViT layers, the real L=2168 prefill and 4
    decode steps; extrapolated linearly in layers and tokens to the whole job."""
    from oracle import teo_oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    LV, LL = 2, 2
    vcfg = O.VitCfg(hidden_act="gelu", num_hidden_layers=24)
    lcfg = O.LlamaCfg(num_hidden_layers=LL)
    mm = O.MMCfg()
    sd = O.make_state_dict(vcfg, lcfg, mm, seed=2, vit_layers=LV, llm_layers=LL)
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, lcfg.vocab_size, seed=1).unsqueeze(0)
    with torch.no_grad():
        t0 = time.perf_counter()
        states = O.vit_hidden_states(torch.stack(frames), sd, vcfg, n_layers=LV)
        t_vit2 = time.perf_counter() - t0
        feats = states[-1][:, 1:]
        t0 = time.perf_counter()
        proj = O.projector(feats, sd, mm.mm_projector_type)
        t_proj = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, pos, mask, _, embeds, _ = O.prepare_inputs_labels_for_multimodal(
            ids, None, None, None, None, [proj[i] for i in range(T)], sd["model.embed_tokens.weight"], mm)
        t_splice = time.perf_counter() - t0
        L = embeds.shape[1]
        t0 = time.perf_counter()
        logits, cache = O.llama_forward(embeds, None, None, None, sd, lcfg, last_only=True)
        t_pre2 = time.perf_counter() - t0
        nd = 4
        t0 = time.perf_counter()
        for _ in range(nd):
            e = sd["model.embed_tokens.weight"][torch.tensor([[5]])]
            logits, cache = O.llama_forward(e, None, None, cache, sd, lcfg)
        t_dec = (time.perf_counter() - t0) / nd
        # lm_head + final norm are inside every llama_forward call once; time them alone to avoid scaling them by layers
        h = torch.randn(1, 1, lcfg.hidden_size)
        t0 = time.perf_counter()
        for _ in range(3):
            O.rmsnorm(h, sd["model.norm.weight"], 1e-5) @ sd["lm_head.weight"].t()
        t_head = (time.perf_counter() - t0) / 3
    t_vit = t_vit2 * 23.0 / LV
    t_prefill = (t_pre2 - t_head) * 32.0 / LL + t_head
    t_decode = ((t_dec - t_head) * 32.0 / LL + t_head) * (n_out - 1)
    total = t_vit + t_proj + t_splice + t_prefill + t_decode
    sample_s = t_vit2 + t_proj + t_splice + t_pre2 + t_dec * nd
    return {"value": n_out / total, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32, full-width layers, {LL}/32 LLaMA + {LV}/23 ViT layers, L={L} prefill, {nd} decode "
                      f"steps ({sample_s:.1f} s of CPU work); extrapolated linearly in layers and tokens",
            "est_phase_s": {"vit": round(t_vit, 2), "projector": round(t_proj, 3), "prefill": round(t_prefill, 2),
                            "decode": round(t_decode, 2)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--new", type=int, default=256)
    ap.add_argument("--shard-frames", action="store_true", help="C4 mode: one conversation, ViT frames sharded over ranks + all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"], help="fp8 = config C5 weight path (decode streams fp8-e4m3 weights); the headline is bf16")
    ap.add_argument("--batch", type=int, default=1, help="config C5 variant: B conversations per GPU decoded together (weights streamed once per step)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on real multi-GPU nodes; gloo for plumbing tests")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing test: every rank uses cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--tune", action="append", default=[], help="key=value passed to teo_tune_set (perf knobs only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from teochat_amd import _lib as L
    from teochat_amd.builder import load_pretrained_model
    T, n_text, n_out = args.frames, args.prompt, args.new
    Lseq = n_text - T + 256 * T
    max_seq = (Lseq + n_out + 255) // 256 * 256
    dtype = torch.bfloat16
    tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=device,
                                             dtype=dtype, max_seq=max_seq, weight_format=("fp8" if args.weights == "fp8" else None))
    eng = model.engine
    for kv in args.tune:
        k_, v_ = kv.split("=")
        L.check(eng.lib.teo_tune_set(k_.encode(), int(v_)), "teo_tune_set")
    frames, ids = synthetic_inputs(T, n_text, model.config.vocab_size, seed=100 * rank if not args.shard_frames else 0,
                                   device=device, dtype=dtype)

    if args.shard_frames and world > 1:
        from teochat_amd.parallel import sharded_frame_features
        tower_call = eng.vit_features
        model.get_model().image_tower.forward = lambda px: sharded_frame_features(tower_call, px)

    B = args.batch
    if B > 1:
        batch_in = [synthetic_inputs(T, n_text, model.config.vocab_size, seed=100 * rank + 10 + b, device=device, dtype=dtype)
                    for b in range(B)]

    def step():
        if B > 1:
            outs = model.generate_batch([i[0] for _, i in batch_in], [f for f, _ in batch_in], do_sample=False,
                                        max_new_tokens=n_out, eos_token_id=None, chunk=n_out)
            return torch.stack(outs)
        out = model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=n_out, eos_token_id=None,
                             chunk=n_out)
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    assert out.shape[1] == n_text + n_out
    tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    convs = args.steps * (1 if args.shard_frames else world) * B
    value = convs * n_out / dt

    # ---- untimed: phase breakdown of one more step
    phases = {}
    torch.cuda.synchronize()
    t = time.perf_counter()
    feats = eng.vit_features(torch.stack(frames)); torch.cuda.synchronize()
    phases["vit_ms"] = (time.perf_counter() - t) * 1e3; t = time.perf_counter()
    proj = eng.project(feats); torch.cuda.synchronize()
    phases["projector_ms"] = (time.perf_counter() - t) * 1e3; t = time.perf_counter()
    (_, _, _, _, emb, _) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, frames)
    torch.cuda.synchronize()
    phases["encode_plus_splice_ms"] = (time.perf_counter() - t) * 1e3
    eng.reset_cache(); t = time.perf_counter()
    lg = eng.prefill(emb[0], last_only=True); torch.cuda.synchronize()
    phases["prefill_ms"] = (time.perf_counter() - t) * 1e3
    eng.decode_begin(int(lg[0].argmax())); t = time.perf_counter()
    eng.decode_steps(n_out - 1, use_graph=not args.no_graph); torch.cuda.synchronize()
    phases["decode_ms"] = (time.perf_counter() - t) * 1e3
    phases["decode_ms_per_token"] = phases["decode_ms"] / (n_out - 1)
    phases["ttft_ms"] = phases["encode_plus_splice_ms"] + phases["prefill_ms"]      # frames in -> first token out
    if B > 1:
        dec = model._batch_decoder
        dec.begin([int(lg[0].argmax())] * B); torch.cuda.synchronize(); t = time.perf_counter()
        dec.steps(min(64, n_out - 1), use_graph=not args.no_graph); torch.cuda.synchronize()
        phases["batched_decode_ms_per_step"] = (time.perf_counter() - t) * 1e3 / min(64, n_out - 1)
        phases["batch"] = B
    phases = {k: round(v, 3) for k, v in phases.items()}

    # ---- roofline of the dominant kernel (decode gate/up GEMV: 43 % of the weight bytes of a token), HIP events
    # on the stream the kernel runs on, over the 32 layers' weights back to back (5.8 GB > L3, so no cache reuse)
    cfg = model.config
    Ws = eng.llama_w["gateup"]
    arr, pp = L.ptr_array([w.data_ptr() for w in Ws])
    x = torch.randn(cfg.hidden_size, device=device).to(dtype)
    y = torch.empty(cfg.intermediate_size, dtype=dtype, device=device)
    avg = C.c_float(0)
    with eng.phase() as st:
        L.check(eng.lib.teo_time_gemv_chain(x.data_ptr(), pp, None, len(Ws), eng.llama_w["post_norm"][0].data_ptr(), y.data_ptr(),
                                            2 * cfg.intermediate_size, cfg.hidden_size, cfg.rms_norm_eps,
                                            L.GEMM_SWIGLU16, L.TEO_BF16, 5, C.byref(avg), st), "teo_time_gemv_chain")
    gemv_bytes = 2 * cfg.intermediate_size * cfg.hidden_size * 2
    achieved = gemv_bytes / (avg.value * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_gemv_gateup.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = {"bound": "hbm", "kernel": "gemv_kernel<bf16,bf16,R=2,U=4,NT,SWIGLU> (decode rmsnorm + gate/up + SwiGLU, N=22016 K=4096)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": gemv_bytes, "avg_launch_ms": round(avg.value, 5)}
    # whole decode step against the HBM roofline (weights + KV per token)
    kv_ctx = Lseq + n_out / 2.0
    tok_bytes = 6.738e9 * (1 if args.weights == "fp8" else 2) + 2 * cfg.num_hidden_layers * cfg.num_key_value_heads * cfg.head_dim * 2 * kv_ctx
    # the MFMA-bound phases against the dense bf16 peak (algorithmic FLOPs of SURVEY.md section 8d)
    MFMA_PEAK_TFLOPS = 2500.0
    prefill_tf = (2.0 * Lseq * 6.476e9 + 2.0 * 4096 * 32000 + float(Lseq) ** 2 * 262144.0) / 1e12
    vit_tf = T * (155.3e9 + 10.74e9) / 1e12
    roofline["prefill_tflops"] = round(prefill_tf / (phases["prefill_ms"] * 1e-3), 1)
    roofline["prefill_frac_of_mfma_peak"] = round(roofline["prefill_tflops"] / MFMA_PEAK_TFLOPS, 4)
    roofline["vit_projector_tflops"] = round(vit_tf / (phases["encode_plus_splice_ms"] * 1e-3), 1)
    roofline["decode_step_frac_of_hbm_peak"] = round(tok_bytes / (phases["decode_ms_per_token"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    result = {
        "metric": "end-to-end tokens/sec (prefill+decode), T=8 frames, LLaMA-2-7B",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.weights == "bf16" else "bf16 activations / fp8-e4m3 decode weights", "data": "synthetic",
        "config": {"workload": f"{'C4' if args.shard_frames else ((f'C5-batched (B={B} conversations per GPU)' if B > 1 else 'C3') if T == 8 else 'C2-like')}: T={T} frames 224x224 -> CLIP-ViT-L/14 (23 layers) -> mlp2x_gelu -> splice of a "
                               f"{n_text}-token prompt (L={Lseq}) -> LLaMA-2-7B prefill -> {n_out} forced greedy tokens; "
                               f"value = generated tokens / total time",
                   "frames": T, "prompt_tokens": n_text, "sequence_len": Lseq, "new_tokens": n_out,
                   "parallelism": ("frame-sharded ViT + all-gather, replicated LLM" if args.shard_frames
                                   else f"dp{world} ({B} conversation{'s' if B > 1 else ''} per GPU, no collective)"),
                   "weights": "random N(0,0.02^2) at LLaMA-2-7B / ViT-L/14 shapes"},
        "total_tokens_per_s_incl_prompt": round(convs * (Lseq + n_out) / dt, 1),
        "phases": phases,
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(T, n_text, n_out)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
