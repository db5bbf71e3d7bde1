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
import threading
import time

_T_PROCESS_START = time.perf_counter()      # before `import torch`: setup_s below is what a rank spends before it can take its first step

import torch  # noqa: E402

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


def workload_label(T, n_text, n_out, B, shard_frames, weights):
    """Which BASELINE.json configuration a run corresponds to."""
    if shard_frames:
        return "C4 (frame-sharded tower + all-gather)" if T == 16 else f"C4-style frame sharding at T={T}"
    if B > 1:
        return f"C5-batched (B={B} conversations per GPU{', fp8 decode weights' if weights == 'fp8' else ''})"
    base = {2: "C2", 8: "C3", 16: "C4 full-length, unsharded tower"}.get(T, f"T={T} variant")
    if T == 2 and n_out != 128:
        base = "C2 shapes"
    if weights == "fp8":
        base += " with fp8 decode weights (C5 single conversation)"
    return base


def _cpu_info():
    """(model name, hardware threads, CPUs this process may actually use: cgroup quota / affinity)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    hw = os.cpu_count() or 1
    eff = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(hw)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()          # cgroup v2: "<quota> <period>" or "max <period>"
        if q != "max":
            eff = min(eff, float(q) / float(per))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                eff = min(eff, q / per)
        except (OSError, ValueError):
            pass
    return model, hw, max(1, int(round(eff)))


def cpu_baseline(T, n_text, n_out, budget_s=240.0, max_layers=None):
    """The reference CPU path (oracle/teo_oracle.py: torch-CPU restatement of H6-H16, validated against the reference's own
    outputs) timed on this host at the true LLaMA-2-7B / ViT-L/14 shapes, KV cache pre-allocated, in fp32 and in bf16 (the
    reference's CPU dtype is 16-bit: builder.py:105, inference.py:53):

      C1  T=2 frames, 32-token prompt (L=542), 8 new tokens -- the whole job, every layer (fp32)
      C3  T=8, 128-token prompt (L=2168): ViT (23 layers) + projector + splice + FULL-DEPTH prefill + 8 decode steps; the
          decode phase is extrapolated linearly from those 8 steps to n_out - 1

    Threads: the box may give this process fewer CPUs than it has (cgroup cpu.max); thread counts around that quota are swept
    on a one-layer probe of each phase and the best is used.  `value` is the faster of the two dtype legs.  On a host too
    slow for the full-depth sample inside `budget_s` the LLaMA depth is reduced and the result says so."""
    from oracle import teo_oracle as O
    cpu_model, hw, eff = _cpu_info()
    vcfg = O.VitCfg(hidden_act="gelu", num_hidden_layers=24)
    mm = O.MMCfg()
    lcfg1 = O.LlamaCfg(num_hidden_layers=1)
    L3 = n_text - T + 256 * T
    t_start = time.perf_counter()

    def timed(fn, reps=1):
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        return (time.perf_counter() - t0) / reps, r

    with torch.no_grad():
        # ---- probe: one LLaMA layer, prefill at L3 and decode at ctx L3, per thread count
        torch.set_num_threads(eff)
        sd1 = O.make_state_dict(vcfg, lcfg1, mm, seed=2, vit_layers=1, llm_layers=1)
        x_pre = torch.randn(1, L3, lcfg1.hidden_size) * 0.02
        x_dec = torch.randn(1, 1, lcfg1.hidden_size) * 0.02
        cands = sorted({c for c in (eff, 2 * eff, 32, 64, 128, hw) if eff <= c <= min(hw, 4 * eff)} | {eff})
        probe = {}
        for nt in cands:
            torch.set_num_threads(nt)
            cache = O.KVCache(capacity=L3 + 8)
            tp, _ = timed(lambda: O.llama_forward(x_pre, None, None, cache, sd1, lcfg1, last_only=True))
            O.llama_forward(x_dec, None, None, cache, sd1, lcfg1)                  # warm the decode shapes
            td, _ = timed(lambda: O.llama_forward(x_dec, None, None, cache, sd1, lcfg1), reps=3)
            probe[nt] = (tp, td)
            if time.perf_counter() - t_start > 0.15 * budget_s:
                break
        nt_pre = min(probe, key=lambda k: probe[k][0])
        nt_dec = min(probe, key=lambda k: probe[k][1])
        t_layer_pre, t_layer_dec = probe[nt_pre][0], probe[nt_dec][1]
        del cache
        # ---- depth that fits the budget (32 on the MI355X host; fewer only on a slow machine, then extrapolated):
        # fp32 legs = C1 (prefill at L=542 + 7 steps) + C3 (prefill at L3 + 7 steps); the bf16 leg costs about a third of that
        LL = 32 if max_layers is None else max_layers
        est = lambda n: 1.35 * n * (t_layer_pre * (1.0 + 542.0 / L3) + 14 * t_layer_dec)
        while LL > 2 and est(LL) > 0.6 * budget_s:
            LL //= 2
        lcfg = O.LlamaCfg(num_hidden_layers=LL)
        torch.set_num_threads(nt_pre)
        t_w, sd = timed(lambda: O.make_state_dict_for_timing(vcfg, lcfg, mm, seed=2, base=sd1))
        del sd1
        scale_layers = 32.0 / LL
        # untimed warm-up: one decode step over every layer (first touch of all weight pages, BLAS thread pools)
        O.llama_forward(x_dec, None, None, O.KVCache(capacity=4), sd, lcfg)

        def run(sdd, Tn, n_txt, n_new, dtype=torch.float32):
            """One job; returns phase seconds (decode: seconds per token over n_new - 1 steps)."""
            ids = O.synthetic_prompt_ids(n_txt, Tn, lcfg.vocab_size, seed=1).unsqueeze(0)
            frames = [f.to(dtype) for f in O.synthetic_frames(Tn, 224, seed=0)]
            torch.set_num_threads(nt_pre)
            ph = {}
            ph["vit"], feats = timed(lambda: O.vit_features(torch.stack(frames), sdd, vcfg, -2, "patch"))
            ph["projector"], proj = timed(lambda: O.projector(feats, sdd, mm.mm_projector_type))
            emb_w = sdd["model.embed_tokens.weight"]
            ph["splice"], r = timed(lambda: O.prepare_inputs_labels_for_multimodal(ids, None, None, None, None,
                                                                                   [proj[i] for i in range(Tn)], emb_w, mm))
            embeds = r[4]
            Lq = embeds.shape[1]
            cache = O.KVCache(capacity=Lq + n_new)
            ph["prefill"], (logits, _) = timed(lambda: O.llama_forward(embeds, None, None, cache, sdd, lcfg, last_only=True))
            toks = [int(logits[0, -1].argmax())]
            torch.set_num_threads(nt_dec)
            t0 = time.perf_counter()
            for _ in range(n_new - 1):
                e = emb_w[torch.tensor([[toks[-1]]])]
                logits, _ = O.llama_forward(e, None, None, cache, sdd, lcfg)
                toks.append(int(logits[0, -1].argmax()))
            ph["decode_per_token"] = (time.perf_counter() - t0) / max(n_new - 1, 1)
            ph["L"] = Lq
            return ph

        def total(ph, n_new):
            """Whole-job seconds at full depth: the LLaMA phases scale with 32 / LL when the depth was reduced (at LL = 32
            nothing is scaled)."""
            return (ph["vit"] + ph["projector"] + ph["splice"] + ph["prefill"] * scale_layers
                    + ph["decode_per_token"] * scale_layers * (n_new - 1))

        def leg(ph, bytes_per_w):
            return {"tokens_per_s": round(n_out / total(ph, n_out), 4),
                    "phase_s": {"vit": round(ph["vit"], 3), "projector": round(ph["projector"], 3), "splice": round(ph["splice"], 4),
                                "prefill": round(ph["prefill"] * scale_layers, 3),
                                "decode_per_token": round(ph["decode_per_token"] * scale_layers, 4)},
                    "prefill_tflops": round((2.0 * ph["L"] * 6.476e9 + float(ph["L"]) ** 2 * 262144.0) / 1e12 / (ph["prefill"] * scale_layers), 3),
                    "decode_weight_stream_GBps": round(6.607e9 * bytes_per_w / (ph["decode_per_token"] * scale_layers) / 1e9, 1)}

        c1 = run(sd, 2, 32, 8)
        c3 = run(sd, T, n_text, 8)
        legs = {"fp32": leg(c3, 4)}
        # what a streaming read gets on the CPUs this process may use: the host-side roofline of the decode phase
        torch.set_num_threads(nt_pre)
        probe_t = torch.ones(128 * 2 ** 20, dtype=torch.float32)
        probe_t.sum()
        t_rd, _ = timed(lambda: probe_t.sum(), reps=3)
        host_read_gbps = probe_t.numel() * 4 / t_rd / 1e9
        del probe_t
        try:
            sd16 = {k: v.to(torch.bfloat16) for k, v in sd.items()}
            del sd
            x16 = x_dec.to(torch.bfloat16)
            best16 = None
            for nt in probe:                                  # the bf16 one-row products thread differently from the fp32 ones
                torch.set_num_threads(nt)
                O.llama_forward(x16, None, None, O.KVCache(capacity=4), sd16, lcfg)
                t16, _ = timed(lambda: O.llama_forward(x16, None, None, O.KVCache(capacity=4), sd16, lcfg), reps=2)
                if best16 is None or t16 < best16[0]:
                    best16 = (t16, nt)
            nt_dec = best16[1]
            legs["bf16"] = leg(run(sd16, T, n_text, 8, dtype=torch.bfloat16), 2)
            legs["bf16"]["decode_threads"] = nt_dec
        except Exception as e:  # noqa: BLE001
            legs["bf16"] = {"error": str(e)[:200]}
        best = max((k for k in legs if "tokens_per_s" in legs[k]), key=lambda k: legs[k]["tokens_per_s"])
        out = {
            "value": legs[best]["tokens_per_s"], "unit": "tokens/s", "cores": nt_dec, "kind": "port", "dtype_of_value": best,
            "cpu": cpu_model, "hw_threads": hw, "cpus_available_to_this_process": eff,
            "host_stream_read_GBps": round(host_read_gbps, 1),
            "threads": {"prefill_and_vit": nt_pre, "decode": nt_dec,
                        "swept_one_layer_s": {str(k): [round(v[0], 3), round(v[1], 4)] for k, v in probe.items()}},
            "sample": (f"oracle (torch-CPU), true 7B / ViT-L shapes, KV cache pre-allocated; C3: ViT 23 layers + projector + splice + "
                       f"{LL}/32-layer prefill at L={c3['L']} + 8 decode steps"
                       + ("" if LL == 32 else f" (LLaMA phases scaled x{scale_layers:g}: host too slow for full depth in {budget_s:.0f} s)")
                       + f"; decode extrapolated linearly from 8 to {n_out} tokens; fp32 and bf16 legs, value = the faster ({best}); "
                         f"weights: one drawn layer + rolled copies ({t_w:.1f} s to build)"),
            "legs": legs,
            "c1_full_job_fp32": {"workload": "C1: T=2, 32-token prompt (L=542), 8 new tokens, every phase in full",
                                 "tokens_per_s": round(8 / total(c1, 8), 4), "seconds": round(total(c1, 8), 3),
                                 "phase_s": {k: round(v * (scale_layers if k in ("prefill", "decode_per_token") else 1.0), 4)
                                             for k, v in c1.items() if k != "L"}},
        }
        out["wall_s"] = round(time.perf_counter() - t_start, 1)
        # `value` is ONE box's host: the gpurun boxes grant 16 of 256 hardware threads and differ among themselves.  The same leg as
        # the committed lines of earlier rounds measured it (profiles/r0N_bench.json, the driver's BENCH_r0N.json; round 1 timed a 2-layer
        # sample with 256 threads and is not comparable), so that the number is not read as a constant:
        seen = []
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "BENCH_r0*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json"))):
            try:
                blob = json.load(open(path))
                cb = (blob.get("parsed") or blob).get("cpu_baseline") or {}
                if cb.get("value", 0) > 0.5:
                    seen.append((round(float(cb["value"]), 3), os.path.relpath(path, ROOT)))
            except Exception:  # noqa: BLE001
                pass
        vals = [v for v, _ in seen] + [out["value"]]
        out["range_across_boxes"] = {"tokens_per_s": [min(vals), max(vals)], "n_boxes": len(vals),
                                     "from": [f"{n}: {v}" for v, n in seen] + [f"this run: {out['value']}"],
                                     "note": "same sample and method on every box (rounds 2-6); the spread is the hosts' (16 granted CPUs of a 2 x 64-core EPYC 9575F)"}
    return out


class Deadline:
    """A hard deadline around one collective / rendezvous step of an N > 1 run: if the block has not finished after `seconds`, the rank
    says which step hung and leaves with exit code 3 -- the launcher then ends the job with a non-zero code and no JSON line, instead
    of the driver's clock running out on a silent hang.  (An exit, never an exec: this process holds the GPU.)"""

    def __init__(self, seconds, what, rank):
        self.seconds, self.what, self.rank, self.timer = float(seconds), what, rank, None

    def _fire(self):
        sys.stderr.write(f"bench.py: rank {self.rank}: '{self.what}' did not finish within {self.seconds:.0f} s -- a rank is missing or the "
                         f"fabric is stuck; exiting with code 3\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *a):
        if self.timer is not None:
            self.timer.cancel()
        return False


def spawn_ranks(args):
    """One child `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` (one rank per GPU, RCCL); returns its
    exit code.  Fails loudly when the node has fewer GPUs than ranks (unless --same-gpu, the one-GPU plumbing test)."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.same_gpu:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node shows {n_dev} GPU(s); refusing to report a line for fewer ranks\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="16-bit storage / compute format: bf16 = BASELINE.json's headline; "
                    "fp16 = the reference's own inference dtype (builder.py:105), same kernels on v_mfma_*_f16 / v_dot2_f32_f16")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"], help="fp8 = config C5 weight path (decode streams fp8-e4m3 weights); the headline is bf16")
    ap.add_argument("--prefill", default="auto", choices=["auto", "bf16", "fp8"], help="prefill Linear layers: bf16 MFMA on the (dequantised) weights, or w8a8 on the fp8 MFMA (default with --weights fp8)")
    ap.add_argument("--batch", type=int, default=1, help="config C5 variant: B conversations per GPU decoded together (weights streamed once per step)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on real multi-GPU nodes; gloo for plumbing tests")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing test: every rank uses cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--rccl-timeout", type=float, default=180.0, help="deadline (s) of the once-per-run RCCL communicator check")
    ap.add_argument("--collective-timeout", type=float, default=900.0, help="N > 1: hard deadline (s) of every rendezvous / barrier / reduction of the "
                    "run, the timed region's barriers included; a rank that hits it exits with code 3 (0 = no deadline)")
    ap.add_argument("--tune", action="append", default=[], help="key=value set in the engine's teo_tune block (perf knobs only)")
    ap.add_argument("--rope-in-attn", type=int, default=None, choices=[0, 1], help="A/B of the engine option: RoPE + KV append inside the decode attention kernel (1) or in the QKV GEMV epilogue (0, the single-conversation default); same values")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks here.  This parent has not touched the GPU (importing
        # torch and counting devices does not initialise HIP), it never execs, and it exits with the launcher's code.
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    import torch.distributed as dist
    import datetime
    deadline = lambda what: Deadline(args.collective_timeout if world > 1 else 0, what, rank)      # noqa: E731
    host_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pg_timeout = datetime.timedelta(seconds=max(args.collective_timeout, 60.0))
        with deadline("init_process_group"):
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device), timeout=pg_timeout)
            else:
                dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=pg_timeout)
        # host-side exchanges (the communicator's unique id, the per-rank statistics of the line) go over their OWN gloo group: they
        # never touch the GPU and never interleave with the timed region's barrier / reductions on the default group
        chk_group = None
        with deadline("new_group(gloo)"):
            try:
                host_group = dist.new_group(backend="gloo", timeout=pg_timeout)
                chk_group = dist.new_group(backend="gloo", timeout=pg_timeout)      # the RCCL check's own: a helper thread that never comes
            except Exception:  # noqa: BLE001 -- no gloo in this build: the default group carries the host objects too      back must not hold host_group
                host_group = chk_group = None

    if os.environ.get("TEO_BENCH_FAIL_RANK") == str(rank) and world > 1:
        # test hook (tests/test_bench_contract_gpu.py): a rank other than 0 dies -> the launcher must end the job with a non-zero code
        raise SystemExit(f"rank {rank}: TEO_BENCH_FAIL_RANK set")
    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit(f"{dist.get_world_size()} ranks joined the process group, --gpus says {args.gpus}")
    from teochat_amd import _lib as L
    from teochat_amd.builder import load_pretrained_model
    T, n_text, n_out = args.frames, args.prompt, args.new
    Lseq = n_text - T + 256 * T
    max_seq = (Lseq + n_out + 255) // 256 * 256
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    if args.dtype == "fp16" and args.weights == "fp8":
        raise SystemExit("--weights fp8 goes with --dtype bf16 (the e4m3 row scales are exact in bfloat16 only)")
    tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=device,
                                             dtype=dtype, max_seq=max_seq, weight_format=("fp8" if args.weights == "fp8" else None))
    eng = model.engine
    prefill_fp8 = args.prefill == "fp8" or (args.prefill == "auto" and args.weights == "fp8")
    if prefill_fp8:
        if args.weights != "fp8":
            raise SystemExit("--prefill fp8 needs --weights fp8 (the e4m3 weight copies)")
        eng.set_options(prefill_fp8=True)
    if args.rope_in_attn is not None:
        eng.set_options(rope_in_attn=bool(args.rope_in_attn))
    for kv in args.tune:
        k_, v_ = kv.split("=")
        eng.tune_set(k_, int(v_))                # the engine's own teo_tune block (nothing process-wide)
    frames, ids = synthetic_inputs(T, n_text, model.config.vocab_size, seed=100 * rank if not args.shard_frames else 0,
                                   device=device, dtype=dtype)
    torch.cuda.synchronize()
    # process start -> model resident and inputs on the device.  The synthetic weights are seed-generated ON this rank's GPU
    # (teochat_amd/synthetic.py: a device generator, no host copy of the 14 GB), so N ranks do not compete for the job's CPUs here
    setup_s = time.perf_counter() - _T_PROCESS_START

    # The library's own RCCL communicator (teo_ctx_create over all ranks): the data path of the frame-sharded tower, and for the
    # data-parallel replicas a once-per-run proof that N ranks really hold one communicator over xGMI (`rccl_ranks` below) plus
    # one checked teo_allgather_visual of a C4-sized block (2 frames x 256 x 1024 bf16 = 1 MiB per rank), outside the timed region.
    comm, rccl_info, th = None, None, None
    if world > 1 and args.dist_backend == "nccl":
        from teochat_amd.parallel import TeoComm
        import threading
        box = {}
        # the check's host-side exchange (the communicator's unique id) runs on its OWN gloo group: if the helper thread is still inside
        # it at the deadline, the timed region's barrier / all_reduce on the default group stay ordered across ranks

        def rccl_check():
            # runs on a helper thread with a deadline: a communicator that never forms (a rank missing, a fabric fault) must not
            # hang the throughput line of the data-parallel replicas, which does not depend on it
            try:
                torch.cuda.set_device(local_rank)
                c = TeoComm(rank, world, local_rank, group=chk_group)
                r_, w_, cu_, hbm_ = c.info()
                send = torch.full((512, 1024), float(rank + 1), dtype=dtype, device=device)
                recv = torch.zeros(world * 512, 1024, dtype=dtype, device=device)
                c.all_gather_rows(send, recv); torch.cuda.synchronize()
                t_ag = time.perf_counter()
                for _ in range(20):
                    c.all_gather_rows(send, recv)
                torch.cuda.synchronize()
                t_ag = (time.perf_counter() - t_ag) / 20
                want = torch.arange(1, world + 1, dtype=torch.float32, device=device).repeat_interleave(512)
                ok = bool((recv.float().amax(dim=1) == want).all()) and bool((recv.float().amin(dim=1) == want).all())
                box["info"] = {"rccl_ranks": w_, "cu_count": cu_, "hbm_bytes": hbm_, "allgather_1MiB_per_rank_us": round(t_ag * 1e6, 1),
                               "allgather_checked": ok}
                if not ok:
                    raise RuntimeError("teo_allgather_visual returned wrong rows")
                box["comm"] = c
            except Exception as e:  # noqa: BLE001 -- report, do not hide
                box["error"] = e

        th = threading.Thread(target=rccl_check, daemon=True)
        th.start()
        th.join(timeout=args.rccl_timeout)
        if th.is_alive():
            box["error"] = TimeoutError(f"the RCCL communicator check did not finish in {args.rccl_timeout:.0f} s")
        if "error" in box:
            if args.shard_frames:
                raise box["error"]
            e = box["error"]
            rccl_info = {"rccl_ranks": None, "error": f"{type(e).__name__}: {str(e)[:300]}"}
            comm = None
        else:
            comm, rccl_info = box["comm"], box["info"]
    if args.shard_frames:
        # C4: every rank encodes its block of frames; with the nccl backend the gather is the library's RCCL all-gather
        # (teo_allgather_visual, no torch collective on the data path); gloo only for the one-GPU plumbing test
        tower = model.get_model().image_tower
        tower.shard_frames(comm)
        # outside the timed region: the gathered features of THIS rank equal the unsharded encode of all T frames bit for bit, in
        # chronological order (rank order = frame order: `cur_image_idx` consumption, llava_arch.py:284-285) -- on every rank
        pix_all = torch.stack(frames)
        feats_sharded = tower(pix_all)
        feats_whole = eng.vit_features(pix_all).to(pix_all.dtype)
        shard_ok = bool(torch.equal(feats_sharded, feats_whole))
        flag = torch.tensor([1 if shard_ok else 0], dtype=torch.int32, device=device if args.dist_backend == "nccl" else "cpu")
        with deadline("all_reduce(MIN) of the shard-frames check"):
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        from teochat_amd.parallel import frame_partition
        shard_check = {"frames_per_rank": [c_ for _, c_ in frame_partition(T, world)],
                       "gathered_equals_unsharded_on_every_rank": bool(int(flag.item()) == 1)}
        if not shard_check["gathered_equals_unsharded_on_every_rank"]:
            raise SystemExit("frame-sharded tower: gathered features differ from the unsharded encode")
        # what the split buys, read off one line (outside the timed region; this rank's wall clock, median of 5 after a warm-up):
        # the whole tower on one GPU, the sharded call (local block + gather + reassembly), its local encode alone, the gather alone
        import statistics

        def _ms(fn, reps=5):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                with deadline("shard-frames timing barrier"):
                    dist.barrier()
                t_ = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t_) * 1e3)
            return statistics.median(ts)
        s_, c_ = frame_partition(T, world)[rank]
        cmax = max(c__ for _, c__ in frame_partition(T, world))
        NVt, Dvt = feats_whole.shape[1], feats_whole.shape[2]
        send_t = torch.zeros(cmax * NVt, Dvt, dtype=feats_whole.dtype, device=device)
        recv_t = torch.empty(world * cmax * NVt, Dvt, dtype=feats_whole.dtype, device=device)
        gather = (lambda: comm.all_gather_rows(send_t, recv_t)) if comm is not None else (lambda: dist.all_gather_into_tensor(recv_t, send_t))
        shard_times = {"tower_ms_unsharded": round(_ms(lambda: eng.vit_features(pix_all)), 3),
                       "tower_ms_sharded": round(_ms(lambda: tower(pix_all)), 3),
                       "local_encode_ms": round(_ms(lambda: eng.vit_features(pix_all[s_:s_ + max(c_, 1)])), 3),
                       "gather_ms": round(_ms(gather), 3), "gather_bytes_per_rank": int(send_t.numel() * send_t.element_size()),
                       "gather_via": "teo_allgather_visual (RCCL)" if comm is not None else f"torch.distributed {args.dist_backend}",
                       "note": "rank 0's clock; every timed call starts behind a barrier"}

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
            with deadline("barrier of the timed region"):
                dist.barrier()
            torch.cuda.synchronize()

    if os.environ.get("TEO_BENCH_STALL_RANK") == str(rank) and world > 1:
        # test hook (tests/test_bench_contract_gpu.py): this rank never reaches the barrier -> the others' deadline must end the job
        time.sleep(10 * max(args.collective_timeout, 1.0))
    for _ in range(args.warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    dt_own = dt
    assert out.shape[1] == n_text + n_out
    tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        with deadline("all_reduce(MAX) of the step time"):
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # every rank's own clock and setup time, gathered on the host group (N > 1): the line shows the spread and names the straggler
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "seconds": dt_own, "setup_s": setup_s}
        box = [None] * world
        with deadline("all_gather_object of the per-rank statistics"):
            dist.all_gather_object(box, mine, group=host_group)
        per_rank = sorted(box, key=lambda e: e["rank"])
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
    # prefill_ms above is ONE prefill right after a decode phase and a host sync (what a conversation's first token pays); the same call three more
    # times back to back, median: what the MFMA phase costs once clocks and translations are warm (tools/shape_sweep.py measures this way)
    warm = []
    for _ in range(3):
        eng.reset_cache(); torch.cuda.synchronize(); t = time.perf_counter()
        eng.prefill(emb[0], last_only=True); torch.cuda.synchronize()
        warm.append((time.perf_counter() - t) * 1e3)
    phases["prefill_ms_back_to_back_median3"] = sorted(warm)[1]
    if B > 1:
        # batched step at the headline context: fresh caches, the B prompts prefilled again, then 64 timed steps
        dec = model._batch_decoder
        dec.reset()
        seqs = []
        for fr_b, ids_b in batch_in:
            (_, _, _, _, emb_b, _) = model.prepare_inputs_labels_for_multimodal(ids_b, None, None, None, None, fr_b)
            seqs.append(emb_b[0])
        lgb = dec.prefill_all(seqs)
        dec.begin([int(lgb[b].argmax()) for b in range(B)]); torch.cuda.synchronize(); t = time.perf_counter()
        dec.steps(min(64, n_out - 1), use_graph=not args.no_graph); torch.cuda.synchronize()
        phases["batched_decode_ms_per_step"] = (time.perf_counter() - t) * 1e3 / min(64, n_out - 1)
        phases["batch"] = B
    # the product default: generate() looks at the tokens every chunk=16 steps (one .tolist() sync per chunk); the timed
    # region above uses chunk=n_out (one look per conversation).  One untimed-region measurement of the default path:
    if B == 1:
        model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=n_out, eos_token_id=None)   # warm (as the timed loop is)
        torch.cuda.synchronize(); t = time.perf_counter()
        model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=n_out, eos_token_id=None)
        torch.cuda.synchronize()
        phases["default_chunk16_tokens_per_s"] = n_out / (time.perf_counter() - t)
        # the reference's DEFAULT call samples (eval/inference.py:64-72: do_sample=True, temperature=0.2; HF's default top_k=50): the
        # device sampler (radix select over the 32 000 logits + multinomial inside decode_tail) replaces the argmax, nothing else
        model.generate(input_ids=ids, images=frames, do_sample=True, temperature=0.2, top_k=50, max_new_tokens=n_out, eos_token_id=None, chunk=n_out)
        torch.cuda.synchronize(); t = time.perf_counter()
        model.generate(input_ids=ids, images=frames, do_sample=True, temperature=0.2, top_k=50, max_new_tokens=n_out, eos_token_id=None, chunk=n_out)
        torch.cuda.synchronize()
        phases["sampled_tokens_per_s"] = n_out / (time.perf_counter() - t)
    phases = {k: round(v, 3) for k, v in phases.items()}

    # ---- roofline of the dominant kernel (decode gate/up GEMV: 43 % of the weight bytes of a token).
    # (1) IN-RUN: 8 real decode steps right after a real prefill (plain launches on the engine stream), every kernel launch timed
    #     by its own dispatch timestamps (teo_llama_decode_step_profile -> hipExtLaunchKernel start / stop events): kernel time
    #     only, per launch, in the real sequence of the step -- the quantity `rocprofv3 --kernel-trace --stats` reports for the same
    #     kernels (profiles/r04_bench_kernel_stats.md).  `achieved` / `frac` are computed from THIS number.
    # (2) chain microbenchmark: the same kernel over the 32 layers' matrices back to back between two HIP events (5.8 GB > L3);
    #     it has no neighbours of other kinds and is a few % faster; reported beside (1), never instead of it.
    cfg = model.config
    w_bytes = 1 if args.weights == "fp8" else 2
    Dh, Fi, Vv = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    QKVn = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
    ctx_prof = Lseq + 4
    kv_bytes = 2 * cfg.num_key_value_heads * cfg.head_dim * 2 * ctx_prof
    alg_bytes = {"qkv_rope_gemv": QKVn * Dh * w_bytes, "o_gemv": Dh * Dh * w_bytes, "gateup_gemv": 2 * Fi * Dh * w_bytes,
                 "down_gemv": Dh * Fi * w_bytes, "lm_head_gemv": Vv * Dh * w_bytes, "attn_decode_partial": kv_bytes}
    if B > 1:
        # the batched step (what this line's `value` is made of): B conversations prefilled again, then profiled batched steps
        dec = model._batch_decoder
        dec.reset()
        lgb = dec.prefill_all(seqs)
        dec.begin([int(lgb[b].argmax()) for b in range(B)])
        dec.steps_profiled(2)                                          # warm
        prof = dec.steps_profiled(4)
        for k_ in ("attn_decode_partial",):
            alg_bytes[k_] = kv_bytes * B                               # every conversation streams its own K / V
    else:
        eng.reset_cache()
        lg = eng.prefill(emb[0], last_only=True)
        eng.decode_begin(int(lg[0].argmax()))
        eng.decode_steps_profiled(2)                                   # warm
        prof = eng.decode_steps_profiled(8)
    in_run = {}
    for name, (per_step, us) in prof.items():
        e = {"launches_per_token": per_step, "avg_us": round(us, 2)}
        if name in alg_bytes:
            e["algorithmic_bytes"] = int(alg_bytes[name])
            e["GBps"] = round(alg_bytes[name] / (us * 1e-6) / 1e9, 1)
            e["frac_of_hbm_peak"] = round(e["GBps"] / HBM_PEAK_GBS, 4)
        if name == "attn_decode_combine":
            # measured with the kernel body removed (profiles/r04_decode_experiments.md, section 5): an EMPTY 32-workgroup kernel reads 4.2 us here
            e["note"] = "dispatch-timestamp floor: an empty kernel of this grid reads ~4.2 us; by step-time difference the body costs ~1.4 us per launch"
        in_run[name] = e
    pair_us = prof["attn_decode_partial"][1] + prof.get("attn_decode_combine", (0, 0.0))[1]
    in_run["attention_pair"] = {"avg_us": round(pair_us, 2), "algorithmic_bytes": int(alg_bytes["attn_decode_partial"]),
                                "frac_of_hbm_peak": round(alg_bytes["attn_decode_partial"] / (pair_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    step_kernel_us = sum(per_step * us for per_step, us in prof.values())
    gemv_bytes = 2 * Fi * Dh * w_bytes
    gu_us = prof["gateup_gemv"][1]
    achieved = gemv_bytes / (gu_us * 1e-6) / 1e9
    chain_ms = None
    if args.weights == "bf16":
        Ws = eng.llama_w["gateup"]
        arr, pp = L.ptr_array([w.data_ptr() for w in Ws])
        x = torch.randn(Dh, device=device).to(dtype)
        y = torch.empty(Fi, dtype=dtype, device=device)
        avg = C.c_float(0)
        # bench-only timing shim over the public C ABI (tools/libteo_bench.so, not part of the product library).  A secondary number
        # ("reported beside the in-run timing, never instead of it"): when the shim is missing or fails the line says why and goes on
        try:
            from tools import bench_shim
            shim = bench_shim.load()
            with eng.phase() as st:
                L.check(shim.teo_bench_gemv_chain(x.data_ptr(), pp, None, len(Ws), eng.llama_w["post_norm"][0].data_ptr(), y.data_ptr(),
                                                  2 * Fi, Dh, cfg.rms_norm_eps, L.GEMM_SWIGLU16, eng.dt, 5, C.byref(avg), st),
                        "teo_bench_gemv_chain")
            chain_ms = avg.value
        except Exception as e:  # noqa: BLE001
            chain_ms, chain_note = None, f"chain microbenchmark skipped: {e}"
            print(chain_note, file=sys.stderr)
    # HBM traffic comes from PMC counters, which need their own rocprofv3 --pmc passes (MI355X_MICROARCH.md section HBM:
    # FETCH_SIZE x2 on gfx950 + WRITE_SIZE); it is NOT measured inside this run: the numbers below are read from the committed
    # summary of those passes (tools/pmc_traffic.sh) and labelled with the file and the commit they were taken at.
    traffic, traffic_src, traffic_all = None, None, None
    try:
        from tools.src_hash import csrc_sha16
        src_now = csrc_sha16(ROOT)
    except Exception:  # noqa: BLE001 -- no sources beside the library: nothing can be tied to the kernels that ran
        src_now = None
    traffic_note = None

    def pmc_summary(names):
        """The newest committed PMC summary whose source fingerprint equals that of the kernel sources beside the running library (the
        summaries are separate rocprofv3 --pmc runs; a kernel change after they were taken must not keep quoting them)."""
        for name in names:
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            try:
                blob = json.load(open(path))
            except Exception:  # noqa: BLE001
                continue
            if src_now is None or blob.get("csrc_sha16") != src_now:
                return None, name, (f"profiles/{name} was taken on kernel sources {blob.get('csrc_sha16', '(no fingerprint: before round 5)')}, "
                                    f"the running library's sources are {src_now}: traffic not quoted until tools/pmc_*_traffic.sh is re-run")
            return blob, name, None
        return None, None, "no PMC summary under profiles/"
    blob, name, traffic_note = pmc_summary(("r06_pmc_decode_traffic.json", "r05_pmc_decode_traffic.json", "r04_pmc_decode_traffic.json"))
    if blob is not None:
        traffic = blob.get("hbm_bytes_per_launch")
        traffic_all = blob.get("kernels")
        traffic_src = {"from_profiles": "profiles/" + name, "commit": blob.get("commit", "see git log of the file"), "csrc_sha16": blob.get("csrc_sha16"),
                       "note": "separate rocprofv3 --pmc passes over the same kernels and shapes (same kernel sources: fingerprints equal); not measured in this run"}
    # Two live measurements of the same kernel: (1) its dispatch timestamps inside real decode steps, (2) HIP events around 32 x 5
    # back-to-back launches.  (2) is what `rocprofv3 --kernel-trace --stats` of this command agrees with (its per-kernel averages carry
    # the inter-kernel gap of the replayed graph: they add up to slightly MORE than the step); `achieved` / `frac` use the LONGER of
    # the two, so the headline fraction never exceeds what the committed rocprofv3 summary supports; both are reported.
    # The DOMINANT kernel of the step this line measures = the class with the largest share of the profiled step (launches x time):
    # the gate/up GEMV for one conversation (bf16 and fp8), the batched attention kernel for --batch B -- the line describes the run
    # it made.  (The attention class counts its combine launch when the split form ran.)
    share = {k_: v_[0] * v_[1] for k_, v_ in prof.items()}
    share["attn_decode_partial"] = share.get("attn_decode_partial", 0.0) + share.pop("attn_decode_combine", 0.0)
    dom = max(share, key=share.get)
    w_tag = "fp8-e4m3 weights" if args.weights == "fp8" else f"{args.dtype} weights"
    if B > 1:
        # which attention form ran: the whole-context kernel has no combine launch in the profiled step (attention.hip takes it when
        # conversations x heads fill the CUs), the split form carries one
        whole_ran = "attn_decode_combine" not in prof
        kern_names = {"attn_decode_partial": (f"attn_decode_whole_kernel<{args.dtype}, 16 lanes per row, 64-key chunks, RoPE + KV append>" if whole_ran
                                              else f"attn_decode_partial_kernel<{args.dtype}, 128-key chunks, RoPE + KV append> + attn_decode_combine_kernel")
                                             + f" (batched decode attention: {B} conversations x 32 heads x {ctx_prof} keys)",
                      "gateup_gemv": f"skinny_stream_kernel<{w_tag}, SWIGLU8> (batched decode gate/up + SwiGLU, {B} rows, N=22016 K=4096)",
                      "qkv_rope_gemv": f"skinny_stream_kernel<{w_tag}> (batched decode qkv, {B} rows, N=12288 K=4096)",
                      "down_gemv": f"skinny_gemm_kernel<{w_tag}> (batched decode down + residual, {B} rows, N=4096 K=11008)",
                      "o_gemv": f"skinny_gemm_kernel<{w_tag}> (batched decode o + residual, {B} rows)",
                      "lm_head_gemv": f"skinny_stream_kernel<{w_tag}> (batched lm_head, {B} rows, N=32000)"}
    else:
        kern_names = {"gateup_gemv": (f"gemv_kernel<{args.dtype},{args.dtype},R=2,U=4,NT,SWIGLU> (decode rmsnorm + gate/up + SwiGLU, N=22016 K=4096)" if args.weights == "bf16"
                                      else "gemv_kernel<fp8 weights, SWIGLU> (decode rmsnorm + gate/up + SwiGLU)"),
                      "attn_decode_partial": "attn_decode_partial_kernel + attn_decode_combine_kernel (decode attention pair)",
                      "qkv_rope_gemv": f"gemv_qkv_rope_kernel<{w_tag}> (decode rmsnorm + qkv + RoPE + KV append)",
                      "down_gemv": f"gemv_splitk_kernel<{w_tag}> (decode down + residual)", "o_gemv": f"gemv_splitk_kernel<{w_tag}> (decode o + residual)",
                      "lm_head_gemv": f"gemv_kernel<{w_tag}, f32 out> (rmsnorm + lm_head)"}
    # the committed PMC passes were taken on the bf16 C3 run: GEMV traffic applies to any bf16 run (it does not depend on the context),
    # attention traffic only at that context; fp8 / fp16 runs stream other bytes through other instantiations -> no traffic figure
    pmc_applies = B == 1 and args.weights == "bf16" and args.dtype == "bf16"
    if not pmc_applies:
        traffic, traffic_all_b1 = None, None
    else:
        traffic_all_b1 = traffic_all
    if dom == "gateup_gemv" and B == 1:
        avg_ms = gu_us * 1e-3 if chain_ms is None else max(gu_us * 1e-3, chain_ms)
        dom_bytes, dom_in_run_ms = gemv_bytes, gu_us * 1e-3
    else:
        dom_us = prof[dom][1] + (prof.get("attn_decode_combine", (0, 0.0))[1] if dom == "attn_decode_partial" else 0.0)
        avg_ms = dom_in_run_ms = dom_us * 1e-3
        dom_bytes = int(alg_bytes[dom])
        traffic = (traffic_all_b1 or {}).get(dom, {}).get("hbm_bytes_per_launch") if isinstance(traffic_all_b1, dict) else None
        if dom.startswith("attn") and abs(ctx_prof - 2176) > 64:
            traffic = None
        if B > 1:
            # the batched step's own PMC passes exist for config C5's per-GPU shape (8 conversations, fp8 weights: tools/pmc_batch_traffic.sh)
            traffic_all, traffic_src = None, None
            if B == 8 and args.weights == "fp8" and abs(ctx_prof - 2178) <= 64:       # same shape as the PMC passes
                blob, name, traffic_note = pmc_summary(("r06_pmc_batch_traffic.json", "r05_pmc_batch_traffic.json", "r04_pmc_batch_traffic.json"))
                if blob is not None:
                    key = {"attn_decode_partial": "attn_decode_whole", "gateup_gemv": "gateup_stream_fp8", "qkv_rope_gemv": "qkv_lmhead_stream_fp8",
                           "o_gemv": "o_tile_fp8", "down_gemv": "down_tile_fp8"}.get(dom)
                    traffic_all = blob.get("kernels")
                    traffic = (traffic_all or {}).get(key, {}).get("hbm_bytes_per_launch")
                    traffic_src = {"from_profiles": "profiles/" + name, "commit": blob.get("commit"), "csrc_sha16": blob.get("csrc_sha16"),
                                   "note": "separate rocprofv3 --pmc passes over `bench.py --batch 8 --weights fp8` (ctx 2178), same kernel sources; not measured in this run"}
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": kern_names.get(dom, dom),
                "share_of_profiled_step": round(share[dom] / sum(share.values()), 4),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src if traffic is not None else ({"note": traffic_note} if traffic_note else None),
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(avg_ms, 5),
                "measured": ("the longer of: in-run per-launch dispatch timestamps over 8 decode steps x 32 layers after a real prefill (ctx %d); "
                             "HIP events around 32 matrices x 5 back-to-back launches" % ctx_prof) if (dom == "gateup_gemv" and B == 1) else
                            ("in-run per-launch dispatch timestamps over %d profiled %s x 32 layers after a real prefill (ctx %d)"
                             % (4 if B > 1 else 8, "batched steps" if B > 1 else "decode steps", ctx_prof)),
                "in_run_avg_launch_ms": round(dom_in_run_ms, 5), "in_run_frac": round(dom_bytes / (dom_in_run_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "chain_microbench_avg_launch_ms": None if chain_ms is None else round(chain_ms, 5),
                "chain_microbench_frac": None if chain_ms is None else round(gemv_bytes / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "decode_kernels_in_run": in_run,
                "decode_step_sum_of_kernels_ms": round(step_kernel_us * 1e-3, 4)}
    if traffic_all and (B > 1 or pmc_applies):
        roofline["traffic_per_kernel"] = traffic_all
    # whole decode step against the HBM roofline (weights + KV per token)
    kv_ctx = Lseq + n_out / 2.0
    # weights a decode step actually streams: the layers' four matrices + lm_head (+ the norm vectors, 16-bit).  NOT the embedding table:
    # a step reads one row of it.  (SURVEY's 6.738e9 counts embed_tokens; rounds 1-4 used it and overstated the step by 1.8 %.)
    w_params = cfg.num_hidden_layers * (QKVn * Dh + Dh * cfg.num_attention_heads * cfg.head_dim + 2 * Fi * Dh + Dh * Fi) + Vv * Dh
    norm_bytes = (2 * cfg.num_hidden_layers + 1) * Dh * 2
    w_tok = w_params * (1 if args.weights == "fp8" else 2) + norm_bytes
    tok_bytes = w_tok + 2 * cfg.num_hidden_layers * cfg.num_key_value_heads * cfg.head_dim * 2 * kv_ctx
    # the MFMA-bound phases against the dense bf16 peak (algorithmic FLOPs of SURVEY.md section 8d)
    MFMA_PEAK_TFLOPS = 2500.0
    prefill_tf = (2.0 * Lseq * 6.476e9 + 2.0 * 4096 * 32000 + float(Lseq) ** 2 * 262144.0) / 1e12
    vit_tf = T * (155.3e9 + 10.74e9) / 1e12
    roofline["prefill_tflops"] = round(prefill_tf / (phases["prefill_ms"] * 1e-3), 1)
    roofline["prefill_frac_of_mfma_peak"] = round(roofline["prefill_tflops"] / MFMA_PEAK_TFLOPS, 4)
    roofline["vit_projector_tflops"] = round(vit_tf / (phases["encode_plus_splice_ms"] * 1e-3), 1)
    roofline["decode_step_frac_of_hbm_peak"] = round(tok_bytes / (phases["decode_ms_per_token"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    # t_min = FLOPs_dense / peak_mfma + Bytes_decode / peak_hbm over the whole job; end_to_end_frac = t_min / t_measured.
    # A batched step reads the weights ONCE and every conversation's K / V: bytes per step = weights + B x KV
    kv_tok = tok_bytes - w_tok
    roofline["decode_bytes_per_token"] = {"weights_streamed": int(w_tok), "kv_read_plus_append": int(kv_tok),
                                          "note": "layers + lm_head + norm vectors; embed_tokens is not streamed (one row per token)"}
    step_bytes = w_tok + B * kv_tok
    if B > 1:
        roofline["batched_step_frac_of_hbm_peak"] = round(step_bytes / (phases["batched_decode_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    t_min_s = B * (vit_tf + prefill_tf) / MFMA_PEAK_TFLOPS + (n_out - 1) * step_bytes / (HBM_PEAK_GBS * 1e9)
    t_meas_s = dt / args.steps                       # one step = B conversations prepared and decoded together
    roofline["t_min_ms"] = round(t_min_s * 1e3, 3)
    roofline["end_to_end_frac"] = round(t_min_s / t_meas_s, 4)

    result = {
        "metric": "end-to-end tokens/sec (prefill+decode), T=8 frames, LLaMA-2-7B",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype if args.weights == "bf16" else ("bf16 activations / fp8-e4m3 weights (decode: fp8 weight stream; prefill: "
                                                       + ("w8a8 on the fp8 MFMA)" if prefill_fp8 else "bf16 MFMA on the dequantised weights)")), "data": "synthetic",
        "config": {"workload": f"{workload_label(T, n_text, n_out, B, args.shard_frames, args.weights)}: T={T} frames 224x224 -> CLIP-ViT-L/14 (23 layers) -> mlp2x_gelu -> splice of a "
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
    result["setup_s"] = round(setup_s, 2)
    if per_rank is not None:
        own_tokens = args.steps * B * n_out          # what ONE rank generated in its own `seconds` (the sharded-tower run: the same conversation on every rank)
        vals = [own_tokens / e["seconds"] for e in per_rank]
        slow = max(range(world), key=lambda r_: per_rank[r_]["seconds"])
        result["per_rank"] = {"tokens_per_s": [round(v, 2) for v in vals], "ms_per_step": [round(e["seconds"] / args.steps * 1e3, 2) for e in per_rank],
                              "setup_s": [round(e["setup_s"], 2) for e in per_rank],
                              "min_tokens_per_s": round(min(vals), 2), "max_tokens_per_s": round(max(vals), 2), "straggler_rank": slow,
                              "spread": round(max(vals) / min(vals) - 1.0, 4),
                              "note": "each rank's own clock between the two barriers of the timed region; `value` uses the slowest rank's"}
    if args.shard_frames:
        result["shard_frames_check"] = shard_check
        result["shard_frames"] = shard_times
    result["rccl_ranks"] = rccl_info["rccl_ranks"] if rccl_info else (1 if world == 1 else None)
    if rccl_info:
        result["rccl"] = rccl_info
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(T, n_text, n_out)
    elif rank == 0 and world > 1:
        # measured at N = 1 only (the contract: rank 0, N = 1); an N > 1 line points at the committed N = 1 line instead of carrying null
        ref_line = None
        for name in ("r06_bench.json", "r05_bench.json", "r04_bench.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                try:
                    cb = json.load(open(path)).get("cpu_baseline")
                    if cb:
                        ref_line = dict(cb, measured_at="n_gpus = 1", from_profiles="profiles/" + name,
                                        note="not re-measured in this N > 1 run: the oracle is timed on rank 0 at N = 1 only")
                        break
                except Exception:  # noqa: BLE001
                    pass
        result["cpu_baseline"] = ref_line
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        stuck = th is not None and th.is_alive()         # the communicator check never came back: do not wait for it at exit either
        if stuck:
            sys.stdout.flush()
            os._exit(0)
        with deadline("destroy_process_group"):
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
