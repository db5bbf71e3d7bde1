"""The bench.py contract the driver depends on: one JSON line on stdout with the agreed keys (a short run on the real
7B-shaped synthetic model: T=2 frames, 32-token prompt, 8 new tokens)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--frames", "2", "--prompt", "32", "--new", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "phases"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] < 1.0
    # round 3: the dominant kernel is timed IN the run (per-launch dispatch timestamps), the chain microbenchmark is beside it,
    # every decode kernel has its own line, and the end-to-end fraction t_min / t_measured of SURVEY 8d is there
    assert "in-run" in rf["measured"] and rf["chain_microbench_frac"] is not None and rf["in_run_frac"] is not None
    assert rf["frac"] <= min(rf["in_run_frac"], rf["chain_microbench_frac"]) + 1e-3          # the headline uses the longer duration
    kinds = rf["decode_kernels_in_run"]
    for k in ("qkv_rope_gemv", "attn_decode_partial", "o_gemv", "gateup_gemv", "down_gemv", "lm_head_gemv", "decode_tail", "attention_pair"):
        assert k in kinds and kinds[k]["avg_us"] > 0, k
    assert kinds["gateup_gemv"]["launches_per_token"] == 32 and kinds["lm_head_gemv"]["launches_per_token"] == 1
    assert abs(kinds["gateup_gemv"]["frac_of_hbm_peak"] - rf["in_run_frac"]) < 2e-3
    assert 0.0 < rf["end_to_end_frac"] < 1.0
    assert "gemv_kernel" in rf["kernel"] and rf["share_of_profiled_step"] > 0.2          # the gate/up GEMV dominates a single-conversation step
    assert d["phases"]["sampled_tokens_per_s"] > 0                                        # the reference's default call samples (temperature 0.2, top_k 50)
    assert d["rccl_ranks"] == 1


def test_bench_variant_line_names_the_kernel_that_dominated_that_run():
    """`--batch 8 --weights fp8` (BASELINE config 5's per-GPU work): the roofline object describes THIS run -- the dominant kernel is
    picked from the profiled batched step (a batched-decode kernel, not the single-conversation gate/up GEMV), its share of the
    step is stated, and end_to_end_frac / the batched step's HBM fraction exist for B > 1."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--frames", "2",
                        "--prompt", "32", "--new", "8", "--no-cpu-baseline", "--batch", "8", "--weights", "fp8"],
                       capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    rf = d["roofline"]
    assert "batched" in rf["kernel"] and "gemv_kernel" not in rf["kernel"], rf["kernel"]     # a kernel of the BATCHED step (attention or a skinny GEMM)
    assert 0.0 < rf["share_of_profiled_step"] <= 1.0 and 0.0 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["end_to_end_frac"] is not None and 0.0 < rf["end_to_end_frac"] < 1.0
    assert 0.0 < rf["batched_step_frac_of_hbm_peak"] < 1.0
    kinds = rf["decode_kernels_in_run"]
    assert kinds["gateup_gemv"]["launches_per_token"] == 32 and kinds["attn_decode_partial"]["launches_per_token"] == 32
    assert d["phases"]["batch"] == 8 and d["phases"]["batched_decode_ms_per_step"] > 0
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The N > 1 plumbing of bench.py (rank env, barrier, max-over-ranks time, whole-job value) with two ranks sharing
    cuda:0 over gloo -- the RCCL path itself needs a multi-GPU node."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "1", "--frames", "2", "--prompt", "32", "--new", "8", "--dist-backend", "gloo", "--same-gpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and (d["cpu_baseline"] is None or d["cpu_baseline"]["measured_at"] == "n_gpus = 1")
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]      # both ranks' tokens over the max time
    assert "dp2" in d["config"]["parallelism"]


def test_bench_gpus2_without_a_launcher_starts_two_ranks_itself():
    """VERDICT r02 missing #1: the driver calls `python bench.py --gpus N` with NO torchrun; bench.py must start the N ranks
    itself (a child `torch.distributed.run`, the parent never touches the GPU) and rank 0 must report the ranks that joined."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--frames", "2", "--prompt", "32", "--new", "8", "--dist-backend", "gloo", "--same-gpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "dp2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    assert "rccl_ranks" in d and d["rccl_ranks"] is None        # gloo plumbing run: no RCCL communicator was built


def _bench(flags, timeout=1500, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    return r, lines


def test_bench_eight_ranks_c4_partition_on_one_gpu():
    """Round 5 (VERDICT r04 "Next round" #5): config C4's REAL partition -- T = 16 frames, 2 frames per rank x 8 ranks, rank order =
    chronological order (llava_arch.py:284-285 consumes the features in order) -- with all eight ranks on this one GPU (8 bf16 replicas
    = 8 x 14.2 GB of 288 GB) over gloo; the reference is single-GPU (scripts/eval_teochat.sh:9-10), the split is this build's own.
    bench.py itself checks, on every rank, that the gathered features equal the unsharded tower's bit for bit.  No scaling number is
    claimed from this: eight ranks share one GPU."""
    r, lines = _bench(["--gpus", "8", "--same-gpu", "--dist-backend", "gloo", "--shard-frames", "--frames", "16", "--new", "8",
                       "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-4000:]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["frames"] == 16 and d["config"]["sequence_len"] == 128 - 16 + 16 * 256
    assert "frame-sharded" in d["config"]["parallelism"]
    assert d["shard_frames_check"] == {"frames_per_rank": [2] * 8, "gathered_equals_unsharded_on_every_rank": True}
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]          # ONE conversation: the ranks replicate the LLM
    # round 6: what the split buys is read off the line -- the whole tower, the sharded call, its local block and the gather, each timed
    sf = d["shard_frames"]
    for k in ("tower_ms_unsharded", "tower_ms_sharded", "local_encode_ms", "gather_ms"):
        assert sf[k] > 0, k
    assert sf["gather_bytes_per_rank"] == 2 * 256 * 1024 * 2 and "gloo" in sf["gather_via"]        # 2 frames x 256 tokens x 1024 x bf16 = 1 MiB per rank
    _check_per_rank(d, 8, 8)
    assert d["cpu_baseline"] is None or d["cpu_baseline"].get("measured_at") == "n_gpus = 1"


def test_bench_eight_data_parallel_ranks_on_one_gpu():
    """The driver's SCALE form at N = 8 (`python bench.py --gpus 8`, conversation-level data parallel, weak scaling) as far as one GPU
    can take it: eight ranks, eight different conversations, barrier + max-over-ranks time, whole-job value, and the N = 1 cpu_baseline
    carried by reference instead of null."""
    r, lines = _bench(["--gpus", "8", "--same-gpu", "--dist-backend", "gloo", "--batch", "1", "--frames", "2", "--prompt", "32", "--new", "8",
                       "--steps", "1", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-4000:]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "dp8" in d["config"]["parallelism"]
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]      # all ranks' tokens over the slowest rank's time
    cb = d["cpu_baseline"]
    assert cb is not None and cb["measured_at"] == "n_gpus = 1" and cb["value"] > 0 and cb["from_profiles"].startswith("profiles/")
    _check_per_rank(d, 8, 8)
    # first contact with 8 ranks must not be a setup cliff: the synthetic weights are seed-generated on each rank's GPU, so eight ranks
    # SHARING one GPU (and the job's CPUs) get ready within 2 x the time one rank needs alone (+ 20 s of launcher / rendezvous slack)
    r1, l1 = _bench(["--gpus", "1", "--frames", "2", "--prompt", "32", "--new", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(l1[0])["setup_s"]
    print(f"setup_s: one rank {one:.1f} s; eight ranks on one GPU {d['per_rank']['setup_s']}")
    assert max(d["per_rank"]["setup_s"]) <= 2.0 * one + 20.0, (one, d["per_rank"]["setup_s"])


def _check_per_rank(d, world, tokens_per_rank_step):
    """Round 6 (VERDICT r05 "Next round" #5b): an N > 1 line carries every rank's own clock, the spread and the straggler."""
    pr = d["per_rank"]
    assert len(pr["tokens_per_s"]) == len(pr["ms_per_step"]) == len(pr["setup_s"]) == world
    assert pr["min_tokens_per_s"] == min(pr["tokens_per_s"]) and pr["max_tokens_per_s"] == max(pr["tokens_per_s"])
    assert 0 <= pr["straggler_rank"] < world and pr["ms_per_step"][pr["straggler_rank"]] == max(pr["ms_per_step"])
    assert abs(max(pr["ms_per_step"]) - d["ms_per_step"]) < 0.02 * d["ms_per_step"]          # `value` is made of the slowest rank's time
    assert all(v > 0 for v in pr["setup_s"]) and d["setup_s"] > 0


def test_a_hung_collective_ends_the_job_with_a_nonzero_code():
    """Round 6 (VERDICT r05 "Next round" #5d): every rendezvous / barrier / reduction of an N > 1 run sits under a hard deadline.  Rank 1
    stalls before the timed region's barrier (test hook); rank 0 must not wait for ever: it leaves with code 3, the launcher ends the job
    with a non-zero code and no JSON line."""
    r, lines = _bench(["--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--frames", "2", "--prompt", "32", "--new", "8", "--steps", "1",
                       "--warmup", "0", "--no-cpu-baseline", "--collective-timeout", "45"], timeout=900, extra_env={"TEO_BENCH_STALL_RANK": "1"})
    assert r.returncode != 0
    assert not lines, r.stdout[-1000:]
    assert "did not finish within" in r.stderr


def test_a_failing_rank_fails_the_whole_bench():
    """rank != 0 dying must surface in the exit code of `python bench.py --gpus N` (the launcher ends the job); no JSON line is printed."""
    r, lines = _bench(["--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--frames", "2", "--prompt", "32", "--new", "8", "--steps", "1",
                       "--warmup", "0", "--no-cpu-baseline"], timeout=900, extra_env={"TEO_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not lines, r.stdout[-1000:]


def test_bench_refuses_more_ranks_than_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
