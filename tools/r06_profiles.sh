#!/bin/bash
# usage (on the GPU box): bash tools/r06_profiles.sh <commit>   -> gpurun_out/r06p/*  (what profiles/r06_* are copied from)
# The round's final measurement pass: every bench line, the rocprofv3 kernel summaries of the same commands, the PMC passes (MFMA busy /
# wait buckets / L2) and the PMC traffic summaries (fingerprinted with the kernel sources they were taken on: tools/src_hash.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
COMMIT=${1:-unknown}
cd $ROOT
bash tools/refresh_profiles.sh
bash tools/profile_variants.sh
bash tools/pmc_bench.sh
bash tools/pmc_decode_traffic.sh $COMMIT
bash tools/pmc_batch_traffic.sh $COMMIT
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06p/bench_driver_style.err | tail -1 > gpurun_out/r06p/bench_driver_style.json
