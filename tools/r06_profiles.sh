#!/bin/bash
# usage (on the GPU box): bash tools/r06_profiles.sh <commit>   -> gpurun_out/r06p/*  (what profiles/r06_* are copied from)
# The round's final measurement pass.  The PMC traffic passes run FIRST: their summaries carry the fingerprint of the kernel sources they
# were taken on (tools/src_hash.py) and are copied into profiles/ on the box, so that the bench lines made afterwards quote a `roofline.traffic`
# whose fingerprint equals the running library's.  Then every bench line, the rocprofv3 kernel summaries of the same commands, the PMC
# passes (MFMA busy / wait buckets / L2), the shape sweep, the fp16 data probe and the driver's own command.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
COMMIT=${1:-unknown}
cd $ROOT
mkdir -p gpurun_out/r06p
bash tools/pmc_decode_traffic.sh $COMMIT > gpurun_out/r06p/pmc_decode_traffic.log 2>&1
bash tools/pmc_batch_traffic.sh $COMMIT > gpurun_out/r06p/pmc_batch_traffic.log 2>&1
cp gpurun_out/r06p/r06_pmc_decode_traffic.json gpurun_out/r06p/r06_pmc_batch_traffic.json profiles/ 2>/dev/null
bash tools/refresh_profiles.sh
bash tools/profile_variants.sh
bash tools/pmc_bench.sh
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06p/bench_driver_style.err | tail -1 > gpurun_out/r06p/bench_driver_style.json
timeout 1500 python3 tools/shape_sweep.py --out gpurun_out/r06p/shape_sweep.json > gpurun_out/r06p/shape_sweep.log 2>&1
timeout 600 python3 tools/fp16_probe.py > gpurun_out/r06p/fp16_probe.txt 2>&1
timeout 300 python3 tools/vit_probe.py 8 50 > gpurun_out/r06p/vit_probe.txt 2>&1; timeout 300 python3 tools/vit_probe.py 2 50 >> gpurun_out/r06p/vit_probe.txt 2>&1
# property walks: a GEMM with more rows cannot be faster (bf16 and w8a8 prefill GEMMs), a batched step with more conversations cannot be faster
timeout 600 python3 tools/dispatch_monotone.py --out gpurun_out/r06p/dispatch_monotone.txt > /dev/null 2>&1
timeout 600 python3 tools/dispatch_monotone.py --fp8 --out gpurun_out/r06p/dispatch_monotone_fp8.txt > /dev/null 2>&1
timeout 900 python3 tools/batch_sweep.py --weights bf16 --batches 1,2,3,4,6,8,9,10,11,12,13,14,15,16 --out gpurun_out/r06p/batch_sweep_bf16.json > gpurun_out/r06p/batch_sweep_bf16.log 2>&1
timeout 900 python3 tools/batch_sweep.py --weights fp8 --batches 1,2,3,4,6,8,9,10,11,12,13,14,15,16 --out gpurun_out/r06p/batch_sweep_fp8.json > gpurun_out/r06p/batch_sweep_fp8.log 2>&1
