#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06d
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_fp16_gpu.py tests/test_gemm_fuzz_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 600 python3 tools/fp16_probe.py > $OUT/fp16_probe.txt 2>&1; tail -4 $OUT/fp16_probe.txt
timeout 1500 python3 tools/shape_sweep.py --out $OUT/shape_sweep.json > $OUT/shape_sweep.log 2>&1; tail -32 $OUT/shape_sweep.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; }
run bf16
run fp16 --dtype fp16
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['unit'], d.get('phases',{}).get('prefill_ms'), d.get('phases',{}).get('decode_ms_per_token'), d['roofline']['frac'])" 2>&1 | tail -1)"; done
