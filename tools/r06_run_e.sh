#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06e
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_fp16_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 600 python3 tools/fp16_probe.py > $OUT/fp16_probe.txt 2>&1; tail -8 $OUT/fp16_probe.txt
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; }
run bf16
run fp16 --dtype fp16
run bf16b
run fp16b --dtype fp16
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); k=d['roofline']['decode_kernels_in_run']; print(d['value'], d['unit'], d.get('phases',{}).get('prefill_ms'), d.get('phases',{}).get('decode_ms_per_token'), d['roofline']['frac'], {a: k[a]['avg_us'] for a in k if 'avg_us' in k[a]})" 2>&1 | tail -1)"; done
