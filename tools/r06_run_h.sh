#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06h
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests/test_fp16_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "fp16 or attn or attention or decode" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2> $OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; }
run bf16a
run fp16a --dtype fp16
run bf16b
run fp16b --dtype fp16
run bf16c
run fp16c --dtype fp16
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); k=d['roofline']['decode_kernels_in_run']; print(d['value'], d.get('phases',{}).get('vit_ms'), d.get('phases',{}).get('prefill_ms'), d.get('phases',{}).get('decode_ms_per_token'), {a: k[a]['avg_us'] for a in ('attn_decode_partial','attn_decode_combine','gateup_gemv','lm_head_gemv')})" 2>&1 | tail -1)"; done
