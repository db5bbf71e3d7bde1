#!/bin/bash
# round 6, first GPU pass: the new tests, the T x prompt shape sweep, fp16 / bf16 kernel summaries for the per-kernel A/B, the C5 baseline line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06a
mkdir -p $OUT
cd $ROOT
timeout 1200 python3 -m pytest tests/test_batch_gpu.py tests/test_model_gpu.py tests/test_fp16_gpu.py -x -q -m gpu -s -k "continuation or tinyC or bf16_matches or fp16_tiny or forward_api or batched_decode_fp32" > $OUT/pytest_new.txt 2>&1
tail -5 $OUT/pytest_new.txt
grep -h "HIP vs oracle" $OUT/pytest_new.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -4 $OUT/smoke.txt
timeout 1500 python3 tools/shape_sweep.py --out $OUT/shape_sweep.json > $OUT/shape_sweep.log 2>&1; tail -40 $OUT/shape_sweep.log
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; }
run bf16
run fp16 --dtype fp16
run batch8weightsfp8 --batch 8 --weights fp8
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/prof_$name.log 2>&1
  DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 $ROOT/tools/prof_summary.py $DB $OUT/bench_${name}_kernel_stats.md > /dev/null; fi; }
prof bf16
prof fp16 --dtype fp16
cd $ROOT
for f in $OUT/bench_*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['unit'], d.get('phases',{}).get('prefill_ms'), d.get('phases',{}).get('decode_ms_per_token'), d.get('phases',{}).get('batched_decode_ms_per_step'), d['roofline']['frac'])" 2>&1 | tail -1)"; done
