#!/bin/bash
# usage (on the GPU box): bash tools/refresh_profiles.sh   -> gpurun_out/r06p/{bench*.json, kernel_stats.md}
# The bench lines and the rocprofv3 kernel summary that profiles/r06_* are copied from.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06p
mkdir -p $OUT
cd $ROOT
timeout 900 python3 bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> $OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; }
run frames16 --frames 16
run frames2new128 --frames 2 --new 128
run frames2new16 --frames 2 --new 16          # the regime the reference's tasks live in: a few answer tokens, TTFT is the latency (eval/classification.py:15-41)
run weightsfp8 --weights fp8
run batch8 --batch 8
run batch8weightsfp8 --batch 8 --weights fp8
run batch16 --batch 16
run batch16weightsfp8 --batch 16 --weights fp8
run fp16 --dtype fp16
run frames16weightsfp8 --frames 16 --weights fp8
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
DB=$(find /tmp/prof_bench -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 $ROOT/tools/prof_summary.py $DB $OUT/bench_kernel_stats.md > /dev/null; fi
for f in $OUT/bench*.json; do echo "$(basename $f): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['unit'], d.get('phases',{}).get('prefill_ms'), d.get('phases',{}).get('decode_ms_per_token'), d['roofline']['frac'])" 2>&1 | tail -1)"; done
head -16 $OUT/bench_kernel_stats.md
