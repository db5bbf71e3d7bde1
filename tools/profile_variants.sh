#!/bin/bash
# usage (on the GPU box): bash tools/profile_variants.sh  -> gpurun_out/r06p/bench_<variant>_kernel_stats.md
# rocprofv3 kernel summaries of the other north-star shapes (the default C3 one is made by tools/refresh_profiles.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/prof_$name.log 2>&1
  DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 $ROOT/tools/prof_summary.py $DB $OUT/bench_${name}_kernel_stats.md > /dev/null; head -14 $OUT/bench_${name}_kernel_stats.md; fi; }
prof frames16 --frames 16
prof weightsfp8 --weights fp8
prof batch8weightsfp8 --batch 8 --weights fp8
prof batch8 --batch 8
prof fp16 --dtype fp16
prof frames2new128 --frames 2 --new 128
