#!/bin/bash
# where the small-M phases go: kernel-time sums vs wall for the tower at T = 2 / 8 and the C2 bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python3 "$@" > $OUT/prof_$name.log 2>&1
  DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 $ROOT/tools/prof_summary.py $DB $OUT/${name}_kernel_stats.md > /dev/null; fi; tail -2 $OUT/prof_$name.log; }
prof vit2 $ROOT/tools/vit_probe.py 2 20
prof vit8 $ROOT/tools/vit_probe.py 8 20
prof c2 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames 2 --new 128
cd $ROOT
python3 tools/vit_probe.py 2 50; python3 tools/vit_probe.py 8 50
timeout 900 python3 tools/vit_gemm_probe.py > $OUT/vit_gemm_probe.txt 2>&1; cat $OUT/vit_gemm_probe.txt | cut -c1-600
