#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
bash tools/pmc_batch_traffic.sh ${1:-unknown} 2>&1 | tail -c 3000
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fp16
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_fp16 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp16 > $GRAFT_REPO_ROOT/$O/prof_fp16.log 2>&1
DB=$(find /tmp/prof_fp16 -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $DB $GRAFT_REPO_ROOT/$O/bench_fp16_kernel_stats.md > /dev/null; head -24 $GRAFT_REPO_ROOT/$O/bench_fp16_kernel_stats.md | cut -c1-160; fi
