#!/bin/bash
# usage (GPU box): bash tools/profile_tower_pipe.sh  -> gpurun_out/r06t/vit{T}_pipe{0,1}_kernel_stats.md
# rocprofv3 --kernel-trace --stats of the tower alone (tools/vit_probe.py) at T = 2 and T = 8 with the software-pipelined small tiles off / on
# (engine knob gemm_narrow_pipe): the per-kernel evidence behind profiles/r06_pipe_ab.txt.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in 2 8; do for p in 0 1; do
  rm -rf /tmp/prof_vit
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_vit -o vit -- python3 $ROOT/tools/vit_probe.py $T 50 gemm_narrow_pipe=$p > $OUT/vit${T}_pipe${p}.log 2>&1
  DB=$(find /tmp/prof_vit -name "*.db" | head -1)
  if [ -n "$DB" ]; then python3 $ROOT/tools/prof_summary.py $DB $OUT/vit${T}_pipe${p}_kernel_stats.md > /dev/null; fi
  grep tower $OUT/vit${T}_pipe${p}.log
done; done
