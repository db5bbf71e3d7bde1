#!/bin/bash
# usage: tools/pmc_run.sh <out-name> <kernel-name-filter> <bench_kernels.py args...>
# Three rocprofv3 --pmc passes (counters only: no trace domains) over a kernel microbenchmark; summaries under gpurun_out/r02/.
set -u
OUT=$1; FILT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
P3="TCC_HIT_sum TCC_MISS_sum"
i=0
: > $ROOT/gpurun_out/r02/$OUT.txt
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${OUT}_$i
  rocprofv3 --pmc $P -d /tmp/pmc_${OUT}_$i -o pmc -- python3 $ROOT/tools/bench_kernels.py "$@" > /tmp/pmc_${OUT}_$i.log 2>&1
  DB=$(find /tmp/pmc_${OUT}_$i -name "*.db" | head -1)
  echo "--- pass $i: $P" >> $ROOT/gpurun_out/r02/$OUT.txt
  grep -E "TFLOP|GB/s" /tmp/pmc_${OUT}_$i.log >> $ROOT/gpurun_out/r02/$OUT.txt
  if [ -n "$DB" ]; then python3 $ROOT/tools/pmc_summary.py $DB "$FILT" >> $ROOT/gpurun_out/r02/$OUT.txt; else echo "no db; log tail:" >> $ROOT/gpurun_out/r02/$OUT.txt; tail -5 /tmp/pmc_${OUT}_$i.log >> $ROOT/gpurun_out/r02/$OUT.txt; fi
done
cat $ROOT/gpurun_out/r02/$OUT.txt
