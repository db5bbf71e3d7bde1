#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c
mkdir -p $OUT
cd $ROOT
timeout 300 python3 tools/cvt_probe.py > $OUT/cvt_probe.txt 2>&1; cat $OUT/cvt_probe.txt
