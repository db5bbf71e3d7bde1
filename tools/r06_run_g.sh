#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06g
mkdir -p $OUT
cd $ROOT
timeout 1200 python3 -m pytest tests/test_gemm_fuzz_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or tile or hybrid or stream_k or fuzz" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 900 python3 tools/dispatch_probe.py 638 > $OUT/dispatch_probe.txt 2>&1; cut -c1-700 $OUT/dispatch_probe.txt
timeout 1500 python3 tools/shape_sweep.py --out $OUT/shape_sweep.json > $OUT/shape_sweep.log 2>&1; tail -30 $OUT/shape_sweep.log
timeout 600 python3 bench.py --no-cpu-baseline 2> $OUT/bench.err | tail -1 > $OUT/bench.json; python3 -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['phases'])"
