#!/bin/bash
# round 6, second GPU pass: fp16 data-dependence probe, dispatch probe across M, batched-step experiments (16-wave tile kernel, grid multiplier)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p $OUT
cd $ROOT
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "skinny" > $OUT/pytest_skinny.txt 2>&1; tail -3 $OUT/pytest_skinny.txt
timeout 600 python3 tools/fp16_probe.py > $OUT/fp16_probe.txt 2>&1; cat $OUT/fp16_probe.txt
timeout 1500 python3 tools/dispatch_probe.py 767 797 893 1022 1148 2304 2432 2552 2688 2816 3072 3188 3328 > $OUT/dispatch_probe.txt 2>&1; cat $OUT/dispatch_probe.txt
for cfg in "0 0" "16 0" "0 2" "0 3" "16 2"; do set -- $cfg
  echo "== skinny microbench SK_WAVES=$1 SK_GRID=$2" >> $OUT/skinny_micro.txt
  SK_WAVES=$1 SK_GRID=$2 SK_SHAPES=decode SK_TILES=1 SK_NORM=0 timeout 300 python3 tools/bench_kernels.py skinny 2>&1 | grep -v "prefill GEMM" >> $OUT/skinny_micro.txt
done
cat $OUT/skinny_micro.txt
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline --batch 8 --weights fp8 "$@" 2> $OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json
  python3 -c "import json; d=json.load(open('$OUT/bench_$name.json')); k=d['roofline']['decode_kernels_in_run']; print('$name', d['value'], d['phases']['batched_decode_ms_per_step'], {a: k[a]['avg_us'] for a in k if 'avg_us' in k[a]})"; }
run base
run w16 --tune skinny_waves=16
run g2 --tune skinny_grid=2
run g3 --tune skinny_grid=3
run w16g2 --tune skinny_waves=16 --tune skinny_grid=2
