#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 300 python tools/vit_gemm_probe.py > $O/vit_gemm_probe.txt 2>&1
for v in "ring0_unr0:" "ring1:skinny_ring=1" "unr4:skinny_unr=4" "stream0:skinny_stream=0"; do
  name=${v%%:*}; kv=${v#*:}
  timeout 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch 8 --weights fp8 ${kv:+--tune $kv} 2> $O/b6_$name.err | tail -1 > $O/b6_$name.json
  python -c "
import json
d=json.load(open('$O/b6_$name.json')); print('b8fp8 $name', d['value'], d['phases'].get('batched_decode_ms_per_step'))"
done
SK_SHAPES=decode SK_TILES=1 SK_NORM=0 timeout 400 python tools/bench_kernels.py skinny > $O/skinny_v3.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "skinny" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_true_shapes_gpu.py tests/test_bf16_walk_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -150 > $O/t6_true_shapes.txt; tail -3 $O/t6_true_shapes.txt
grep -v amdgpu $O/vit_gemm_probe.txt; grep -v "amdgpu\|prefill" $O/skinny_v3.txt
