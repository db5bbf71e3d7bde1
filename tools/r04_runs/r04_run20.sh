#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
for ring in 0 1; do for stream in 1 0; do
echo "== SK_RING=$ring SK_STREAM=$stream SK_NORM=0"
SK_SHAPES=decode SK_TILES=1 SK_NORM=0 SK_RING=$ring SK_STREAM=$stream timeout 300 python tools/bench_kernels.py skinny 2>&1 | grep -v "amdgpu\|prefill GEMM"
done; done > $O/skinny_ring_sweep.txt
cat $O/skinny_ring_sweep.txt
for v in "skinny_ring=0" "skinny_ring=1" "skinny_stream=0"; do
timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --batch 16 --tune $v 2> /dev/null | tail -1 > $O/b20_b16_${v/=/_}.json
python -c "
import json
d=json.load(open('$O/b20_b16_${v/=/_}.json')); print('B16 bf16 $v', d['value'], d['phases']['batched_decode_ms_per_step'])"
done
