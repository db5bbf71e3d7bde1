#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "weight_ring or skinny" 2>&1 | tail -5
for w in 1 0; do echo "== skinny_wdma=$w"; SK_WDMA=$w SK_SHAPES=decode SK_TILES=1 SK_NORM=0 timeout 300 python tools/bench_kernels.py skinny 2>&1 | grep "down\| o " ; done
for w in 1 0 1 0; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch 8 --weights fp8 --tune skinny_wdma=$w 2>/dev/null | tail -1 > $O/b50_$w.json; python -c "
import json
d=json.load(open('$O/b50_$w.json')); k=d['roofline']['decode_kernels_in_run']; print('B8 fp8 wdma=$w', d['value'], d['phases']['batched_decode_ms_per_step'], k['down_gemv']['avg_us'], k['o_gemv']['avg_us'])"; done
for w in 1 0; do
timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --batch 16 --tune skinny_wdma=$w 2>/dev/null | tail -1 > $O/b50_b16_$w.json; python -c "
import json
d=json.load(open('$O/b50_b16_$w.json')); k=d['roofline']['decode_kernels_in_run']; print('B16 bf16 wdma=$w', d['value'], d['phases']['batched_decode_ms_per_step'], k['down_gemv']['avg_us'])"; done
