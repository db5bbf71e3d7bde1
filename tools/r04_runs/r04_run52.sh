#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemv" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_fp16_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -3
GV_SHAPES=splitk timeout 300 python tools/bench_kernels.py gemv_fp8 2>&1 | grep gemv
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --weights fp8 2>/dev/null | tail -1 > $O/b52_fp8_$i.json; python -c "
import json
d=json.load(open('$O/b52_fp8_$i.json')); k=d['roofline']['decode_kernels_in_run']; print('fp8 B1', d['value'], d['phases']['decode_ms_per_token'], k['o_gemv']['avg_us'], k['down_gemv']['avg_us'])"; done
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/b52.json; python -c "
import json
d=json.load(open('$O/b52.json')); k=d['roofline']['decode_kernels_in_run']; print('bf16 B1', d['value'], d['phases']['decode_ms_per_token'], k['o_gemv']['avg_us'], k['down_gemv']['avg_us'])"
