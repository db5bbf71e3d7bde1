#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_batch_gpu.py tests/test_fp16_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_true_shapes_gpu.py -q -x > $O/pytest_71.txt 2>&1; grep "passed\|failed" $O/pytest_71.txt | tail -2
for d in 0 4 0 4; do timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --tune attn_combine_dbg=$d 2>/dev/null | tail -1 > $O/b71_$d.json; python -c "
import json
d=json.load(open('$O/b71_$d.json')); p=d['phases']; print('merge form $d (0 one-wave, 4 barriers):', d['value'], p['decode_ms_per_token'], p['sampled_tokens_per_s'])"; done
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch 8 --weights fp8 2>/dev/null | tail -1 > $O/b71_b8.json; python -c "
import json
d=json.load(open('$O/b71_b8.json')); p=d['phases']; print('b8 fp8', d['value'], p['batched_decode_ms_per_step'])"
