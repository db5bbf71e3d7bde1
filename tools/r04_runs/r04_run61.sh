#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_batch_gpu.py tests/test_fp16_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -q -x > $O/pytest_61.txt 2>&1; grep "passed\|failed" $O/pytest_61.txt | tail -2
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b61_$tag.json; python -c "
import json
d=json.load(open('$O/b61_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), p.get('sampled_tokens_per_s'), {n:v['avg_us'] for n,v in k.items()})"; }
run base1
run base2
run b4fp8 --batch 4 --weights fp8
timeout 300 python tools/attn_probe.py 1 2300 2>&1 | grep -v amdgpu | head -4
timeout 300 python tools/attn_probe.py 4 2300 2>&1 | grep -v amdgpu | head -4
