#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_batch_gpu.py tests/test_fp16_gpu.py tests/test_model_gpu.py -q -x > $O/pytest_66.txt 2>&1; grep "passed\|failed" $O/pytest_66.txt | tail -2
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b66_$tag.json; python -c "
import json
d=json.load(open('$O/b66_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), p['prefill_ms'], p['vit_ms'], {n:v['avg_us'] for n,v in k.items()})"; }
run base1
run base2
run b8fp8 --batch 8 --weights fp8
