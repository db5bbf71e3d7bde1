#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 1800 python -m pytest tests -q -x -m gpu > $O/pytest_72.txt 2>&1; grep "passed\|failed" $O/pytest_72.txt | tail -2
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b72_$tag.json; python -c "
import json
d=json.load(open('$O/b72_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), {n:v['avg_us'] for n,v in k.items() if n in ('qkv_rope_gemv','o_gemv','down_gemv')})"; }
run base1
run base2
run b8fp8 --batch 8 --weights fp8
run b8fp8_2 --batch 8 --weights fp8
