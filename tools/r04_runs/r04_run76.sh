#!/bin/bash
# K rows of the decode attention split kernel requested before the position arrives: tests, then A/B against the previous library on this box
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_batch_gpu.py tests/test_fp16_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -q -x > $O/pytest_76.txt 2>&1; grep "passed\|failed" $O/pytest_76.txt | tail -2
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b76_$tag.json; python -c "
import json
d=json.load(open('$O/b76_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), {n:v['avg_us'] for n,v in k.items() if 'attn' in n})"; }
for rep in 1 2; do
cp teochat_amd/libteo_hip_base.so teochat_amd/libteo_hip.so; run base_$rep
cp teochat_amd/libteo_hip_spec.so teochat_amd/libteo_hip.so; run spec_$rep
done
cp teochat_amd/libteo_hip_base.so teochat_amd/libteo_hip.so; run base_b4 --batch 4
cp teochat_amd/libteo_hip_spec.so teochat_amd/libteo_hip.so; run spec_b4 --batch 4
