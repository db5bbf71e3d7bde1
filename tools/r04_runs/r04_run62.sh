#!/bin/bash
# A/B of -mllvm -amdgpu-kernarg-preload-count=16 (kernel arguments preloaded into SGPRs at wave launch) on one box
O=gpurun_out/r04; mkdir -p $O
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b62_$tag.json; python -c "
import json
d=json.load(open('$O/b62_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), p['prefill_ms'], p['vit_ms'], {n:v['avg_us'] for n,v in k.items()})"; }
for rep in 1 2; do
cp teochat_amd/libteo_hip_nopl.so teochat_amd/libteo_hip.so
run nopl_$rep
[ $rep = 1 ] && run nopl_b8fp8 --batch 8 --weights fp8
cp teochat_amd/libteo_hip_pl.so teochat_amd/libteo_hip.so
run pl_$rep
[ $rep = 1 ] && run pl_b8fp8 --batch 8 --weights fp8
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x > $O/pytest_62.txt 2>&1; grep "passed\|failed" $O/pytest_62.txt | tail -2
