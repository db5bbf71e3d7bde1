#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 300 python tools/wave_probe.py 2>&1 | grep -v amdgpu | tee $O/wave_probe.txt
timeout 300 python tools/sampler_time.py 2>&1 | grep -v amdgpu | tee $O/sampler_time_54.txt
timeout 600 python tools/sampler_fuzz.py 2>&1 | grep -v amdgpu | tail -2 | tee $O/sampler_fuzz_54.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/b54_$i.json; python -c "
import json
d=json.load(open('$O/b54_$i.json')); print('bf16 B1', d['value'], d['phases']); k=d['roofline']['decode_kernels_in_run']; print({n:v['avg_us'] for n,v in k.items()})"; done
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --weights fp8 2>/dev/null | tail -1 > $O/b54_fp8.json; python -c "
import json
d=json.load(open('$O/b54_fp8.json')); print('fp8 B1', d['value'], d['phases']['decode_ms_per_token']); k=d['roofline']['decode_kernels_in_run']; print({n:v['avg_us'] for n,v in k.items()})"
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --weights fp8 --batch 8 2>/dev/null | tail -1 > $O/b54_fp8b8.json; python -c "
import json
d=json.load(open('$O/b54_fp8b8.json')); print('fp8 B8', d['value'], d['phases']['batched_decode_ms_per_step']); k=d['roofline']['decode_kernels_in_run']; print({n:v['avg_us'] for n,v in k.items()})"
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4
