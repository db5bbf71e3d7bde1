#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 300 python tools/wave_probe.py 2>&1 | grep -v amdgpu | tee $O/wave_probe.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_batch_gpu.py tests/test_fp16_gpu.py tests/test_model_gpu.py -q -x > $O/pytest_56.txt 2>&1; grep "passed\|failed" $O/pytest_56.txt | tail -2
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/b56_$i.json; python -c "
import json
d=json.load(open('$O/b56_$i.json')); print('bf16 B1', d['value'], d['phases']['decode_ms_per_token'], d['phases']['sampled_tokens_per_s']); k=d['roofline']['decode_kernels_in_run']; print({n:v['avg_us'] for n,v in k.items()})"; done
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --weights fp8 --batch 8 2>/dev/null | tail -1 > $O/b56_fp8b8.json; python -c "
import json
d=json.load(open('$O/b56_fp8b8.json')); print('fp8 B8', d['value'], d['phases']['batched_decode_ms_per_step']); k=d['roofline']['decode_kernels_in_run']; print({n:v['avg_us'] for n,v in k.items()})"
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch 16 2>/dev/null | tail -1 > $O/b56_b16.json; python -c "
import json
d=json.load(open('$O/b56_b16.json')); print('bf16 B16', d['value'], d['phases']['batched_decode_ms_per_step']); k=d['roofline']['decode_kernels_in_run']; print({n:v['avg_us'] for n,v in k.items()})"
