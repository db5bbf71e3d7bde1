#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "sampl" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_batch_gpu.py -q -x 2>&1 | tail -3
timeout 300 python tools/sampler_time.py 2>&1 | grep -v amdgpu | tee $O/sampler_time_53.txt
timeout 600 python tools/sampler_fuzz.py 2>&1 | grep -v amdgpu | tail -5 | tee $O/sampler_fuzz_53.txt
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/b53.json; python -c "
import json
d=json.load(open('$O/b53.json')); print('bf16 B1', d['value'], d['phases'])"
