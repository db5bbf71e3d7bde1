#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "sampler" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_batch_gpu.py -q -k "sampl or generate" 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/b32.json; python -c "
import json
d=json.load(open('$O/b32.json')); print(d['value'], d['phases'])"
