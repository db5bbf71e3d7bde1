#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_fp16_gpu.py -x -q -s 2>&1 | tail -30 > $O/t9_fp16.txt
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --dtype fp16 2> $O/b9_fp16.err | tail -1 > $O/b9_fp16.json
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2> $O/b9_bf16.err | tail -1 > $O/b9_bf16.json
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --dtype fp16 --batch 8 2> $O/b9_fp16_b8.err | tail -1 > $O/b9_fp16_b8.json
for n in fp16 bf16 fp16_b8; do python -c "
import json
d=json.load(open('$O/b9_$n.json')); print('$n', d['value'], d['dtype'][:8], d['phases'])"; done
cat $O/t9_fp16.txt | grep -v amdgpu
timeout 2600 python -m pytest tests -m gpu -q --deselect tests/test_fp16_gpu.py 2>&1 | tail -25 > $O/pytest_gpu_9.txt; tail -8 $O/pytest_gpu_9.txt
