#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "polling_mergers or whole_context" 2>&1 | tail -6
timeout 600 python tools/attn_probe.py 1 2300 2>&1 | grep -v amdgpu | grep "split\|B=" 
for f in 1 0 1 0; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --tune attn_fused=$f 2> $O/b44_$f.err | tail -1 > $O/b44_fused$f.json
python -c "
import json
d=json.load(open('$O/b44_fused$f.json')); k=d['roofline']['decode_kernels_in_run']; print('fused=$f', d['value'], d['phases']['decode_ms_per_token'], {a: k[a]['avg_us'] for a in k if 'attn' in a})" || tail -3 $O/b44_$f.err
done
