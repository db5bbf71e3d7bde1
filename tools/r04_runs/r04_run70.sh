#!/bin/bash
# what is the combine launch made of?  timing diagnostics (wrong results): 1 empty kernel, 2 position + one round of record loads + store, 3 no position load
O=gpurun_out/r04; mkdir -p $O
for d in 0 1 2 3 0; do timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --tune attn_combine_dbg=$d 2>/dev/null | tail -1 > $O/b70_$d.json; python -c "
import json
d=json.load(open('$O/b70_$d.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('combine_dbg=$d', d['value'], p['decode_ms_per_token'], {n:v['avg_us'] for n,v in k.items() if 'attn' in n or n=='o_gemv'})"; done
