#!/bin/bash
# knob sweep after the DPP butterflies: do the earlier choices still hold?
O=gpurun_out/r04; mkdir -p $O
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b57_$tag.json; python -c "
import json
d=json.load(open('$O/b57_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), {n:v['avg_us'] for n,v in k.items() if 'attn' in n or 'qkv' in n})"; }
run base
run chunk128 --tune attn_chunk=128
run chunk32 --tune attn_chunk=32
run ropeattn --rope-in-attn 1
run fp8 --weights fp8
run fp8_ropeattn --weights fp8 --rope-in-attn 1
run fp8_ropeqkv --weights fp8 --rope-in-attn 0
run b2 --batch 2
run b2_whole --batch 2 --tune attn_whole=2
run b4 --batch 4
run b4_nowhole --batch 4 --tune attn_whole=0
run b8fp8 --batch 8 --weights fp8
run b8fp8_chunk128 --batch 8 --weights fp8 --tune attn_chunk=128
run b8fp8_nowhole --batch 8 --weights fp8 --tune attn_whole=0
