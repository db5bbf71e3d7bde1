#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 > $O/b58_$tag.json; python -c "
import json
d=json.load(open('$O/b58_$tag.json')); p=d['phases']; k=d['roofline']['decode_kernels_in_run']; print('$tag', d['value'], p.get('batched_decode_ms_per_step', p['decode_ms_per_token']), {n:v['avg_us'] for n,v in k.items() if 'attn' in n})"; }
for B in 4 8 16; do
  run b${B}fp8_whole --batch $B --weights fp8 --tune attn_whole=2
  run b${B}fp8_split64 --batch $B --weights fp8 --tune attn_whole=0 --tune attn_chunk=64
  run b${B}fp8_split128 --batch $B --weights fp8 --tune attn_whole=0 --tune attn_chunk=128
  run b${B}fp8_split256 --batch $B --weights fp8 --tune attn_whole=0 --tune attn_chunk=256
done
run b8fp8_whole_again --batch 8 --weights fp8 --tune attn_whole=2
run b8fp8_split128_again --batch 8 --weights fp8 --tune attn_whole=0 --tune attn_chunk=128
