#!/bin/bash
# GPU box: whole-context batched decode attention -- tests, then A/B bench lines
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn_decode" 2>&1 | tail -6 > $O/t3_attn.txt
timeout 900 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -6 > $O/t3_batch.txt
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2> $O/b3_$name.err | tail -1 > $O/b3_$name.json; python - <<PY
import json
try:
    d=json.load(open("$O/b3_$name.json")); print("$name", d["value"], d["phases"].get("batched_decode_ms_per_step"), d["phases"].get("decode_ms_per_token"))
except Exception as e: print("$name failed", e)
PY
}
run b8fp8_whole --batch 8 --weights fp8
run b8fp8_split --batch 8 --weights fp8 --tune attn_whole=0
run b8fp8_whole128 --batch 8 --weights fp8 --tune attn_chunk=128
run b8fp8_whole32 --batch 8 --weights fp8 --tune attn_chunk=32
run b8_whole --batch 8
run b8_split --batch 8 --tune attn_whole=0
run b16fp8_whole --batch 16 --weights fp8
run b16fp8_split --batch 16 --weights fp8 --tune attn_whole=0
run b4fp8_whole --batch 4 --weights fp8
run b4fp8_split --batch 4 --weights fp8 --tune attn_whole=0
cat $O/t3_attn.txt $O/t3_batch.txt
