#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "skinny" 2>&1 | tail -8 > $O/t5_skinny.txt
timeout 300 python tools/skinny_probe.py fp8 8 > $O/skinny_probe2_fp8_8.txt 2>&1
SK_SHAPES=decode SK_TILES=1 timeout 400 python tools/bench_kernels.py skinny > $O/skinny_v2.txt 2>&1
SK_SHAPES=decode SK_TILES=1 SK_STREAM=2 timeout 400 python tools/bench_kernels.py skinny > $O/skinny_v2_force.txt 2>&1
run() { name=$1; shift; timeout 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2> $O/b5_$name.err | tail -1 > $O/b5_$name.json; python - <<PY
import json
try:
    d=json.load(open("$O/b5_$name.json")); print("$name", d["value"], d["phases"].get("batched_decode_ms_per_step"), d["phases"].get("decode_ms_per_token"))
except Exception as e: print("$name failed", e)
PY
}
run b8fp8 --batch 8 --weights fp8
run b8fp8_force --batch 8 --weights fp8 --tune skinny_stream=2
run b8 --batch 8
run b8_force --batch 8 --tune skinny_stream=2
run b16fp8 --batch 16 --weights fp8
run b16 --batch 16
timeout 900 python -m pytest tests/test_batch_gpu.py -x -q 2>&1 | tail -4 > $O/t5_batch.txt
cat $O/t5_skinny.txt $O/t5_batch.txt; grep "stream kernel" -A5 $O/skinny_probe2_fp8_8.txt | head -14; cat $O/skinny_v2.txt | grep fp8
