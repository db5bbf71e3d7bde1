#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "last_arriving or whole_context" 2>&1 | tail -8 > $O/t11_attn.txt; cat $O/t11_attn.txt
timeout 600 python tools/attn_probe.py 1 2300 > $O/attn_probe_b1.txt 2>&1; cat $O/attn_probe_b1.txt
timeout 600 python tools/attn_probe.py 2 2300 > $O/attn_probe_b2.txt 2>&1; cat $O/attn_probe_b2.txt
for f in 1 0; do
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --tune attn_fused=$f 2> $O/b11_fused$f.err | tail -1 > $O/b11_fused$f.json
python -c "
import json
d=json.load(open('$O/b11_fused$f.json')); print('fused=$f', d['value'], d['phases']); print(d["roofline"].get("decode_kernels_in_run"))"
done
timeout 900 python -m pytest tests/test_fp16_gpu.py tests/test_model_gpu.py tests/test_batch_gpu.py -q -x 2>&1 | tail -8 > $O/t11_models.txt; cat $O/t11_models.txt
