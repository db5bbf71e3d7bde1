#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/attn_probe.py 1 2300 > $O/attn_probe_b1.txt 2>&1; cat $O/attn_probe_b1.txt
timeout 600 python tools/attn_probe.py 2 2300 > $O/attn_probe_b2.txt 2>&1; cat $O/attn_probe_b2.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_batch_gpu.py tests/test_fp16_gpu.py -q -x > $O/t12_models.txt 2>&1; grep -v "^  File" $O/t12_models.txt | head -60
