#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/bench_kernels.py attn_prefill > $O/flash_order.txt 2>&1; grep -v amdgpu $O/flash_order.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or attn or flash" 2>&1 | tail -5
