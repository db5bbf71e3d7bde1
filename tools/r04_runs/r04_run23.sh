#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 300 python tools/flash_probe.py 2168 2>&1 | grep -v amdgpu > $O/flash_probe2.txt; cat $O/flash_probe2.txt
FA_PIPE=0 timeout 300 python tools/flash_probe.py 2168 2>&1 | grep -v amdgpu | grep "wave" | head -4
timeout 600 python tools/bench_kernels.py attn_prefill 2>&1 | grep -v amdgpu > $O/flash_order2.txt; cat $O/flash_order2.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or attn or flash" 2>&1 | tail -12
