#!/bin/bash
mkdir -p gpurun_out/r06m
timeout 900 python3 -m pytest tests/test_gemm_fuzz_gpu.py tests/test_tune_gpu.py -x -q > gpurun_out/r06m/fuzz.txt 2>&1; tail -15 gpurun_out/r06m/fuzz.txt
timeout 600 python3 tools/dispatch_monotone.py --only gateup --hi 512 --lo 64 --step 64 --out gpurun_out/r06m/gu_auto.txt > /dev/null 2>&1; grep gateup gpurun_out/r06m/gu_auto.txt
