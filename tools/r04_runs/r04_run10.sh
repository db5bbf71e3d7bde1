#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 300 python tools/xcc_probe.py > $O/xcc_probe.txt 2>&1; cat $O/xcc_probe.txt
timeout 900 python -m pytest tests/test_fp16_gpu.py -q -s 2>&1 | grep -v amdgpu | tail -40 > $O/t10_fp16.txt; tail -15 $O/t10_fp16.txt
timeout 1500 python -m pytest tests/test_realistic_checkpoint_gpu.py -q -s 2>&1 | grep -v amdgpu | tail -40 > $O/t10_realistic.txt; cat $O/t10_realistic.txt
timeout 600 python -m pytest tests/test_bench_contract_gpu.py -q 2>&1 | tail -8 > $O/t10_bench_contract.txt; cat $O/t10_bench_contract.txt
