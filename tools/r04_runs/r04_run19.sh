#!/bin/bash
O=gpurun_out/r04; mkdir -p $O gpurun_out/r04p
bash tools/pmc_decode_traffic.sh ${1:-unknown} > $O/pmc_traffic.log 2>&1; tail -c 600 $O/pmc_traffic.log
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> $O/b19_driver.err | tail -1 > gpurun_out/r04p/bench_driver_style.json
timeout 900 python -m pytest tests/test_fp16_gpu.py tests/test_realistic_checkpoint_gpu.py -q -s 2>&1 | grep -v amdgpu > $O/t19_fp16_realistic.txt; grep "^\[\|passed\|failed" $O/t19_fp16_realistic.txt | cut -c1-400
timeout 2800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu_19.txt; tail -6 $O/pytest_gpu_19.txt
