#!/bin/bash
# end-of-round run: the full GPU suite, then the judged measurements (tools/r04_profiles.sh + the batched PMC traffic + the driver-style line)
O=gpurun_out/r04; mkdir -p $O gpurun_out/r04p
timeout 2800 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu_final.txt; tail -5 $O/pytest_gpu_final.txt
bash tools/r04_profiles.sh ${1:-unknown} | tail -30
bash tools/pmc_batch_traffic.sh ${1:-unknown} > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> /dev/null | tail -1 > gpurun_out/r04p/bench_driver_style.json
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --dtype fp16 --batch 8 2> /dev/null | tail -1 > gpurun_out/r04p/bench_fp16_batch8.json
timeout 300 python tools/flash_probe.py 2168 2>&1 | grep -v amdgpu > gpurun_out/r04p/flash_probe.txt
timeout 300 python tools/sampler_time.py 2>&1 | grep -v amdgpu > gpurun_out/r04p/sampler_time.txt
timeout 300 python tools/vit_probe.py 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/r04p/vit_probe.txt
ls gpurun_out/r04p | wc -l
