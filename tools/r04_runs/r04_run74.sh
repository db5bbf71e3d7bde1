#!/bin/bash
# soak at the end of the round: the GPU suite twice more on this box, the two fuzzers, the wave probe
O=gpurun_out/r04; mkdir -p $O
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/soak_$i.txt 2>&1; grep "passed\|failed" $O/soak_$i.txt | tail -1; done
timeout 600 python tools/sampler_fuzz.py 2>&1 | grep -v amdgpu | tail -1
timeout 900 python tools/flash_fuzz.py 2>&1 | grep -v amdgpu | tail -2
timeout 300 python tools/wave_probe.py 2>&1 | grep -v amdgpu | tail -6
