#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "im2col_vt or sampler or vit" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_true_shapes_gpu.py -q -x -k "vit or tower or c2 or C2" 2>&1 | tail -4
timeout 300 python tools/vit_probe.py 2>&1 | grep -v amdgpu | tail -12
