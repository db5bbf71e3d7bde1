#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ksplit or gemm" 2>&1 | tail -5 > $O/t7_gemm.txt
timeout 1500 python -m pytest tests/test_true_shapes_gpu.py tests/test_bf16_walk_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -160 > $O/t7_true_shapes.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_shard_frames_gpu.py tests/test_batch_gpu.py -x -q 2>&1 | tail -6 > $O/t7_model.txt
python - > $O/vit_ksplit.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, '.')
from teochat_amd import _lib as L
from tests import _gpu as G
lib = L.load(); bf = torch.bfloat16
def timeit(fn, iters=24, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K in (("vit out", 2056, 1024, 1024), ("vit fc2", 2056, 1024, 4096), ("vit qkv", 2056, 3072, 1024), ("vit fc1", 2056, 4096, 1024)):
    A = torch.randn(M, K, device="cuda").to(bf)
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
    bias = torch.randn(N, device="cuda").to(bf); res = torch.randn(M, N, device="cuda").to(bf)
    cnt = [0]
    def run():
        cnt[0] += 1
        return G.gemm(A, Ws[cnt[0] % 8], bias=bias, res=res)
    line = name
    for ks in (0, 1, 2, 3):
        lib.teo_tune_set(b"gemm_ksplit", ks)
        run(); k = lib.teo_last_kernel().decode()
        line += f" | ksplit={ks} [{k}] {min(timeit(run) for _ in range(3)):6.1f} us"
    lib.teo_tune_reset()
    print(line, flush=True)
PY
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2> $O/b7_c3.err | tail -1 > $O/b7_c3.json
python -c "
import json
d=json.load(open('$O/b7_c3.json')); print('C3', d['value'], d['phases'])"
timeout 500 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch 8 --weights fp8 2> $O/b7_b8fp8.err | tail -1 > $O/b7_b8fp8.json
python -c "
import json
d=json.load(open('$O/b7_b8fp8.json')); print('b8fp8', d['value'], d['phases'])"
cat $O/t7_gemm.txt $O/t7_model.txt; tail -4 $O/t7_true_shapes.txt; grep -v amdgpu $O/vit_ksplit.txt
