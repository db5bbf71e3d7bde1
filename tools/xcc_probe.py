#!/usr/bin/env python3
"""blockIdx.x -> XCD map of 1-D launches (HW_REG_XCC_ID): is it `blockIdx.x % 8` for every grid / workgroup size, alone and right behind
another kernel on the stream?  usage: python tools/xcc_probe.py"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libxcc_probe.so")
if not os.path.exists(SO):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", os.path.join(HERE, "xcc_probe.hip"), "-o", SO], check=True)
lib = C.CDLL(SO)
lib.xcc_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for blocks, threads, lds in ((256, 512, 0), (256, 1024, 0), (264, 256, 0), (1184, 256, 0), (2048, 256, 0), (256, 512, 65536), (512, 512, 65536), (32, 512, 0)):
    res = []
    for trial in range(6):
        out = torch.full((blocks,), -1, dtype=torch.int32, device="cuda")
        if trial % 2:
            big.add_(1)                                     # a busy machine in front of the launch
        assert lib.xcc_probe(out.data_ptr(), blocks, threads, lds, 200 if trial % 3 else 0, st) == 0
        torch.cuda.synchronize()
        o = out.cpu()
        ok = bool((o == (torch.arange(blocks) % 8)).all())
        perm = [int(o[i]) for i in range(min(blocks, 16))]
        # is it at least a consistent function of blockIdx % 8?
        cons = all(len(set(o[r::8].tolist())) == 1 for r in range(8))
        res.append((ok, cons, perm))
    print(f"blocks {blocks:5d} x {threads:4d} threads, lds {lds:6d}: identity map {[r[0] for r in res]}  consistent per residue {[r[1] for r in res]}  first 16: {res[0][2]}", flush=True)
