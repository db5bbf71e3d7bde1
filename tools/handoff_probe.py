#!/usr/bin/env python3
"""Times the tagged-granule all-gather of tools/handoff_probe.hip (cost of one all-to-all edge inside a persistent
decode layer).  Usage on the GPU box: python tools/handoff_probe.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libhandoff_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "handoff_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.handoff_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
st = torch.cuda.current_stream().cuda_stream
for batched, blocks in ((0, 256), (1, 64), (1, 128), (1, 256)):
    for gpw in (1, 4, 8, 16, 22, 64):            # granules per workgroup: vector of blocks*gpw*4 payload bytes
        vec = torch.zeros(blocks * gpw, dtype=torch.int64, device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        sink = torch.zeros(1, dtype=torch.int64, device="cuda")
        for iters in (20, 220):
            vec.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.handoff_run(vec.data_ptr(), blocks, gpw, iters, err.data_ptr(), sink.data_ptr(), st, batched)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            if iters == 20:
                t20 = ms
            else:
                per = (ms - t20) / 200.0 * 1e3
                print(f"{'8 polls in flight' if batched else '1 poll in flight '} blocks {blocks:3d} granules/wg {gpw:3d} (vector {blocks * gpw * 4 / 1024:6.1f} KB payload): "
                      f"{per:6.2f} us per all-gather round  err={int(err.item())} rc={rc}", flush=True)
