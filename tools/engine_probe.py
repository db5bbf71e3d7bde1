"""Loader / consumer persistent decode-layer probe (tools/engine_probe.hip).  Usage: python tools/engine_probe.py [layers]"""
import ctypes as C
import os
import sys

import torch

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("ENGINE_PROBE_LIB", "libengine_probe.so")))


class Op(C.Structure):
    _fields_ = [("W", C.c_void_p), ("layer_stride", C.c_longlong), ("N", C.c_int), ("K", C.c_int), ("k_in", C.c_int)]


class EParams(C.Structure):
    _fields_ = [("op", Op * 8), ("nop", C.c_int), ("layers", C.c_int), ("vec", C.c_void_p * 2), ("err", C.c_void_p), ("use_nt", C.c_int), ("mode", C.c_int)]


lib.engine_probe_run.argtypes = [C.POINTER(EParams), C.c_int, C.c_void_p]
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("qkv", 12288, 4096), ("kv", 4608, 4096), ("o", 4096, 4096), ("gateup", 22016, 4096), ("down", 4096, 11264)]
if len(sys.argv) > 2:
    keep = sys.argv[2].split(",")
    shapes = [s for s in shapes if s[0] in keep]
P = EParams()
keepalive = []
total = 0
for i, (name, N, K) in enumerate(shapes):
    w = torch.empty(layers, N, K, dtype=torch.bfloat16, device="cuda")
    w[0].copy_((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
    for l in range(1, layers):
        w[l].copy_(torch.roll(w[0], l, 0))
    keepalive.append(w)
    P.op[i] = Op(w.data_ptr(), N * K, N, K, K)
    total += N * K * 2
P.nop, P.layers = len(shapes), layers
vecs = [torch.zeros(16384, dtype=torch.int64, device="cuda") for _ in range(2)]
P.vec[0], P.vec[1] = vecs[0].data_ptr(), vecs[1].data_ptr()
err = torch.zeros(16, dtype=torch.int32, device="cuda")
P.err = err.data_ptr()
st = torch.cuda.current_stream().cuda_stream
print(f"layer bytes {total / 1e6:.1f} MB x {layers} layers; ops {[s[0] for s in shapes]}", flush=True)
print("mode 0: loaders + consumers + granule all-gathers; 1: consumers only release the slots; 2: the two loader waves alone, free-running", flush=True)
for nt, mode in ((1, 0), (1, 1), (1, 2)):
    P.use_nt, P.mode = nt, mode
    for rep in range(2):
        for v in vecs:
            v.zero_()
        err.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.engine_probe_run(C.byref(P), 256, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"engine (nt={nt} mode={mode}) rep {rep}: rc {rc} err {err.tolist()}  {ms * 1e3 / layers:7.2f} us/layer  {total * layers / ms / 1e9:6.2f} TB/s", flush=True)
