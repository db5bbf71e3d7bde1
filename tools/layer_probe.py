"""Persistent-layer probe driver (tools/layer_probe.hip): one launch walking qkv / KV / o / gate-up / down with counter barriers
vs one launch per stage of the same code.  Usage: python tools/layer_probe.py [layers]"""
import ctypes as C
import os
import sys

import torch

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblayer_probe.so"))


class Stage(C.Structure):
    _fields_ = [("W", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("split", C.c_int), ("norm", C.c_int), ("layer_stride", C.c_longlong)]


class Params(C.Structure):
    _fields_ = [("st", Stage * 8), ("nstage", C.c_int), ("layers", C.c_int), ("act", C.c_void_p * 2), ("counter", C.c_void_p),
                ("err", C.c_void_p), ("prefetch", C.c_int), ("sleep", C.c_int)]


lib.layer_probe_run.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
lib.layer_probe_run_multi.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_size_t, C.c_void_p]

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = 2300
shapes = [("qkv", 12288, 4096, 1, 1), ("kv", 2 * ctx * 4096 * 2 // 8192, 4096, 1, 0), ("o", 4096, 4096, 2, 0), ("gateup", 22016, 4096, 1, 1),
          ("down", 4096, 11008, 4, 0)]
if len(sys.argv) > 2:
    keep = sys.argv[2].split(",")
    shapes = [s for s in shapes if s[0] in keep]
weights = []
P = Params()
total_bytes = 0
for i, (name, N, K, split, norm) in enumerate(shapes):
    w = torch.empty(layers, N, K, dtype=torch.bfloat16, device="cuda")
    w[0].copy_((torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16))
    for l in range(1, layers):
        w[l].copy_(torch.roll(w[0], l, 0))
    weights.append(w)
    P.st[i] = Stage(w.data_ptr(), N, K, split, norm, N * K)
    total_bytes += N * K * 2
P.nstage, P.layers = len(shapes), layers
acts = [torch.randn(4 * 32768, device="cuda") for _ in range(2)]
P.act[0], P.act[1] = acts[0].data_ptr(), acts[1].data_ptr()
counter = torch.zeros(4, dtype=torch.int32, device="cuda")
err = torch.zeros(4, dtype=torch.int32, device="cuda")
P.counter, P.err = counter.data_ptr(), err.data_ptr()
st = torch.cuda.current_stream().cuda_stream
lds = (12288 + 1024) * 4


def timed(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def persistent(waves, blocks):
    counter.zero_()
    rc = lib.layer_probe_run(C.byref(P), waves, blocks, 0, -1, lds, st)
    assert rc == 0, rc


def multi(waves, blocks):
    rc = lib.layer_probe_run_multi(C.byref(P), waves, blocks, lds, st)
    assert rc == 0, rc


print(f"layer bytes {total_bytes / 1e6:.1f} MB x {layers} layers; stages {[s[0] for s in shapes]}", flush=True)
for waves, blocks in ((4, 1024), (16, 256)):
    for pf in (1, 0):
        P.prefetch = pf
        ms = timed(lambda: multi(waves, blocks))
        print(f"multi-launch waves {waves:2d} blocks {blocks:5d} prefetch {pf}: {ms * 1e3 / layers:7.2f} us/layer {total_bytes * layers / ms / 1e9:7.2f} TB/s", flush=True)
for waves, blocks in ((16, 256), (8, 256)):
    for pf in (1, 0):
        P.prefetch = pf
        ms = timed(lambda: persistent(waves, blocks))
        torch.cuda.synchronize()
        print(f"persistent   waves {waves:2d} blocks {blocks:5d} prefetch {pf}: {ms * 1e3 / layers:7.2f} us/layer {total_bytes * layers / ms / 1e9:7.2f} TB/s"
              f" err {int(err[0])}", flush=True)
