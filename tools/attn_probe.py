#!/usr/bin/env python3
"""Batched decode attention at config C5's shape (B conversations x 32 heads x ctx keys, bf16, head_dim 128, RoPE + KV append inside):
the shipped split + combine pair vs the whole-context kernel in 8 / 16-wave and 32 / 64 / 128-key forms, and the same kernels with
the arithmetic removed (PROBE: what the memory system gives this access pattern).  K/V caches rotate over > 1 GB.
usage: python tools/attn_probe.py [B] [ctx]"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libattn_probe.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           os.path.join(HERE, "attn_probe.hip"), "-o", SO]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2300
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    lib.attn_probe_launch.restype = C.c_int
    lib.attn_probe_launch.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p]
    H, d, S = 32, 128, 2560
    nl = 6
    dev, bf = "cuda", torch.bfloat16
    K = torch.randn(nl, B, H, S, d, device=dev).to(bf)
    V = torch.randn(nl, B, H, S, d, device=dev).to(bf)
    VT = torch.zeros(nl, B, H, d, S, device=dev, dtype=bf)
    qkv = torch.randn(B, 3 * H * d, device=dev).to(bf)
    out = torch.zeros(B, H * d, device=dev, dtype=bf)
    part = torch.zeros(B * H * (S // 32) * 130 * 4 + 8192, dtype=torch.uint8, device=dev)      # zeroed: the arrival counters live behind the records
    pos = torch.full((B,), ctx - 1, dtype=torch.int32, device=dev)
    ang = torch.arange(S, dtype=torch.float32)[:, None] * (1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d)))[None]
    cs, sn = ang.cos().to(dev).contiguous(), ang.sin().to(dev).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    nbytes = 2 * B * H * ctx * d * 2

    def run(variant, chunk, waves, reps=30):
        def go(n):
            for i in range(n):
                l = i % nl
                rc = lib.attn_probe_launch(variant, chunk, waves, qkv.data_ptr(), K[l].data_ptr(), V[l].data_ptr(), VT[l].data_ptr(), cs.data_ptr(), sn.data_ptr(),
                                           out.data_ptr(), part.data_ptr(), pos.data_ptr(), S, H, B, 3 * H * d, H * S * d, st)
                assert rc == 0, rc
        go(nl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        go(reps)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    print(f"B={B} ctx={ctx}: {nbytes / 1e6:.1f} MB of K/V per launch")
    for name, v, ch, w in (("split + combine, 128 keys", 0, 128, 0), ("split + combine, 64 keys", 0, 64, 0),
                           ("whole 64 keys, 8 waves", 1, 64, 8), ("whole 64 keys, 16 waves", 1, 64, 16),
                           ("whole 32 keys, 8 waves", 1, 32, 8), ("whole 32 keys, 16 waves", 1, 32, 16), ("whole 128 keys, 8 waves", 1, 128, 8),
                           ("loads only, 64 keys, 8 waves", 2, 64, 8), ("loads only, 64 keys, 16 waves", 2, 64, 16),
                           ("loads only, 32 keys, 16 waves", 2, 32, 16), ("loads only, 128 keys, 8 waves", 2, 128, 8)):
        us = min(run(v, ch, w) for _ in range(3))
        print(f"  {name:34s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
