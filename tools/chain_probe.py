#!/usr/bin/env python3
"""Probe of the overlapped decode step: host enqueue time vs device time, chain vs ordered launches of the same kernels, vs the
round-2 graph replay.  python tools/chain_probe.py"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from teochat_amd.builder import load_pretrained_model  # noqa: E402

_, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device="cuda:0", dtype=torch.bfloat16, max_seq=2560)
eng = model.engine
lib = eng.lib
emb = torch.randn(2168, 4096, device="cuda:0").to(torch.bfloat16) * 0.02


def arm():
    eng.reset_cache()
    lg = eng.prefill(emb, last_only=True)
    eng.decode_begin(5)
    torch.cuda.synchronize()


def run_chain(n, label, **tune):
    for k, v in tune.items():
        L.check(lib.teo_tune_set(k.encode(), v), k)
    arm()
    ws = eng._workspace("decode", lib.teo_llama_decode_workspace_bytes(C.byref(eng.llama_desc)))
    st = C.c_void_p(eng.stream.cuda_stream)
    with torch.cuda.stream(eng.stream):
        L.check(lib.teo_llama_decode_steps(C.byref(eng.llama_desc), C.byref(eng.decode_state), ws.data_ptr(), ws.numel(), 4, eng.cache_len, st), "warm")
        eng.cache_len += 4
        eng.stream.synchronize()
        t0 = time.perf_counter()
        L.check(lib.teo_llama_decode_steps(C.byref(eng.llama_desc), C.byref(eng.decode_state), ws.data_ptr(), ws.numel(), n, eng.cache_len, st), "steps")
        t1 = time.perf_counter()
        eng.stream.synchronize()
        t2 = time.perf_counter()
        eng.cache_len += n
    flag = C.c_int(0)
    L.check(lib.teo_llama_decode_chain_error(C.byref(eng.llama_desc), ws.data_ptr(), ws.numel(), C.byref(flag), st), "err")
    print(f"{label}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/step ({1e6 * (t1 - t0) / n / 226:.2f} us per launch), total {1e3 * (t2 - t0) / n:.3f} ms/step, err={flag.value}", flush=True)


run_chain(32, "chain (any-order launches)")
run_chain(32, "chain, 1024 blocks", decode_chain_blocks=1024)
run_chain(32, "chain, 256 blocks", decode_chain_blocks=256)
L.check(lib.teo_tune_set(b"decode_chain_blocks", 512), "t")
L.check(lib.teo_tune_set(b"decode_chain", 0), "t")
arm()
eng.decode_steps(8)
torch.cuda.synchronize(); t = time.perf_counter()
eng.decode_steps(64)
torch.cuda.synchronize()
print(f"round-2 graph replay: {(time.perf_counter() - t) / 64 * 1e3:.3f} ms/step")
L.check(lib.teo_tune_set(b"decode_chain", 1), "t")
for name, (per, us) in (lambda: (arm(), eng.decode_steps_profiled(2), eng.decode_steps_profiled(4))[2])().items():
    print(f"  ordered {name}: {per} x {us:.2f} us")
