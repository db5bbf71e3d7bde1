"""Issue cost of the conversion / pack / MFMA instructions of the batched fp8 decode GEMMs: python tools/cvt_probe.py
(ns per wave instruction and cycles at the clock the baseline v_and_b32 implies: a full-rate VALU op is 4 cycles per wave64)."""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libcvt_probe.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "cvt_probe.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.cvt_probe_run.restype = C.c_int
lib.cvt_probe_run.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "v_and_b32 (baseline, full rate)", 1: "v_cvt_scalef32_pk_bf16_fp8", 2: "v_cvt_pk_f32_fp8", 3: "v_cvt_pk_bf16_f32", 4: "v_perm_b32",
         5: "v_cvt_scalef32_pk_f16_fp8", 6: "v_lshl_or_b32", 7: "v_mfma_f32_16x16x32_bf16", 8: "v_mfma_scale_f32_16x16x128_f8f6f4 fp8 x fp8",
         9: "v_mfma_scale_f32_16x16x128_f8f6f4 fp8 x bf8", 10: "v_cvt_scalef32_pk_bf8_bf16", 11: "v_dot2c_f32_bf16"}
base = {}
for waves in (1, 2):
    threads = 256 * waves
    for mode in sorted(names):
        iters = 4000
        lib.cvt_probe_run(mode, out.data_ptr(), 256, threads, 50, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = lib.cvt_probe_run(mode, out.data_ptr(), 256, threads, iters, st)
        e1.record(); torch.cuda.synchronize()
        ns = e0.elapsed_time(e1) * 1e6 / (iters * n * waves)          # per wave instruction on one SIMD
        if mode == 0:
            base[waves] = ns
        print(f"{waves} wave(s) per SIMD  {names[mode]:46s} {ns:7.2f} ns per wave instruction = {4.0 * ns / base[waves]:6.1f} cycles", flush=True)
