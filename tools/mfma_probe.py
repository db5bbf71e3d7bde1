"""On-box MFMA ceilings (dense bf16 16x16x32 / 32x32x16, fp8 16x16x128): python tools/mfma_probe.py"""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_probe.so"))
lib.mfma_probe_run.restype = C.c_double
lib.mfma_probe_run.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "v_mfma_f32_16x16x32_bf16", 1: "v_mfma_f32_32x32x16_bf16", 2: "v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)"}
for mode in (0, 1, 2):
    for wgs_per_cu in (1, 2, 4):
        blocks, iters = 256 * wgs_per_cu, 20000
        lib.mfma_probe_run(mode, out.data_ptr(), blocks, 100, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fl = lib.mfma_probe_run(mode, out.data_ptr(), blocks, iters, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"{names[mode]:42s} {wgs_per_cu} x 4 waves per CU: {fl / ms / 1e9:8.1f} TFLOP/s ({ms:.2f} ms)", flush=True)
