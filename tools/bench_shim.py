"""ctypes binding of tools/libteo_bench.so (tools/bench_shim.hip): bench-only timing helpers over the PUBLIC C ABI of
libteo_hip.so.  Not product code: only bench.py and tools/bench_kernels.py load it."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libteo_bench.so")
PP = C.POINTER(C.c_void_p)
_lib = None


def build(hipcc=None, arch="gfx950"):
    """hipcc -shared tools/bench_shim.hip against teochat_amd/libteo_hip.so (rpath-relative)."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, f"--offload-arch={arch}", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(HERE, "bench_shim.hip"), "-o", LIB_PATH,
           "-L" + os.path.join(ROOT, "teochat_amd"), "-lteo_hip", "-Wl,-rpath,$ORIGIN/../teochat_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building tools/libteo_bench.so failed:\n" + r.stderr[-4000:])


def load():
    global _lib
    if _lib is None:
        from teochat_amd import _lib as L
        L.load()                                   # libteo_hip.so first (after torch: one HIP runtime in the process)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        lib.teo_bench_gemv_chain.restype = C.c_int
        lib.teo_bench_gemv_chain.argtypes = [C.c_void_p, PP, PP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint,
                                             C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
        lib.teo_bench_skinny_chain.restype = C.c_int
        lib.teo_bench_skinny_chain.argtypes = [C.c_void_p, PP, PP, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint,
                                               C.c_int, C.POINTER(C.c_float), C.c_void_p]
        _lib = lib
    return _lib
