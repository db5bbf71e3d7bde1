import ctypes as C, os, sys, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream_probe.so"))
lib.probe_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
lib.probe_rows_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p]
big = 4 << 30
buf = torch.randint(0, 255, (big,), dtype=torch.uint8, device="cuda")
out = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n):
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# (A) contiguous-per-wave streaming, short kernels rotating through the 4 GB buffer (no cache reuse)
for size_mb in (33, 90, 180, 1024):
    size = size_mb << 20
    nbuf = big // size
    for blocks in (512, 1024, 2048):
        ms = timed(lambda i: lib.probe_run(1, buf.data_ptr() + (i % nbuf) * size, out.data_ptr(), size, blocks, st), 20)
        per_wave = size // (blocks * 4) // 8192 * 8192
        print(f"contig nt  {size_mb:5d} MB blocks {blocks:5d}: {ms * 1e3:7.1f} us {per_wave * blocks * 4 / ms / 1e6:8.1f} GB/s", flush=True)

# (B) GEMV-like row access at the gate/up shape (22016 rows x 8 KB) and down shape (4096 x 22016 B)
for (N, rb, name) in ((22016, 8192, "gateup"), (4096, 22016, "down"), (12288, 8192, "qkv")):
    size = N * rb
    nbuf = big // size
    for (R, U) in ((4, 2), (2, 4), (1, 8), (8, 1), (2, 2)):
        for blocks in (1024, 2048):
            ms = timed(lambda i: lib.probe_rows_run(R, U, buf.data_ptr() + (i % nbuf) * size, out.data_ptr(), N, rb, blocks, st), 20)
            print(f"rows {name:7s} R={R} U={U} blocks {blocks:5d}: {ms * 1e3:7.1f} us {size / ms / 1e6:8.1f} GB/s", flush=True)
