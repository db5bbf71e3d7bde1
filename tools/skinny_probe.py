#!/usr/bin/env python3
"""Timeline of the batched-decode GEMM kernels (GPU box): builds tools/libskinny_probe.so (skinny.hip with TRACE on) and runs
layer-like chains qkv -> o -> gate/up -> down at the 7B shapes; prints, per launch, when its workgroups entered, consumed
their first weights, passed each tile barrier and stored, relative to the first workgroup's entry, plus the gap to the
previous launch's last store.  usage: python tools/skinny_probe.py [fp8|bf16] [MB]"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libskinny_probe.so")
SLOTS = 16


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           os.path.join(HERE, "skinny_probe.hip"), "-o", SO]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])


def main():
    fp8 = (sys.argv[1] if len(sys.argv) > 1 else "fp8") == "fp8"
    MB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    lib.skinny_probe_launch.restype = C.c_int
    lib.skinny_probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    dev = "cuda"
    wb = 1 if fp8 else 2
    shapes = [("qkv", 12288, 4096, 0, True, False), ("o", 4096, 4096, 0, False, True), ("gateup", 22016, 4096, 1, True, False),
              ("down", 4096, 11008, 0, False, True)]
    layers = 8
    pool = torch.randint(0, 120, (700 << 20,), dtype=torch.uint8, device=dev)       # > 256 MB Infinity Cache in rotation
    scale = torch.full((32768,), 2.0 ** -9, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for mode, mname in ((0, "tile"), (1, "stream")):
        for with_fuse in (1, 0):
            x4 = torch.randn(MB, 4096, device=dev).to(torch.bfloat16)
            x11 = torch.randn(MB, 11008, device=dev).to(torch.bfloat16)
            h = torch.randn(MB, 4096, device=dev).to(torch.bfloat16)
            hg = torch.empty_like(h)
            g = torch.ones(4096, device=dev).to(torch.bfloat16)
            ssq = torch.rand(MB, 256, device=dev)
            outs = {n: torch.empty(MB, N, dtype=torch.bfloat16, device=dev) for n, N, *_ in shapes}
            launches = []
            off = 0
            for l in range(layers):
                for name, N, K, sw8, takes, gives in shapes:
                    nbytes = N * K * wb
                    if off + nbytes > pool.numel():
                        off = 0
                    launches.append((name, N, K, sw8, takes, gives, off))
                    off += (nbytes + 255) // 256 * 256
            trace = torch.zeros(len(launches), 2048, SLOTS, dtype=torch.int64, device=dev)
            for rep in range(2):                         # second pass is the one read
                trace.zero_()
                torch.cuda.synchronize()
                for i, (name, N, K, sw8, takes, gives, o_) in enumerate(launches):
                    x = x11 if K == 11008 else x4
                    rc = lib.skinny_probe_launch(mode, x.data_ptr(), pool.data_ptr() + o_, scale.data_ptr() if fp8 else None, int(fp8),
                                                 h.data_ptr() if gives else None, (h if gives else outs[name]).data_ptr(), MB, N, K, sw8,
                                                 ssq.data_ptr() if (takes and with_fuse) else None, 256,
                                                 g.data_ptr() if (gives and with_fuse) else None, hg.data_ptr() if (gives and with_fuse) else None,
                                                 ssq.data_ptr() if (gives and with_fuse) else None, trace[i].data_ptr(), 256, st)
                    assert rc == 0
                torch.cuda.synchronize()
            tr = trace.cpu().numpy()
            print(f"== {mname} kernel, {'fp8' if fp8 else 'bf16'} weights, MB={MB}, norm hand-off {'on' if with_fuse else 'off'} (us; relative to the launch's first workgroup entry)")
            prev_end = None
            agg = {}
            for i, (name, N, K, sw8, takes, gives, o_) in enumerate(launches):
                t = tr[i]
                used = t[:, 0] > 0
                t = t[used].astype("float64") / 100.0          # 100 MHz -> us
                e0 = t[:, 0].min()
                rel = lambda col: (t[:, col][t[:, col] > 0] - e0)
                row = {"wgs": int(used.sum()), "entry_last": rel(0).max()}
                if mode == 0:
                    row.update(first_w=rel(1).mean(), loop_done_mean=rel(2).mean(), loop_done_max=rel(2).max(), all_waves_done_mean=rel(3).mean(), all_waves_done_max=rel(3).max(), end_mean=rel(4).mean(), end_max=rel(4).max())
                    end = t[:, 4].max()
                else:
                    ends = []
                    for s_ in range(1, 8):
                        if (t[:, s_] > 0).any():
                            row[f"tile{s_}_mean"] = rel(s_).mean()
                            row[f"tile{s_}_max"] = rel(s_).max()
                    ep = t[:, 9:16]
                    end = ep.max()
                    row["epi_first_mean"] = rel(9).mean()
                    row["end_max"] = end - e0
                row["gap_from_prev_end"] = (e0 - prev_end) if prev_end is not None else float("nan")
                prev_end = end
                if i >= 4:                                       # skip the first layer (cold)
                    a = agg.setdefault(name, [])
                    a.append(row)
            for name, rows in agg.items():
                keys = rows[0].keys()
                print(f"  {name:7s} " + "  ".join(f"{k}={sum(r[k] for r in rows) / len(rows):.2f}" for k in keys))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
