#!/usr/bin/env python3
"""Timeline of the prefill flash-attention kernel's heaviest workgroup (C3: 32 heads x 2168 queries x 128, causal, bf16): per KV-tile
iteration, how long every wave spends staging (LDS writes + global loads issued), in the score MFMAs, in the softmax, in the PV MFMAs and
at the barrier.  usage: python tools/flash_probe.py [S]"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

SO = os.path.join(HERE, "libflash_probe.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           "-I" + os.path.join(os.path.dirname(HERE), "include"), os.path.join(HERE, "flash_probe.hip"), "-o", SO]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])


def vit():
    """the tower's attention: 8 frames x 16 heads x 257 tokens x 64, non-causal"""
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    lib.flash_probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    NIT = lib.flash_probe_trace_iters()
    B, H, S, d = 8, 16, 257, 64
    bf = torch.bfloat16
    q, k, v = (torch.randn(B, H, S, d, device="cuda").to(bf) for _ in range(3))
    vt = G.make_vt(v)
    o = torch.empty(B, S, H * d, dtype=bf, device="cuda")
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o, a.vt = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), vt.data_ptr()
    a.q_bs, a.q_hs, a.q_rs = q.stride(0), q.stride(1), q.stride(2)
    a.k_bs, a.k_hs, a.k_rs = k.stride(0), k.stride(1), k.stride(2)
    a.v_bs, a.v_hs, a.v_rs = v.stride(0), v.stride(1), v.stride(2)
    a.vt_bs, a.vt_hs, a.vt_rs = vt.stride(0), vt.stride(1), vt.stride(2)
    a.o_bs, a.o_rs = o.stride(0), o.stride(1)
    a.batch, a.heads, a.kv_heads, a.head_dim, a.q_len, a.kv_len = B, H, H, d, S, S
    a.causal, a.scale = 0, d ** -0.5
    st = torch.cuda.current_stream().cuda_stream
    trace = torch.zeros(4 * NIT * 6, dtype=torch.int64, device="cuda")
    for _ in range(3):
        assert lib.flash_probe_launch(C.byref(a), 0, None, st, 0) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.flash_probe_launch(C.byref(a), 0, None, st, 0)
    e1.record()
    torch.cuda.synchronize()
    print(f"tower attention (8 x 16 x 257 x 64): {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch back to back")
    for rep in range(2):
        trace.zero_()
        assert lib.flash_probe_launch(C.byref(a), 0, trace.data_ptr(), st, 0) == 0
        torch.cuda.synchronize()
    tr = trace.cpu().view(4, NIT, 6).double() * 0.01
    t0 = tr[:, 0, 0].min()
    print("workgroup 0, wave 0: marks of the 5 iterations, us since its first mark (loop top | DMA issued | scores + softmax | PV | DMA landed | past the barrier):")
    print("   " + " | ".join(" ".join(f"{float(tr[0, t, p] - t0):.2f}" for p in range(6)) for t in range(5)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "vit":
        return vit()
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2168
    if not os.path.exists(SO):
        build()
    lib = C.CDLL(SO)
    lib.flash_probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    NIT = lib.flash_probe_trace_iters()
    B, H, d = 1, 32, 128
    bf = torch.bfloat16
    q = torch.randn(B, H, S, d, device="cuda").to(bf)
    k = torch.randn(B, H, S, d, device="cuda").to(bf)
    v = torch.randn(B, H, S, d, device="cuda").to(bf)
    vt = G.make_vt(v)
    o = torch.empty(B, S, H * d, dtype=bf, device="cuda")
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o, a.vt = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), vt.data_ptr()
    a.q_bs, a.q_hs, a.q_rs = q.stride(0), q.stride(1), q.stride(2)
    a.k_bs, a.k_hs, a.k_rs = k.stride(0), k.stride(1), k.stride(2)
    a.v_bs, a.v_hs, a.v_rs = v.stride(0), v.stride(1), v.stride(2)
    a.vt_bs, a.vt_hs, a.vt_rs = vt.stride(0), vt.stride(1), vt.stride(2)
    a.o_bs, a.o_rs = o.stride(0), o.stride(1)
    a.batch, a.heads, a.kv_heads, a.head_dim, a.q_len, a.kv_len = B, H, H, d, S, S
    a.causal, a.scale = 1, d ** -0.5
    st = torch.cuda.current_stream().cuda_stream
    trace = torch.zeros(4 * NIT * 6, dtype=torch.int64, device="cuda")

    def timed(pair_c, pipe=1, n=20):
        for _ in range(3):
            assert lib.flash_probe_launch(C.byref(a), pair_c, None, st, pipe) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            lib.flash_probe_launch(C.byref(a), pair_c, None, st, pipe)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    print(f"S={S}: kernel {timed(32):.1f} us (in-wave pipeline, mirrored order), {timed(0):.1f} us (pipeline, heavy-first), "
          f"{timed(32, 0):.1f} us (one tile at a time, mirrored order)")
    o_pipe = o.clone()
    assert lib.flash_probe_launch(C.byref(a), 32, None, st, 0) == 0
    torch.cuda.synchronize()
    print("pipeline == one tile at a time, bitwise:", bool(torch.equal(o, o_pipe)))
    pipe = int(os.environ.get("FA_PIPE", "1"))
    for rep in range(2):
        trace.zero_()
        assert lib.flash_probe_launch(C.byref(a), 32, trace.data_ptr(), st, pipe) == 0
        torch.cuda.synchronize()
    tr = trace.cpu().view(4, NIT, 6).double() * 0.01                            # us
    nt = min(NIT, (S + 63) // 64)
    t0 = tr[:, 0, 0].min()
    names = ["DMA issue", "scores(+1) | softmax", "PV", "wait DMA", "barrier"]
    print(f"workgroup 0 (heaviest query block, {nt} KV tiles); per wave: mean us per iteration in each phase, and the iteration period")
    for w in range(4):
        ph = tr[w, :nt, 1:] - tr[w, :nt, :-1]
        per = tr[w, 1:nt, 0] - tr[w, :nt - 1, 0]
        print(f"  wave {w}: " + "  ".join(f"{n} {float(ph[2:nt - 2, i].mean()):.2f}" for i, n in enumerate(names)) +
              f"   period {float(per[2:nt - 2].mean()):.2f} (min {float(per.min()):.2f}, max {float(per.max()):.2f})   "
              f"first mark {float(tr[w, 0, 0] - t0):.2f} us, last barrier {float(tr[w, nt - 1, 5] - t0):.2f} us")
    print("  wave 0, iterations 0..7 and the last 4 (us since the first mark): " +
          " | ".join(" ".join(f"{float(tr[0, t, p] - t0):.2f}" for p in range(6)) for t in list(range(8)) + list(range(nt - 4, nt))))


if __name__ == "__main__":
    main()
