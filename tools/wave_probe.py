"""Device check that common.h's wave reductions (permlane swaps + DPP) give the bits of the __shfl_xor butterfly in every lane,
and what a reduction costs in either form.  usage: python tools/wave_probe.py"""
import ctypes as C, os, subprocess, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libwave_probe.so")
if not os.path.exists(SO):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(HERE, "..", "include"),
                    os.path.join(HERE, "wave_probe.hip"), "-o", SO], check=True)
lib = C.CDLL(SO)
lib.wave_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
lib.wave_lat.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(7)
n_waves = 1 << 16
cases = {
    "normal": torch.randn(n_waves, 64, device="cuda", generator=g),
    "wide exponents": torch.randn(n_waves, 64, device="cuda", generator=g) * torch.exp2(torch.randint(-40, 40, (n_waves, 64), device="cuda", generator=g).float()),
    "signed zeros / inf / denormals": torch.tensor([0.0, -0.0, float("inf"), -float("inf"), 1e-40, -1e-40, 1.0, -1.0], device="cuda")[
        torch.randint(0, 8, (n_waves, 64), device="cuda", generator=g)],
    "cancelling": (torch.randn(n_waves, 32, device="cuda", generator=g).repeat(1, 2) * torch.tensor([1.0] * 32 + [-1.0] * 32, device="cuda")
                   + 1e-7 * torch.randn(n_waves, 64, device="cuda", generator=g)),
}
ok = True
for name, x in cases.items():
    diff = torch.zeros(2, dtype=torch.int32, device="cuda")
    x = x.contiguous().float()
    assert lib.wave_probe(x.data_ptr(), diff.data_ptr(), n_waves, st) == 0
    torch.cuda.synchronize()
    d = diff.tolist()
    print(f"{name:32s}: {n_waves} waves x 64 lanes, wave sum / max and the 16- / 32- / 8-lane and cross-group butterflies in every lane: {'bit-identical' if d[0] == 0 else f'DIFFERENT (mask {d[0]}, {d[1]} lanes)'}")
    ok = ok and d[0] == 0
out = torch.zeros(64, device="cuda")
cyc = torch.zeros(1, dtype=torch.int64, device="cuda")
for which, name in ((0, "__shfl_xor (ds_bpermute)"), (1, "permlane swaps + DPP")):
    for _ in range(2):
        assert lib.wave_lat(which, out.data_ptr(), cyc.data_ptr(), 1000, st) == 0
        torch.cuda.synchronize()
    print(f"dependent chain of 1000 wave_sum + fma, one wave: {name:28s} {int(cyc.item()) / 1000:7.1f} shader clocks per reduction")
sys.exit(0 if ok else 1)
