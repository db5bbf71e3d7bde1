import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from teochat_amd import _lib as L
from tests import _gpu as G
lib = L.load(); bf = torch.bfloat16
ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws")
def timeit(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K in (("proj1", 2048, 4096, 1024), ("vit fc1", 2056, 4096, 1024), ("vit qkv", 2056, 3072, 1024)):
    A = torch.randn(M, K, device="cuda").to(bf)
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(bf) for _ in range(8)]
    bias = torch.randn(N, device="cuda").to(bf)
    Cc = torch.empty(M, N, dtype=bf, device="cuda")
    C32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    cnt = [0]
    for tag, act, b, out, od in (("bias+gelu_erf", L.ACT_GELU_ERF, bias, Cc, L.TEO_BF16), ("bias+quick_gelu", L.ACT_QUICK_GELU, bias, Cc, L.TEO_BF16), ("bias", 0, bias, Cc, L.TEO_BF16), ("nothing", 0, None, Cc, L.TEO_BF16)):
        def run():
            W = Ws[cnt[0] % 8]; cnt[0] += 1
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), G.p(b), None, G.p(out), M, N, K, K, N, act, 0, L.TEO_BF16, od, G.p(ws), G.stream()), "gemm")
        t = min(timeit(run) for _ in range(3))
        print(f"{name} {tag:16s} [{lib.teo_last_kernel().decode()}] {t:6.1f} us", flush=True)
