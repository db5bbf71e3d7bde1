"""Host-side probe of the GPU box's CPU (for bench.py's cpu_baseline): cgroup limits, memory bandwidth and GEMM / GEMV rates
of torch-CPU at several thread counts.  Prints one JSON object."""
import json, os, time
import torch
import torch.nn.functional as F

def rd(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None

out = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cpu.max": rd("/sys/fs/cgroup/cpu.max"),
       "cpuset": rd("/sys/fs/cgroup/cpuset.cpus.effective"), "mem.max": rd("/sys/fs/cgroup/memory.max"),
       "numa_nodes": len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]) if os.path.isdir("/sys/devices/system/node") else None,
       "omp": os.environ.get("OMP_NUM_THREADS"), "torch_threads": torch.get_num_threads()}

def timeit(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps

res = {}
a = torch.empty(256 * 2 ** 20, dtype=torch.float32).fill_(1.0)      # 1 GiB
b = torch.empty_like(a)
W = torch.randn(11008, 4096)
Wl = [W.clone() for _ in range(8)]
x1 = torch.randn(1, 4096)
X = torch.randn(2168, 4096)
for nt in (8, 16, 32, 64):
    if nt > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(nt)
    t = timeit(lambda: b.copy_(a))
    r = {"copy_GBps": round(2 * a.numel() * 4 / t / 1e9, 1)}
    t = timeit(lambda: [F.linear(x1, w) for w in Wl])
    r["linear_M1_GBps"] = round(8 * W.numel() * 4 / t / 1e9, 1)
    t = timeit(lambda: [torch.mv(w, x1[0]) for w in Wl])
    r["mv_GBps"] = round(8 * W.numel() * 4 / t / 1e9, 1)
    t = timeit(lambda: F.linear(X, W), reps=2)
    r["gemm_M2168_TFLOPs"] = round(2 * 2168 * 4096 * 11008 / t / 1e12, 3)
    res[nt] = r
# which formulation of a one-row product does this host's BLAS run at memory speed?
m1 = {}
for nt in (16, 32):
    torch.set_num_threads(nt)
    r = {}
    for M in (2, 4, 8, 16, 32):
        xp = x1.expand(M, -1).contiguous()
        t = timeit(lambda: [F.linear(xp, w) for w in Wl])
        r[f"linear_M{M}_GBps"] = round(8 * W.numel() * 4 / t / 1e9, 1)
    t = timeit(lambda: [(w * x1).sum(-1) for w in Wl])
    r["mul_sum_GBps"] = round(8 * W.numel() * 4 / t / 1e9, 1)
    t = timeit(lambda: [torch.cat([torch.mv(c, x1[0]) for c in w.chunk(16)]) for w in Wl])
    r["chunked_mv_GBps"] = round(8 * W.numel() * 4 / t / 1e9, 1)
    Wb = [w.to(torch.bfloat16) for w in Wl]
    xb = x1.to(torch.bfloat16)
    t = timeit(lambda: [F.linear(xb, w) for w in Wb])
    r["bf16_linear_M1_GBps"] = round(8 * W.numel() * 2 / t / 1e9, 1)
    Xb = X.to(torch.bfloat16)
    t = timeit(lambda: F.linear(Xb, Wb[0]), reps=2)
    r["bf16_gemm_M2168_TFLOPs"] = round(2 * 2168 * 4096 * 11008 / t / 1e12, 3)
    t = timeit(lambda: [w.sum() for w in Wl])
    r["read_sum_GBps"] = round(8 * W.numel() * 4 / t / 1e9, 1)
    m1[nt] = r
out["one_row_products"] = m1
out["by_threads"] = res
print(json.dumps(out))
