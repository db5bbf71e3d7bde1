#!/bin/bash
# HBM traffic of EVERY kernel of the decode step from PMC counters, inside a real bench.py run (round 3).
# Collected as MI355X_MICROARCH.md (section HBM) prescribes: separate rocprofv3 --pmc passes (nothing but --kernel-trace beside
# them), FETCH_SIZE doubled on gfx950 (128-B requests are tallied at 64 B), WRITE_SIZE as is, TCC_EA0_RDREQ_sum x 128 B as the
# cross-check.  The program itself follows `--` (no wrapper hop).  Writes gpurun_out/r06p/r06_pmc_decode_traffic.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
COMMIT=${1:-unknown}
mkdir -p $ROOT/gpurun_out/r06p
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  rm -rf /tmp/pmcd_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcd_$C -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --new 12 --no-cpu-baseline --no-graph > /tmp/pmcd_$C.log 2>&1
done
python3 - "$COMMIT" <<'PY'
import glob, json, os, sqlite3, sys
KERNELS = {   # name -> (substring test on the kernel name, algorithmic bytes per launch at LLaMA-2-7B bf16, ctx ~2175)
    "gateup_gemv": (lambda n: "gemv_kernel<" in n and "true, true, true" in n, 2 * 11008 * 4096 * 2),
    "qkv_rope_gemv": (lambda n: "gemv_qkv_rope_kernel<" in n, 12288 * 4096 * 2),
    "o_gemv": (lambda n: "gemv_splitk_kernel<" in n and ", 2, 2, true>" in n, 4096 * 4096 * 2),
    "down_gemv": (lambda n: "gemv_splitk_kernel<" in n and ", 2, 6, true>" in n, 4096 * 11008 * 2),
    "lm_head_gemv": (lambda n: "gemv_kernel<" in n and "float" in n.split("<")[1].split(",")[1], 32000 * 4096 * 2),
    "attn_decode_partial": (lambda n: "attn_decode_partial_kernel<" in n, 2 * 32 * 2175 * 128 * 2),
    "attn_decode_combine": (lambda n: "attn_decode_combine_kernel<" in n, None),
}
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
    dbs = glob.glob(f"/tmp/pmcd_{c}/**/*.db", recursive=True)
    if not dbs:
        raw[c] = {}
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = list(cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (c,)))
    raw[c] = {r[0]: (r[1], r[2]) for r in rows}
out = {"source": "tools/pmc_decode_traffic.sh: three separate rocprofv3 --kernel-trace --pmc passes over `python3 bench.py --steps 1 --warmup 0 --new 12 --no-cpu-baseline --no-graph` (rounds 3-4)",
       "commit": sys.argv[1], "csrc_sha16": __import__("subprocess").check_output(["python3", os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/src_hash.py"]).decode().strip(),
       "correction": "gfx950: read bytes = 2 * FETCH_SIZE * 1024 (128-B requests tallied at 64 B) + WRITE_SIZE * 1024; cross-check TCC_EA0_RDREQ_sum * 128 B",
       "kernels": {}}
for key, (match, alg) in KERNELS.items():
    e = {"algorithmic_bytes": alg}
    try:
        names = [n for n in raw["FETCH_SIZE"] if match(n)]
        name = max(names, key=lambda n: raw["FETCH_SIZE"][n][0])
        f, w, r = raw["FETCH_SIZE"][name], raw["WRITE_SIZE"].get(name, (0, 0.0)), raw["TCC_EA0_RDREQ_sum"].get(name, (0, 0.0))
        e.update({"kernel": name[:140], "launches_sampled": f[0], "FETCH_SIZE_KB_avg": f[1], "WRITE_SIZE_KB_avg": w[1], "TCC_EA0_RDREQ_sum_avg": r[1],
                  "hbm_bytes_per_launch": int(2 * f[1] * 1024 + w[1] * 1024), "rdreq_bytes_per_launch": int(r[1] * 128)})
        if alg:
            e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / alg, 4)
    except Exception as ex:  # noqa: BLE001
        e["error"] = str(ex)
    out["kernels"][key] = e
gu = out["kernels"].get("gateup_gemv", {})
out["hbm_bytes_per_launch"] = gu.get("hbm_bytes_per_launch")        # the dominant kernel: bench.py's roofline.traffic
out["kernel"] = "gemv_kernel<bf16,bf16,R=2,U=4,NT,SWIGLU> N=22016 K=4096 (decode rmsnorm + gate/up + SwiGLU)"
json.dump(out, open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r06p/r06_pmc_decode_traffic.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY
