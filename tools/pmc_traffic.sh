#!/bin/bash
# HBM traffic of the decode gate/up GEMV from PMC counters, collected as MI355X_MICROARCH.md (section HBM) prescribes:
# separate rocprofv3 --pmc passes (no trace domains beside --kernel-trace), FETCH_SIZE doubled on gfx950 (128-B requests
# are tallied at 64 B), WRITE_SIZE as is, TCC_EA0_RDREQ_sum x 128 B as the cross-check.  Writes gpurun_out/r02/r02_pmc_gemv_gateup.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
COMMIT=${1:-unknown}
mkdir -p $ROOT/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  rm -rf /tmp/pmct_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmct_$C -o p -- python3 $ROOT/tools/bench_kernels.py gemv > /tmp/pmct_$C.log 2>&1
done
python3 - "$COMMIT" <<'PY'
import glob, json, sqlite3, sys
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
    db = glob.glob(f"/tmp/pmct_{c}/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (c,)))
    # the gate/up launch: SWIGLU variant of gemv_kernel (last template flag true)
    pick = [r for r in rows if "gemv_kernel<" in r[0] and r[0].rstrip(">( ").endswith("true, true, true") or "true, true, true>" in r[0]]
    pick = [r for r in pick if "gemv_kernel<" in r[0]]
    vals[c] = {"kernel": pick[0][0][:120], "n": pick[0][2], "avg": pick[0][3]} if pick else {"rows": [r[0][:80] for r in rows]}
alg = 2 * 11008 * 4096 * 2
out = {"kernel": "gemv_kernel<bf16,bf16,R=2,U=4,NT,SWIGLU> N=22016 K=4096 (decode rmsnorm + gate/up + SwiGLU)",
       "source": "tools/pmc_traffic.sh: three separate rocprofv3 --kernel-trace --pmc passes over tools/bench_kernels.py gemv (round 2)",
       "commit": sys.argv[1], "raw": vals, "algorithmic_bytes_per_launch": alg}
try:
    fetch_kb, write_kb, rd = vals["FETCH_SIZE"]["avg"], vals["WRITE_SIZE"]["avg"], vals["TCC_EA0_RDREQ_sum"]["avg"]
    out["FETCH_SIZE_KB_avg"], out["WRITE_SIZE_KB_avg"], out["TCC_EA0_RDREQ_sum_avg"] = fetch_kb, write_kb, rd
    out["correction"] = "gfx950: read bytes = 2 * FETCH_SIZE * 1024 (128-B requests tallied at 64 B); cross-check TCC_EA0_RDREQ_sum * 128 B"
    out["hbm_bytes_per_launch"] = int(2 * fetch_kb * 1024 + write_kb * 1024)
    out["rdreq_bytes_per_launch"] = int(rd * 128)
    out["traffic_over_algorithmic"] = round(out["hbm_bytes_per_launch"] / alg, 4)
except Exception as e:
    out["error"] = str(e)
import os
json.dump(out, open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r02/r02_pmc_gemv_gateup.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
