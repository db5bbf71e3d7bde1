#!/bin/bash
# HBM traffic of the kernels of the BATCHED decode step (config C5 per GPU: 8 conversations, fp8 weights) from PMC counters inside a real
# bench.py run -- the batched counterpart of tools/pmc_decode_traffic.sh, same recipe (MI355X_MICROARCH.md section HBM: separate rocprofv3
# --pmc passes with nothing but --kernel-trace beside them, FETCH_SIZE doubled on gfx950, WRITE_SIZE as is, TCC_EA0_RDREQ_sum x 128 B as the
# cross-check; the program itself follows `--`).  Writes gpurun_out/r06p/r06_pmc_batch_traffic.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
COMMIT=${1:-unknown}
mkdir -p $ROOT/gpurun_out/r06p
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  rm -rf /tmp/pmcb8_$C
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcb8_$C -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --new 12 --no-cpu-baseline --no-graph --batch 8 --weights fp8 > /tmp/pmcb8_$C.log 2>&1
done
python3 - "$COMMIT" <<'PY'
import glob, json, os, sqlite3, sys
B, CTX = 8, 2178          # 12 new tokens after a 2168-token prompt: mean context of the sampled launches
KERNELS = {   # name -> (test on the kernel name, algorithmic HBM bytes per launch)
    "attn_decode_whole": (lambda n: "attn_decode_whole_kernel<" in n, B * 2 * 32 * CTX * 128 * 2),
    "gateup_stream_fp8": (lambda n: "skinny_stream_kernel<" in n and "true, false>" in n, 22016 * 4096),
    "qkv_lmhead_stream_fp8": (lambda n: "skinny_stream_kernel<" in n and "false, false>" in n, int((32 * 12288 * 4096 + 32000 * 4096) / 33)),
    "o_tile_fp8": (lambda n: "skinny_gemm_kernel<" in n and ", 4, " in n, 4096 * 4096),
    "down_tile_fp8": (lambda n: "skinny_gemm_kernel<" in n and ", 6, " in n, 4096 * 11008),
}
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
    dbs = glob.glob(f"/tmp/pmcb8_{c}/**/*.db", recursive=True)
    if not dbs:
        raw[c] = {}
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = list(cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (c,)))
    raw[c] = {r[0]: (r[1], r[2]) for r in rows}
out = {"source": "tools/pmc_batch_traffic.sh: three separate rocprofv3 --kernel-trace --pmc passes over `python3 bench.py --steps 1 --warmup 0 --new 12 --no-cpu-baseline --no-graph --batch 8 --weights fp8`",
       "commit": sys.argv[1], "csrc_sha16": __import__("subprocess").check_output(["python3", os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/src_hash.py"]).decode().strip(),
       "correction": "gfx950: read bytes = 2 * FETCH_SIZE * 1024 (128-B requests tallied at 64 B) + WRITE_SIZE * 1024; cross-check TCC_EA0_RDREQ_sum * 128 B",
       "kernels": {}, "all_teo_kernels": sorted(n[:120] for n in raw.get("FETCH_SIZE", {}) if "teo::" in n and ("skinny" in n or "attn_decode" in n))}
for key, (match, alg) in KERNELS.items():
    e = {"algorithmic_bytes": alg}
    try:
        names = [n for n in raw["FETCH_SIZE"] if match(n)]
        name = max(names, key=lambda n: raw["FETCH_SIZE"][n][0])
        f, w, r = raw["FETCH_SIZE"][name], raw["WRITE_SIZE"].get(name, (0, 0.0)), raw["TCC_EA0_RDREQ_sum"].get(name, (0, 0.0))
        e.update({"kernel": name[:160], "launches_sampled": f[0], "FETCH_SIZE_KB_avg": f[1], "WRITE_SIZE_KB_avg": w[1], "TCC_EA0_RDREQ_sum_avg": r[1],
                  "hbm_bytes_per_launch": int(2 * f[1] * 1024 + w[1] * 1024), "rdreq_bytes_per_launch": int(r[1] * 128)})
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / alg, 4)
    except Exception as ex:  # noqa: BLE001
        e["error"] = str(ex)
    out["kernels"][key] = e
out["hbm_bytes_per_launch"] = out["kernels"].get("attn_decode_whole", {}).get("hbm_bytes_per_launch")        # the dominant kernel of this step
json.dump(out, open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r06p/r06_pmc_batch_traffic.json", "w"), indent=1)
print(json.dumps(out)[:4000])
PY
