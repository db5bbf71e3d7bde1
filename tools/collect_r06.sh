#!/bin/bash
# usage (here, after `gpurun -- bash tools/r06_profiles.sh <commit>`): bash tools/collect_r06.sh  -> copies gpurun_out/r06p/* to profiles/r06_*
set -u
cd "$(dirname "$0")/.."
S=gpurun_out/r06p
for f in $S/bench*.json $S/*_kernel_stats.md $S/pmc_bench.txt $S/shape_sweep.json $S/shape_sweep.md $S/fp16_probe.txt $S/vit_probe.txt \
         $S/batch_sweep_*.json $S/batch_sweep_*.md; do
    [ -s "$f" ] && cp "$f" profiles/r06_$(basename "$f")
done
cp $S/dispatch_monotone.txt profiles/r06_dispatch_monotone_after.txt
cp $S/dispatch_monotone_fp8.txt profiles/r06_dispatch_monotone_fp8_after.txt
cp $S/r06_pmc_decode_traffic.json $S/r06_pmc_batch_traffic.json profiles/
python3 - <<'PY'
import glob, json, os, subprocess
want = subprocess.check_output(["python3", "tools/src_hash.py"]).decode().strip()
for f in sorted(glob.glob("profiles/r06_bench*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "UNREADABLE", e); continue
    ph, rf = j.get("phases", {}), j.get("roofline", {})
    src = (rf.get("traffic_source") or {})
    print(f"{os.path.basename(f):44s} {j['value']:8.1f} {j['unit']:9s} prefill {ph.get('prefill_ms', 0):6.2f} ms  decode {ph.get('decode_ms_per_token', 0):.3f}  "
          f"batched {ph.get('batched_decode_ms_per_step', 0):.3f}  ttft {ph.get('ttft_ms', 0):5.1f}  frac {rf.get('frac')}  traffic {rf.get('traffic')}  sha {src.get('csrc_sha16')}")
for f in ("profiles/r06_pmc_decode_traffic.json", "profiles/r06_pmc_batch_traffic.json"):
    print(f, json.load(open(f)).get("csrc_sha16"), "library sources:", want)
PY
