#!/bin/bash
# PMC passes over a real bench.py run (rounds 3-4): MFMA busy / wave wait buckets / L2 hit-miss per kernel of the step.
# Counters only (no trace domains beside --kernel-trace), the program itself after `--`.  -> gpurun_out/r06p/pmc_bench.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06p/pmc_bench.txt
mkdir -p $ROOT/gpurun_out/r06p
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum"
: > $OUT
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  timeout 900 rocprofv3 --kernel-trace --pmc $P -d /tmp/pmcb_$i -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --new 12 --no-cpu-baseline --no-graph > /tmp/pmcb_$i.log 2>&1
  DB=$(find /tmp/pmcb_$i -name "*.db" | head -1)
  echo "--- pass $i: $P" >> $OUT
  if [ -n "$DB" ]; then
    python3 - "$DB" >> $OUT <<'PY'
import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
by = {}
for k, c, n, v in rows:
    by.setdefault(k, {})[c] = (n, v)
def short(n):
    n = re.sub(r"^void ", "", n).replace("teo::", "").replace("unsigned short", "bf16").replace("unsigned char", "fp8")
    return re.sub(r"\(.*$", "", n)[:78]
keep = [k for k in by if "teo::" in k]
keep.sort(key=lambda k: -by[k].get("GRBM_GUI_ACTIVE", by[k].get("TCC_MISS_sum", (0, 0)))[0] * by[k].get("GRBM_GUI_ACTIVE", by[k].get("TCC_MISS_sum", (0, 1)))[1])
for k in keep[:22]:
    c = by[k]
    if "GRBM_GUI_ACTIVE" in c:
        g = c["GRBM_GUI_ACTIVE"][1]
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]
        wc, wa, wi, ac = (c.get(x, (0, 0.0))[1] for x in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
        # MFMA busy: busy cycles summed over the 1024 SIMDs / (GUI_ACTIVE cycles per XCD x 1024): GRBM_GUI_ACTIVE is summed over the 8 XCD instances
        print(f"{short(k):78s} n={c['GRBM_GUI_ACTIVE'][0]:5d} gui_active {g:10.0f} cyc | MFMA busy {100 * mf / (g / 8 * 1024) if g else 0:5.1f} % | of wave cycles: waiting {100 * wa / wc if wc else 0:5.1f} %, wait-inst {100 * wi / wc if wc else 0:5.1f} %, issuing {100 * ac / wc if wc else 0:5.1f} %")
    else:
        h, m = c.get("TCC_HIT_sum", (0, 0.0))[1], c.get("TCC_MISS_sum", (0, 0.0))[1]
        print(f"{short(k):78s} n={c.get('TCC_HIT_sum', (0, 0))[0]:5d} L2 hit {h:12.0f} miss {m:12.0f}  hit rate {100 * h / (h + m) if h + m else 0:5.1f} %")
PY
  else echo "no db"; tail -5 /tmp/pmcb_$i.log >> $OUT; fi
done
cat $OUT
