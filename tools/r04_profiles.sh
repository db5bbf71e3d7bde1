#!/bin/bash
# the round's judged measurements in one call: bench lines, rocprofv3 kernel summaries, PMC passes -> gpurun_out/r04p/
bash tools/refresh_profiles.sh
bash tools/profile_variants.sh
bash tools/pmc_bench.sh > /dev/null
bash tools/pmc_decode_traffic.sh ${1:-unknown} > /dev/null
ls gpurun_out/r04p | head -60
