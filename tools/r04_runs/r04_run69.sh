#!/bin/bash
# which commit moved the bf16 / fp16 smoke numbers (8.49e-3 -> 7.51e-3)?  Same box, one library per commit.
cp teochat_amd/libteo_hip.so teochat_amd/libteo_hip_head.so
for h in 9e61e62 e7249ed 011e18f b4451d9 b3eed65 head; do
  [ -f teochat_amd/libteo_hip_$h.so ] || continue
  cp teochat_amd/libteo_hip_$h.so teochat_amd/libteo_hip.so
  echo "== $h"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "rel logits"
done
cp teochat_amd/libteo_hip_head.so teochat_amd/libteo_hip.so
