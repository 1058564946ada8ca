#!/bin/bash
# round 6, after the hand-over without a progress word: rows per lane of the 64-lane sweeps again (GNX_W64_R / GNX_W64_RC) -- the constants of w64_pick_rows
out=gpurun_out/r6_rows2; mkdir -p $out; : > $out/rows.jsonl
for r in 6 8 10 16; do
  GNX_W64_R=$r timeout 600 python tools/long_pairs.py gpu affine_340k affine_1M affine_2M 2>> $out/err.log | sed "s/^{/{\"GNX_W64_R\": $r, /" >> $out/rows.jsonl
done
for r in 4 10; do
  GNX_W64_RC=$r timeout 600 python tools/long_pairs.py gpu const_150k const_300k_2M 2>> $out/err.log | sed "s/^{/{\"GNX_W64_RC\": $r, /" >> $out/rows.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r6_rows2/rows.jsonl"):
    r = json.loads(l)
    print(r.get("GNX_W64_R"), r.get("GNX_W64_RC"), r["case"], "sweep", r["sweep_ms"], "walk", r["walk_ms"], "call", r["call_s"], "ok", r["ok"])
PY
