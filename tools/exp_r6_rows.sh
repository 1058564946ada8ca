#!/bin/bash
# round 6: rows per lane of the 64-lane affine sweep (GNX_W64_R) and its publish interval (GNX_W64_PUB: a switch of the build this ran on -- the progress words it spaced are gone, profiles/r6_experiments.md section 4) on the long pairs; one box
out=gpurun_out/r6_rows; mkdir -p $out
timeout 1200 python -m pytest tests/test_long_range.py -k "w64" -x -q > $out/pytest_w64.log 2>&1; tail -3 $out/pytest_w64.log
: > $out/rows.jsonl
for r in 6 8 10 16; do
  GNX_W64_R=$r timeout 600 python tools/long_pairs.py gpu affine_340k affine_1M 2>> $out/err.log | sed "s/^{/{\"GNX_W64_R\": $r, /" >> $out/rows.jsonl
done
timeout 600 python tools/long_pairs.py gpu affine_340k affine_1M affine_q1_300k 2>> $out/err.log | sed "s/^{/{\"GNX_W64_R\": \"auto\", /" >> $out/rows.jsonl
for pub in 16 32; do
  for r in 6 8; do
    GNX_W64_PUB=$pub GNX_W64_R=$r timeout 600 python tools/long_pairs.py gpu affine_340k affine_1M 2>> $out/err.log | sed "s/^{/{\"GNX_W64_PUB\": $pub, \"GNX_W64_R\": $r, /" >> $out/rows.jsonl
  done
done
for r in 10 16; do
  GNX_W64_R=$r timeout 600 python tools/long_pairs.py gpu affine_2M 2>> $out/err.log | sed "s/^{/{\"GNX_W64_R\": $r, /" >> $out/rows.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r6_rows/rows.jsonl"):
    r = json.loads(l)
    print(r.get("GNX_W64_R"), r.get("GNX_W64_PUB"), r["case"], "call", r["call_s"], "first", r["first_call_s"], "sweep", r["sweep_ms"], "walk", r["walk_ms"], "ws", r["workspace_bytes"] / 1e9, "ok", r["consumes_n_m"], r.get("equals_oracle"), r["rescored_minus_score"])
PY
