out=gpurun_out/r6_e; mkdir -p $out
timeout 1500 python -m pytest tests/test_long_range.py tests/test_n1_gpu.py tests/test_ticket.py -x -q -k "w64 or natural or n1 or oracle and not affine_1M or megabase or ticket" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for rc in 10 4; do
GNX_W64_RC=$rc timeout 900 python tools/long_pairs.py gpu const_150k const_300k_2M 2>> $out/err.log | sed "s/^{/{\"GNX_W64_RC\": $rc, /" >> $out/long_pairs.jsonl
done
timeout 900 python tools/long_pairs.py gpu const_150k const_300k_2M affine_340k affine_1M >> $out/long_pairs.jsonl 2>> $out/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r6_e/long_pairs.jsonl"):
    r = json.loads(l)
    print(r.get("GNX_W64_RC"), r["case"], "call", r["call_s"], "first", r["first_call_s"], "sweep", r["sweep_ms"], "walk", r["walk_ms"], "ws", r["workspace_bytes"] / 1e9, "R", r.get("rows_per_lane"), r.get("snapshot_steps"), "ok", r["ok"], r.get("equals_oracle"), r["rescored_minus_score"], r["q1_restarts_changed"], r["launches"])
PY
