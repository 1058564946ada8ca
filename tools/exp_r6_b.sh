out=gpurun_out/r6_i; mkdir -p $out
timeout 1500 python -m pytest tests/test_lat.py tests/test_gpu_parity.py tests/test_host_entry.py tests/test_cmds.py -m gpu -x -q > $out/pytest.log 2>&1; tail -3 $out/pytest.log
GNX_LIB_PATH=/root/repo/tools/lib_before_lat.so timeout 300 python tools/pair_latency.py 300 > $out/pair_latency_before.jsonl 2>> $out/err.log
timeout 300 python tools/pair_latency.py 300 > $out/pair_latency_after.jsonl 2>> $out/err.log
python - <<'PY'
import json
def rd(f):
    return [json.loads(l) for l in open(f) if l.startswith("{")]
b, a = rd("gpurun_out/r6_i/pair_latency_before.jsonl"), rd("gpurun_out/r6_i/pair_latency_after.jsonl")
for x, y in zip(b, a):
    ks = [k for k in x if isinstance(x[k], (int, float)) and ("us" in k or "ms" in k)]
    print({k: x[k] for k in x if k in ("fn", "shape", "n", "m", "mode")}, {k: (x[k], y.get(k)) for k in ks})
PY
