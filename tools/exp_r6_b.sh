out=gpurun_out/r6_b; mkdir -p $out
timeout 300 tools/alloc_probe.bin 48 > $out/alloc_probe2.txt 2>&1
timeout 1200 python -m pytest tests/test_long_range.py -x -q -k "megabase or natural" > $out/pytest_long.log 2>&1; tail -5 $out/pytest_long.log
cat $out/alloc_probe2.txt
