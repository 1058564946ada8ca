mkdir -p gpurun_out/ab
(timeout 900 python -m pytest tests/test_const_long.py -m gpu -x -q | tail -2
GNX_CL_WALK_NP=2 timeout 900 python -m pytest tests/test_const_long.py -m gpu -x -q | tail -2
for np in 4 2 1 4 2 1; do GNX_CL_WALK_NP=$np python bench.py --no-cpu --no-host --no-extras --verify 0 --steps 3 --warmup 1 --series long --pairs 1024 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('NP=$np', 'step %.3f ms' % d['ms_per_step'], d['kernel_ms'])"; done
for np in 4 1; do GNX_CL_WALK_NP=$np python bench.py --no-cpu --no-host --no-extras --verify 0 --steps 2 --warmup 1 --series long --pairs 2048 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('2048 NP=$np', 'step %.3f ms' % d['ms_per_step'], d['kernel_ms'])"; done
for np in 4 2 1; do GNX_CL_WALK_NP=$np python tools/bench_shapes.py const 320,10000,32768 2>/dev/null | cut -c1-400; GNX_CL_WALK_NP=$np python tools/bench_shapes.py const 3200,10000,3000 2>/dev/null | cut -c1-400; done
) 2>&1 | tee gpurun_out/ab/walknp.log
