python -m pytest tests/test_gpu_kernels.py -q --no-header -k "gram" 2>&1 | tail -1
for w in 0 1; do PYGSD_GRAM_WIDE=$w PYGSD_CONFIGS=C5b PYGSD_CONFIGS_COMPACT=1 PYGSD_CONFIGS_OUT=gpurun_out/r6l_c5b_$w.json python tools/bench_configs.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('C5 '): continue
    r=json.loads(l.split(' ',1)[1])['bfloat16']; print('wide=$w', round(r['ms_per_block_step'],3), {k:round(v['launches_per_step']*v['ms_per_launch'],4) for k,v in r['kernels'].items() if v['launches_per_step']})"; done
