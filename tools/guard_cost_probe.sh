python -m pytest tests/test_gpu_nonfinite.py -q --no-header 2>&1 | tail -2
python -m pytest tests/test_gpu_kernels.py -q --no-header -k "tall or dense" 2>&1 | tail -1
python tools/tall_forms_probe.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for c in d: print(c['n'],c['K'],c['f_out'],'split',c['split']['ms'],'exact',c['exact']['ms'])"
python tools/dense_forms_probe.py 2>/dev/null > gpurun_out/r6_dense_forms_e.json
PYGSD_CONFIGS=northstar,C4 PYGSD_CONFIGS_COMPACT=1 PYGSD_CONFIGS_OUT=gpurun_out/r6k_mag.json python tools/bench_configs.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: n,r=l.split(' ',1); r=json.loads(r)
    except Exception: continue
    print(n, round(r['ms_per_step'],3), {k:round(v['launches_per_step']*v['ms_per_launch'],3) for k,v in r['kernels'].items() if v['launches_per_step']})"
