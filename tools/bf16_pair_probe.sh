# Round 6 experiment: spmm_vec_bf16_kernel<8> (one row per wavefront) against spmm_vec_bf16_pair_kernel (two) at C5b, plus parity
# tests of the bf16 paths under the pair kernel.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
PYGSD_BF16_PAIR=1 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py -q --no-header -k "bf16" 2>&1 | tail -2
for m in 0 1; do
  PYGSD_BF16_PAIR=$m PYGSD_CONFIGS=C5b PYGSD_CONFIGS_COMPACT=1 PYGSD_CONFIGS_OUT=gpurun_out/r6n_c5b_pair$m.json python tools/bench_configs.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('C5 '): continue
    r=json.loads(l.split(' ',1)[1])['bfloat16']; print('pair=$m step', round(r['ms_per_block_step'],3), 'spmm launch ms', round(r['kernels']['spmm']['ms_per_launch'],4), 'frac', round(r['fraction_of_8TBps'],3))"
  rm -rf gpurun_out/r6n_pmc_pair$m
  PYGSD_BF16_PAIR=$m PYGSD_CONFIGS=C5b PYGSD_CONFIGS_COMPACT=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/r6n_pmc_pair$m -o k -- python tools/bench_configs.py > /dev/null 2>&1
  python - <<PY
import csv,glob
tot={}
for f in glob.glob('gpurun_out/r6n_pmc_pair$m/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmm_vec_bf16' in r['Kernel_Name']:
            k=(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
            t=tot.setdefault(k,[0,0]); t[0]+=float(r['Counter_Value']); t[1]+=1
for k,(v,n) in sorted(tot.items()): print('pair=$m', k, 'per launch', v/n)
PY
done
