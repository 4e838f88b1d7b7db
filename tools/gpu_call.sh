mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "sharded or pack_slices or rccl or bench_self" 2>&1 | tail -3
timeout 400 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.96 --link-gbps 61 --shapes grid:1:2 grid:2:2 grid:2:4 rows:1:1 --out gpurun_out/emul_g.json 2>&1 | grep -v Warn | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); p = r['per_propagate']
    print(r['layout'], r['phases'], r['return_chunks'], 'step %.3f x%.2f | prop %.3f pack %.3f win %.3f prod %.3f wout %.3f merge %.3f' % (r['step_ms_median'], r['projected_speedup'], p['total_ms'], p['pack_ms'], p['wait_in_ms'], p['product_ms'], p['wait_out_ms'], p['merge_ms']))
"
