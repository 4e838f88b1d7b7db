mkdir -p gpurun_out
timeout 900 python tools/emulate_sharded.py --world 8 --hidden 128 --K 2 --signed --single-gpu-ms 30.43 --link-gbps 61 --steps 5 --shapes grid:1:1 grid:2:2 grid:2:4 rows:2:1 --out gpurun_out/emulated_sharded_c4.json 2>&1 | grep -v Warn | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): print(ln.strip()[:200]); continue
    r = json.loads(ln)
    if 'error' in r: print(r); continue
    p = r['per_propagate']
    print(r['layout'], r['p_r'], r['p_c'], r['phases'], r['return_chunks'], 'step %.3f x%.2f | prop %.3f pack %.3f win %.3f prod %.3f wout %.3f merge %.3f wire %.3f' % (r['step_ms_median'], r['projected_speedup'], p['total_ms'], p['pack_ms'], p['wait_in_ms'], p['product_ms'], p['wait_out_ms'], p['merge_ms'], r['wire_ms_per_propagate']))
"
