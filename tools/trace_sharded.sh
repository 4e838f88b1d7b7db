export TMPDIR=/tmp
rm -rf gpurun_out/r6g_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6g_trace -o k -- python tools/emulate_sharded.py --steps 20 --no-graph --shapes grid:0.4,0.6:2 --out gpurun_out/r6g_emul.json > gpurun_out/r6g_trace.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6g_trace/**/k_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:40]:
    print(r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), r['Name'][:110])
PY
rm -f gpurun_out/r6g_trace/*/k_kernel_trace.csv
