#!/bin/bash
# GPU box: per-kernel stats of the unweighted operator build (tools/build_probe.py --only fused), TAG names the output
TAG=${TAG:-r4}
cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/prof_build_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_build_$TAG -o build -- python tools/build_probe.py --only fused --iters 6 ${PROBE_ARGS:-} > gpurun_out/${TAG}_build_probe.log 2>&1
f=$(find gpurun_out/prof_build_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${TAG}_build_kernel_stats.csv
python - <<P
import csv
for r in csv.DictReader(open("gpurun_out/${TAG}_build_kernel_stats.csv")):
    n=r['Name']
    short = n[:70] if 'rocprim' not in n else 'rocprim:'+('onesweep ' if 'onesweep' in n else '')+('histogram ' if 'histogram' in n else '')+('scan' if 'scan' in n else '')
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>3} {r['Percentage']:>6}%  {short}")
P
tail -2 gpurun_out/${TAG}_build_probe.log
rm -f gpurun_out/prof_build_$TAG/*kernel_trace.csv
