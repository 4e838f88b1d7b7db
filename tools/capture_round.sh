#!/bin/bash
# Run ON THE GPU BOX from the repo root: the round's committed measurement set (ROUND=r5 by default; one parameterised script
# instead of one per capture).  GPU tests with the parity record on a named path; smoke; bench line, rocprofv3 kernel stats and PMC
# passes of the bench step (tools/capture_profiles.sh); per-configuration kernel stats + FETCH / WRITE / SQ passes
# (tools/capture_configs.sh); the operator-build probe with its kernel stats and counters; the magnetic configurations; the
# sharded rehearsals.  Summaries are condensed into profiles/ afterwards, off the box: tools/pmc_summary.py, tools/configs_summary.py.
set -u
R=${ROUND:-r6}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_errors_*.json
( PYGSD_PARITY_OUT=$O/${R}_parity_errors.json timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/${R}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${R}_pytest_gpu.log )
tail -4 $O/${R}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUND=$R bash tools/capture_profiles.sh > $O/${R}_capture_profiles.log 2>&1
cp $O/bench_line.json $O/${R}_bench_line.json
TAG=${R}j SQ_CONFIGS="C3a C5a" bash tools/capture_configs.sh > $O/${R}j_capture.log 2>&1
PYGSD_DENSE_GRAD=1 PYGSD_CONFIGS=C3a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_C3a_dense_grad -o k -- python tools/bench_configs.py > $O/${R}_prof_C3a_dense_grad.log 2>&1
cp $O/configs_partial.json $O/${R}_configs_C3a_dense_grad.json
find $O/${R}_prof_C3a_dense_grad -name "*kernel_trace.csv" -delete
timeout 300 python tools/build_probe.py --iters 10 > $O/${R}_build_probe.log 2>&1 && cp $O/build_probe.json $O/${R}_build_probe.json
rm -rf $O/${R}_prof_build $O/${R}_pmc_build_*
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_prof_build -o build -- python tools/build_probe.py --only fused,fused_signed,fused_real_weights --iters 5 > $O/${R}_prof_build.log 2>&1
find $O/${R}_prof_build -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $O/${R}_pmc_build_$c -o b -- python tools/build_probe.py --only fused,fused_signed,fused_real_weights --iters 2 > $O/${R}_pmc_build_$c.log 2>&1
done
PYGSD_CONFIGS=northstar,C2,C4 timeout 500 python tools/bench_configs.py > $O/${R}_configs_magnetic.log 2>&1; cp $O/configs_partial.json $O/${R}_configs_magnetic.json
timeout 300 python tools/uncached_step.py > $O/${R}_uncached_step.log 2>&1
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.74 --shapes grid:0.4,0.6:2 grid:2:2 grid:1:1 rows:1:1 --out $O/${R}_emulated_sharded_w8.json > $O/${R}_emulated_sharded_w8.log 2>&1
PYGSD_SHARD_MERGE_ON_READ=0 PYGSD_SHARD_PACKED_BACKWARD=0 timeout 300 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.74 --shapes grid:0.4,0.6:2 grid:2:2 --out $O/${R}_emulated_sharded_w8_no_shortcuts.json > $O/${R}_emulated_sharded_w8_no_shortcuts.log 2>&1
timeout 400 python tools/emulate_sharded.py --world 4 --single-gpu-ms 6.74 --shapes grid:0.4,0.6:2 grid:2:2 rows:1:1 --out $O/${R}_emulated_sharded_w4.json > $O/${R}_emulated_sharded_w4.log 2>&1
timeout 400 python tools/emulate_sharded.py --world 8 --signed --hidden 128 --K 2 --single-gpu-ms 30.0 --shapes grid:0.4,0.6:2 grid:2:2 --out $O/${R}_emulated_sharded_c4.json > $O/${R}_emulated_sharded_c4.log 2>&1
grep -E "^northstar|^C2|^C4" $O/${R}_configs_magnetic.log | cut -c1-300
cat $O/${R}_build_probe.log | cut -c1-160
cat $O/${R}_uncached_step.log | tail -2
tail -c 900 $O/${R}_bench_line.json
