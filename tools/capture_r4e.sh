#!/bin/bash
# Run ON THE GPU BOX: round 4, operator build -- tests of the build family, the probe (one-pass unweighted build vs the two-stage
# pipeline vs the generic one), rocprofv3 kernel stats + FETCH / WRITE passes of the one-pass leg, the uncached step.
set -u
O=gpurun_out
T=${TAG:-r4e}
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py -m gpu -q -p no:cacheprovider --timeout 600 -x -k "operator or laplacian or magnetic or kat or memo or tall or column or dense" > $O/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_pytest.log )
tail -4 $O/${T}_pytest.log
timeout 200 python tools/build_probe.py --iters 10 > $O/${T}_build_probe.log 2>&1 && cp $O/build_probe.json $O/${T}_build_probe.json
cat $O/${T}_build_probe.log | cut -c1-200
rm -rf $O/${T}_prof_build $O/${T}_pmc_build_*
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_build -o build -- python tools/build_probe.py --only fused --iters 5 > $O/${T}_prof_build.log 2>&1
rm -f $O/${T}_prof_build/build_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/${T}_pmc_build_$c -o b -- python tools/build_probe.py --only fused --iters 2 > $O/${T}_pmc_build_$c.log 2>&1
done
PYGSD_CONFIGS=northstar,C3a,C5a timeout 300 python tools/bench_configs.py > $O/${T}_configs.log 2>&1; cp $O/configs_partial.json $O/${T}_configs.json
grep -E "^northstar|^C3|^C5" $O/${T}_configs.log | cut -c1-900
head -12 $O/${T}_prof_build/build_kernel_stats.csv | cut -c1-160
