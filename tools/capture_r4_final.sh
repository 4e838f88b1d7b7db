#!/bin/bash
# Run ON THE GPU BOX: round 4, the round's committed measurement set -- GPU tests; bench line, rocprofv3 kernel stats and PMC passes
# of the bench step (tools/capture_profiles.sh); per-configuration kernel stats + FETCH / WRITE / SQ passes (tools/capture_r4b.sh);
# the operator-build probe and its kernel stats; the uncached north-star step.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
TAG=r4j SQ_CONFIGS="C3a C5b" bash tools/capture_r4b.sh > $O/r4j_capture.log 2>&1
tail -3 $O/r4j_pytest_gpu.log
ROUND=r4 bash tools/capture_profiles.sh > $O/r4_capture_profiles.log 2>&1
cp $O/bench_line.json $O/r4_bench_line.json
timeout 200 python tools/build_probe.py --iters 10 > $O/r4_build_probe.log 2>&1 && cp $O/build_probe.json $O/r4_build_probe.json
rm -rf $O/r4_prof_build $O/r4_pmc_build_*
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_prof_build -o build -- python tools/build_probe.py --only fused --iters 5 > $O/r4_prof_build.log 2>&1
rm -f $O/r4_prof_build/build_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/r4_pmc_build_$c -o b -- python tools/build_probe.py --only fused --iters 2 > $O/r4_pmc_build_$c.log 2>&1
done
PYGSD_CONFIGS=northstar,C2,C4 timeout 400 python tools/bench_configs.py > $O/r4_configs_magnetic.log 2>&1; cp $O/configs_partial.json $O/r4_configs_magnetic.json
grep -E "^northstar|^C2|^C4" $O/r4_configs_magnetic.log | cut -c1-400
cat $O/r4_build_probe.log | cut -c1-160
tail -c 700 $O/r4_bench_line.json
