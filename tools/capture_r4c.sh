#!/bin/bash
# Run ON THE GPU BOX: round 4, third capture -- GPU tests, the C3 / C5 step timings, the packed-row probe (round 3's kernel
# against round 4's, crossover against one wavefront per row), a short bench line.
set -u
O=gpurun_out
T=${TAG:-r4c}
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_errors_*.json
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x > $O/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/${T}_pytest_gpu.log )
tail -5 $O/${T}_pytest_gpu.log
PYGSD_CONFIGS=C3,C5 timeout 300 python tools/bench_configs.py > $O/${T}_configs.log 2>&1; cp $O/configs_partial.json $O/${T}_configs_c3c5.json
grep -E "^C3|^C5" $O/${T}_configs.log | cut -c1-600
timeout 400 python tools/packed_probe.py > $O/${T}_packed_probe.log 2>&1; cp $O/packed_probe.json $O/${T}_packed_probe.json; tail -60 $O/${T}_packed_probe.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/${T}_bench.json 2> $O/${T}_bench.err; cut -c1-400 $O/${T}_bench.json
