#!/bin/bash
# Run ON THE GPU BOX from the repo root (TAG=<round tag>): ONE rocprofv3 kernel-stats run and two PMC passes (FETCH_SIZE,
# WRITE_SIZE; counters only, never combined with a trace) per configuration step, each step alone in its process so that a
# kernel's average belongs to one configuration: C3a SGCNConv, C3b SIMPA, C5a inception block fp32, C5b bf16
# (tools/bench_configs.py); tools/configs_summary.py <tag> condenses the result into profiles/<tag>_configs.json.
set -u
O=gpurun_out
T=${TAG:-r4b}
mkdir -p $O
export TMPDIR=/tmp
C="python tools/bench_configs.py"
for cfg in C3a C3b C5a C5b; do
  export PYGSD_CONFIGS=$cfg
  rm -rf $O/${T}_prof_$cfg $O/${T}_pmc_${cfg}_*
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_$cfg -o k -- $C > $O/${T}_prof_$cfg.log 2>&1
  cp $O/configs_partial.json $O/${T}_configs_$cfg.json
  rm -f $O/${T}_prof_$cfg/k_kernel_trace.csv
  for grp in "FETCH_SIZE" "WRITE_SIZE" ${EXTRA_PMC:-}; do
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/${T}_pmc_${cfg}_$grp -o k -- $C > $O/${T}_pmc_${cfg}_$grp.log 2>&1
  done
done
for cfg in ${SQ_CONFIGS:-C3a C5b}; do
  export PYGSD_CONFIGS=$cfg
  for grp in "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | tr ' ' '_')
    rm -rf $O/${T}_pmc_${cfg}_$tag
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/${T}_pmc_${cfg}_$tag -o k -- $C > $O/${T}_pmc_${cfg}_$tag.log 2>&1
  done
done
unset PYGSD_CONFIGS
ls $O | grep ${T}_ | head -60
