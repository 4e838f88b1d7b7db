#!/bin/bash
# Run ON THE GPU BOX from the repo root: round 4, first capture -- the GPU test suite, then the evidence the round-3 review
# asked for: rocprofv3 kernel stats + PMC passes of the C3 / C5 steps AFTER csrc/tall.hip, the north star at 4x its size.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_errors_*.json
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $O/r4a_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r4a_pytest_gpu.log )
tail -5 $O/r4a_pytest_gpu.log
./tools/probes/tr_probe > $O/r4_tr_probe.txt 2>&1; tail -2 $O/r4_tr_probe.txt
timeout 400 python tools/northstar_x4.py > $O/r4_northstar_x4.log 2>&1; tail -c 400 $O/r4_northstar_x4.log
C="python tools/bench_configs.py"
export PYGSD_CONFIGS=C3,C5
rm -rf $O/r4a_prof $O/r4a_pmc_*
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4a_prof -o c3c5 -- $C > $O/r4a_prof.log 2>&1
cp $O/configs_partial.json $O/r4a_configs_c3c5.json
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/r4a_pmc_$tag -o c3c5 -- $C > $O/r4a_pmc_$tag.log 2>&1
done
ls $O/r4a_prof $O | head -50
