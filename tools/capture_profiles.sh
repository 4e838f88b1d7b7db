#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/capture_profiles.sh'):
# bench line, rocprofv3 kernel stats and the PMC passes (separate runs, counters only -- never combined
# with sys/hip/hsa traces), all into gpurun_out/; tools/pmc_summary.py then condenses them into profiles/.
set -u
R=${ROUND:-r2}
B="python bench.py --no-cpu-baseline --no-pmc --no-x4 --no-configs"
O=gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o $R -- $B --steps 10 --warmup 3 > $O/prof_stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o $R -- $B --steps 3 --warmup 1 > $O/prof_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o $R -- $B --steps 3 --warmup 1 > $O/prof_write.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/prof_l2 -o $R -- $B --steps 3 --warmup 1 > $O/prof_l2.log 2>&1
# memory-side request counters: all read requests of the L2 vs the ones routed to DRAM (MC)
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/prof_dram -o $R -- $B --steps 3 --warmup 1 > $O/prof_dram.log 2>&1
ls $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_l2 $O/prof_dram
tail -c 600 $O/bench_line.json
