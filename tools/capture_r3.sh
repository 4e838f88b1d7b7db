#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/capture_r3.sh'): the round's measurement set into gpurun_out/
# (tools/pmc_summary.py with ROUND=r3 and tools/build_profile_summary.py then condense it into profiles/).
#   1. tools/capture_profiles.sh with ROUND=r3: bench line (N = 1), rocprofv3 kernel stats of the bench step, PMC passes
#      (FETCH_SIZE | WRITE_SIZE | L2 hit / miss | EA read requests; counters only, one pass per group)
#   2. the other BASELINE configs incl. the cached=False step (tools/bench_configs.py)
#   3. the operator build: probe (fused vs generic, unweighted and signed), rocprofv3 kernel stats of the unweighted fused
#      leg, FETCH_SIZE / WRITE_SIZE passes of the same command
#   4. one rank of an 8-rank and of a 4-rank job rehearsed on this GPU (tools/emulate_sharded.py)
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
ROUND=r3 bash tools/capture_profiles.sh > $O/r3_capture_profiles.log 2>&1
cp $O/bench_line.json $O/r3_bench_line.json
timeout 400 python tools/bench_configs.py > $O/r3_configs.log 2>&1 && cp $O/configs.json $O/r3_configs.json
timeout 120 python tools/build_probe.py --iters 10 > $O/r3_build_probe.log 2>&1 && cp $O/build_probe.json $O/r3_build_probe.json
rm -rf $O/r3_prof_build $O/r3_pmc_build_fetch $O/r3_pmc_build_write
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_build -o build -- python tools/build_probe.py --only fused --iters 5 > $O/r3_prof_build.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r3_pmc_build_fetch -o b -- python tools/build_probe.py --only fused --iters 2 > $O/r3_pmc_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3_pmc_build_write -o b -- python tools/build_probe.py --only fused --iters 2 > $O/r3_pmc_write.log 2>&1
cp $O/r3_build_probe.json $O/build_probe.json
timeout 200 python tools/emulate_sharded.py --world 8 > $O/r3_emulate_w8.log 2>&1 && cp $O/emulated_sharded.json $O/r3_emulated_sharded_w8.json
timeout 200 python tools/emulate_sharded.py --world 4 > $O/r3_emulate_w4.log 2>&1 && cp $O/emulated_sharded.json $O/r3_emulated_sharded_w4.json
tail -c 600 $O/r3_bench_line.json
