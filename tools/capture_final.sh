#!/bin/bash
# Run ON THE GPU BOX from the repo root: the consistent end-of-round set after the last kernel change -- per-configuration kernel
# stats + FETCH / WRITE passes (TAG), the bench line with its rocprofv3 stats and counter passes, the magnetic configurations.
set -u
export TMPDIR=/tmp
T=${TAG:-r6m}
R=${ROUND:-r6}
TAG=$T SQ_CONFIGS="" bash tools/capture_configs.sh > gpurun_out/${T}_capture.log 2>&1
ROUND=$R bash tools/capture_profiles.sh > gpurun_out/${R}_capture_profiles2.log 2>&1
cp gpurun_out/bench_line.json gpurun_out/${R}_bench_line.json
PYGSD_CONFIGS=northstar,C2,C4 timeout 500 python tools/bench_configs.py > gpurun_out/${R}_configs_magnetic.log 2>&1
cp gpurun_out/configs_partial.json gpurun_out/${R}_configs_magnetic.json
tail -c 400 gpurun_out/${R}_bench_line.json
