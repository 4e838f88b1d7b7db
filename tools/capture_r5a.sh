#!/bin/bash
# Run ON THE GPU BOX (round 5, first capture): GPU suite with the parity record written to a named path, the bench line
# with the dense-gradient passes and the in-run x4 measurement.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_errors_*.json
( PYGSD_PARITY_OUT=$O/r5a_parity_errors.json timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r5a_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r5a_pytest_gpu.log )
tail -5 $O/r5a_pytest_gpu.log
( time timeout 600 python bench.py > $O/r5a_bench_line.json 2> $O/r5a_bench.err ) 2>&1 | tail -3
tail -c 1500 $O/r5a_bench_line.json
tail -3 $O/r5a_bench.err
