#!/bin/bash
# Run ON THE GPU BOX: round 4 -- K = 1 forward dense stage in the dual SpMM's epilogue (pygsd_spmm2_k1_dense_f32) against the
# default two-kernel form at the north star: parity test, step time and rocprofv3 kernel stats for both (PYGSD_FUSE_K1=0 / 1).
set -u
O=gpurun_out
T=${TAG:-r4i}
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --timeout 300 -k "fused_k1 or operator" > $O/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_pytest.log )
tail -3 $O/${T}_pytest.log
for f in 0 1; do
  export PYGSD_FUSE_K1=$f PYGSD_CONFIGS=northstar
  rm -rf $O/${T}_prof_k1_$f
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_k1_$f -o k -- python tools/bench_configs.py > $O/${T}_k1_$f.log 2>&1
  rm -f $O/${T}_prof_k1_$f/k_kernel_trace.csv
  cp $O/configs_partial.json $O/${T}_configs_k1_$f.json
  grep "^northstar" $O/${T}_k1_$f.log | cut -c1-700
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/${T}_bench_k1_$f.json 2> $O/${T}_bench_k1_$f.err; cut -c1-330 $O/${T}_bench_k1_$f.json
done
