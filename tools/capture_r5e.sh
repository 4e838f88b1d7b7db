#!/bin/bash
# Run ON THE GPU BOX (round 5, fifth capture): the weighted bucket build (tests, probe), the whole operator-build test family, the
# gram forms once more, the sharded signed layers at size.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "operator_build or laplacian or tall_gram" > $O/r5e_pytest_build.log 2>&1; echo "rc=$?" >> $O/r5e_pytest_build.log )
tail -8 $O/r5e_pytest_build.log
timeout 300 python tools/build_probe.py --iters 10 --only fused,fused_signed,fused_real_weights,sorted_real_weights > $O/r5e_build_probe.log 2>&1 && cp $O/build_probe.json $O/r5e_build_probe.json
tail -4 $O/r5e_build_probe.log | cut -c1-200
rm -rf $O/r5e_prof_build
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5e_prof_build -o b -- python tools/build_probe.py --only fused_real_weights,fused_signed --iters 5 > $O/r5e_prof_build.log 2>&1
rm -f $O/r5e_prof_build/b_kernel_trace.csv
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r5e_prof_build/b_kernel_stats.csv")))[:16]:
    print("%6s %9.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY



