#!/bin/bash
# Run ON THE GPU BOX (round 5, fourth capture): SGCNConv after its node took the layer's own parameters; the weight-gradient product
# in its three forms at C3a / C5a; C3a with a dense upstream gradient under rocprofv3 (HIP-kernel share of the step); more shapes
# of the 8-rank rehearsal.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "sgcn or tall_gram or SGCN or sssnet" > $O/r5d_pytest_sgcn.log 2>&1; echo "rc=$?" >> $O/r5d_pytest_sgcn.log )
tail -6 $O/r5d_pytest_sgcn.log
for v in "0 0" "1 0" "1 1"; do
  set -- $v
  PYGSD_GRAM_32X32=$1 PYGSD_GRAM_RING=$2 PYGSD_CONFIGS=C3a,C5a timeout 400 python tools/bench_configs.py > $O/r5d_configs_gram_$1_$2.log 2>&1; cp $O/configs_partial.json $O/r5d_configs_gram_$1_$2.json
done
python - <<'PY'
import json
for tag in ("0_0", "1_0", "1_1"):
    d = json.load(open(f"gpurun_out/r5d_configs_gram_{tag}.json"))
    c3, c5 = d["C3_sgcnconv_first"], d["C5_digcn_inception_block_1gpu"]["float32"]
    print("gram32_ring", tag, "C3a step %.4f graph %.4f dense_bwd %.4f | C5a step %.3f dense_bwd %.4f" % (
        c3["ms_per_step"], c3["ms_per_step_hipgraph_replay"], c3["kernels"]["dense_bwd"]["ms_per_launch"], c5["ms_per_block_step"], c5["kernels"]["dense_bwd"]["ms_per_launch"]))
PY
rm -rf $O/r5d_prof_C3a
PYGSD_DENSE_GRAD=1 PYGSD_CONFIGS=C3a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5d_prof_C3a -o k -- python tools/bench_configs.py > $O/r5d_prof_C3a.log 2>&1
cp $O/configs_partial.json $O/r5d_configs_C3a_dense_grad.json
rm -f $O/r5d_prof_C3a/*/k_kernel_trace.csv $O/r5d_prof_C3a/k_kernel_trace.csv
find $O/r5d_prof_C3a -name "*kernel_stats.csv" | head -2
grep -E "^C3_sgcn" $O/r5d_prof_C3a.log | cut -c1-330
timeout 600 python tools/emulate_sharded.py --world 8 --single-gpu-ms 6.92 --shapes grid:0.4,0.6:0.34,0.33,0.33 grid:0.4,0.6:0.4,0.35,0.25 grid:0.35,0.65:2 grid:0.45,0.55:2 --out $O/r5d_emulated_w8.json > $O/r5d_emulated_w8.log 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d_emulated_w8.json"))
for r in d["runs"]:
    if "error" in r: print(r); continue
    p = r["per_propagate"]
    print(r["phases"], r["return_chunks"], "step %.3f host %.3f graph %s" % (r["step_ms_median"], r["host_issue_ms_per_step"], r["step_ms_hipgraph_replay"]),
          "prop %.3f pack %.3f in %.3f prod %.3f out %.3f merge %.3f" % (p["total_ms"], p["pack_ms"], p["wait_in_ms"], p["product_ms"], p["wait_out_ms"], p["merge_ms"]))
PY
